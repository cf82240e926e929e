"""Busy time and gaps of a single-stream kernel sequence from a rocprofv3 --kernel-trace run (rocpd database or CSV):
    python tools/trace_gaps.py <dir> [skip_first_n_kernels]
Prints the span, the sum of the kernel durations, the sum of the idle gaps between consecutive kernels and the gap histogram:
whether a launch-latency-bound chain (the drop-in API's per-octave calls) is short of kernel time or of back-to-back issue."""
import glob
import sqlite3
import sys

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = []
for f in glob.glob(path + "/**/*.db", recursive=True):
    cur = sqlite3.connect(f).cursor()
    rows += list(cur.execute("select start, end, name from kernels order by start"))
rows = rows[skip:]
busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
span = rows[-1][1] - rows[0][0]
print("%d kernels, span %.1f us, busy %.1f us (%.0f %%), gaps %.1f us; mean kernel %.2f us, mean gap %.2f us"
      % (len(rows), span / 1e3, busy / 1e3, 100.0 * busy / span, sum(gaps) / 1e3, busy / 1e3 / len(rows), sum(gaps) / 1e3 / max(1, len(gaps))))
edges = [0, 500, 1000, 2000, 4000, 8000, 16000, 10 ** 12]
for lo, hi in zip(edges, edges[1:]):
    g = [x for x in gaps if lo <= x < hi]
    print("  gaps %6.1f .. %8.1f us: %5d  (%.1f us in all)" % (lo / 1e3, min(hi, 10 ** 9) / 1e3, len(g), sum(g) / 1e3))
