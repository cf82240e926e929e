"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) count / avg / min microseconds."""
import collections
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-44:], "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]),
           r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
    agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print("%-46s %-18s %5s %5s %7s %6s %10s %10s %6s" % ("kernel", "grid", "vgpr", "agpr", "lds", "calls", "avg_us", "min_us", "pct"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-46s %-18s %5s %5s %7s %6d %10.1f %10.1f %6.2f" % (k[0], k[1], k[2], k[3], k[4], len(v), sum(v) / len(v) / 1e3,
                                                              min(v) / 1e3, 100.0 * sum(v) / tot))
