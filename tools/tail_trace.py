#!/usr/bin/env python3
"""Per-item timeline of ONE octave-tail launch (csrc/nm_tail.hip) for a B-frame 1080p call: when each item drew its ticket, when
its inputs were ready and when it was done (100 MHz device clock), summarised per segment of the plan.
    python tools/tail_trace.py [frames_per_call] [width height]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["NM_TAIL_TRACE"] = "1"
import numpy as np
import torch
import niftymatch_amd as nm
import helpers as H

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
dev = torch.device("cuda:0")
ars = [nm.SiftArena(w, h, 16384, device=dev) for _ in range(B)]
fr = [torch.from_numpy(H.blurred_frame(s, w, h)).to(dev) for s in range(B)]
for _ in range(6):
    nm.detect_describe_batch(ars, fr) if B > 1 else ars[0].detect_describe(fr[0])
    torch.cuda.synchronize()
lib = nm.lib()
nseg = lib.nm_sift_arena_tail_segments(ars[0]._h)
segs = (C.c_int * (5 * nseg))()
ipf = lib.nm_sift_arena_tail_trace(ars[0]._h, None, 0, segs, nseg)
n = ipf * B
buf = (C.c_ulonglong * (16 * n))()
assert lib.nm_sift_arena_tail_trace(ars[0]._h, buf, n, None, 0) == ipf
allw = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
t = allw[:4 * n].reshape(n, 4)
ph = allw[4 * n:].reshape(n, 12)
t0 = t[:, 1].min()
us = lambda x: (x - t0) / 100.0
names = {0: "conv A / whole", 1: "conv B", 2: "detect", 3: "scan+gather", 4: "grad"}
print("%d frame(s), %d items per frame, launch span %.1f us, %d workgroups used" %
      (B, ipf, us(t[:, 3].max()), len(set((t[:, 0] >> 48).tolist()))))
print("%-16s %-7s %6s  %9s %9s %9s   %8s %8s" % ("segment", "octave", "items", "first tkt", "last rdy", "last done", "avg wait", "avg work"))
for i in range(nseg):
    kind, slot, per, first, o = segs[5 * i: 5 * i + 5]
    rows = t[B * first: B * (first + per)]
    print("%-16s %-7d %6d  %9.1f %9.1f %9.1f   %8.2f %8.2f" % (names[kind], o, per * B, us(rows[:, 1].min()), us(rows[:, 2].max()),
          us(rows[:, 3].max()), ((rows[:, 2] - rows[:, 1]) / 100.0).mean(), ((rows[:, 3] - rows[:, 2]) / 100.0).mean()))

print("phases of the conv items (us after inputs ready; zero, load, then per level: computed, published):")
for i in range(nseg):
    kind, slot, per, first, o = segs[5 * i: 5 * i + 5]
    if kind > 1:
        continue
    rows = slice(B * first, B * (first + per))
    d = (ph[rows] - t[rows, 2:3]) / 100.0
    d[ph[rows] == 0] = np.nan
    if per > 1 or kind == 1:                     # tiles: words 10 / 11 are the shader clock at stamp 0 and at the end
        pr = ph[rows]
        last = np.where(pr[:, 7] != 0, pr[:, 7], pr[:, 5])
        mhz = (pr[:, 11] - pr[:, 10]) / np.maximum((last - pr[:, 0]) / 100.0, 1e-9)
        print("  shader clock during these items: median %.0f MHz" % np.median(mhz))
        d[:, 10:] = np.nan
    print("  %-16s octave %d:" % (names[kind], o), " ".join("%6.1f" % v for v in np.nanmean(d, axis=0) if not np.isnan(v)))
