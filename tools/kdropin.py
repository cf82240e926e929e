"""The drop-in C++ API client loop (nm_client_pair_loop_ex) on the bench's 1080p pair: microseconds per pair, with the lazy
counts of nm/lazy_count.h (default) and with one synchronisation per octave (NM_EAGER_COUNTS), distance NULL / materialised
(MFMA pass / exact kernel), one and two client streams."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import niftymatch_amd as nm  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
f = bench.make_frames(nm, torch, dev, [0, 1])
lib = nm.lib()
eager = C.CDLL(nm.LIB_PATH).nm_set_eager_counts
for lazy in (1, 0):
    eager(0 if lazy else 1)
    for dmode in ("mfma", "exact"):
        nm.set_distance_mode(dmode)
        for wd in (0, 1):
            if wd == 0 and dmode == "exact":
                continue
            for streams in (1, 2):
                n = (C.c_int * 3)()
                us = lib.nm_client_pair_loop_ex(f[0].data_ptr(), f[1].data_ptr(), bench.W, bench.H, bench.CAP, reps, wd, streams, n)
                print("%s counts, distance %s, %d stream(s): %.1f us per pair = %.1f pairs/s; keypoints %d %d matches %d"
                      % ("lazy " if lazy else "eager", ("NULL" if not wd else dmode), streams, us, 1e6 / us, n[0], n[1], n[2]), flush=True)
eager(0)
