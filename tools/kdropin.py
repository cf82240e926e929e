"""The drop-in C++ API client loop (nm_client_pair_loop) on the bench's 1080p pair: microseconds per pair."""
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
for wd in (0, 1):
    n = (C.c_int * 3)()
    us = nm.lib().nm_client_pair_loop(f[0].data_ptr(), f[1].data_ptr(), bench.W, bench.H, bench.CAP, reps, wd, n)
    print("with_distance %d: %.1f us per pair = %.1f pairs/s; keypoints %d %d matches %d" % (wd, us, 1e6 / us, n[0], n[1], n[2]))
