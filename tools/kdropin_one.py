"""One configuration of the drop-in client loop, for rocprofv3 --kernel-trace --stats: argv = reps with_distance streams."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import niftymatch_amd as nm  # noqa: E402

dev = torch.device("cuda:0")
reps, wd, streams = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
f = bench.make_frames(nm, torch, dev, [0, 1])
n = (C.c_int * 3)()
us = nm.lib().nm_client_pair_loop_ex(f[0].data_ptr(), f[1].data_ptr(), bench.W, bench.H, bench.CAP, reps, wd, streams, n)
print("%.1f us per pair = %.1f pairs/s; keypoints %d %d matches %d" % (us, 1e6 / us, n[0], n[1], n[2]))
