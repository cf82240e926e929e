"""Where the waves of a scale-space tile spend their cycles: reads the s_memtime stamps of a -DNM_CONV_STAMPS=1 build
(python tools/build_variant.py cstamps nm_pyramid.hip -DNM_CONV_STAMPS=1; NM_DIAGNOSTIC=1 NM_HIP_LIB=tools/_variants/libnm_hip_cstamps.so).
One wave in 61 workgroups of frame 0 of a 64-frame chain records; shares only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import niftymatch_amd as nm
dev = torch.device("cuda:0")
B = 64
frames = bench.make_frames(nm, torch, dev, list(range(B)))
arenas = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(B)]
f = nm.lib().nm_debug_conv_stamps
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; f.restype = C.c_int
buf = np.zeros(4096 * 8, np.uint64); n = np.zeros(1, np.uint32)
for _ in range(2):
    nm.scale_space_batch(arenas, frames, write_dog=False, write_grad=True)
torch.cuda.synchronize()
assert f(buf.ctypes.data, n.ctypes.data, 1) == 0
nm.scale_space_batch(arenas, frames, write_dog=False, write_grad=True)
torch.cuda.synchronize()
assert f(buf.ctypes.data, n.ctypes.data, 0) == 0
t = buf.reshape(4096, 8)[: min(int(n[0]), 4096)].astype(np.int64)
names = ["loads -> LDS", "barrier 1", "row pass", "barrier 2", "columns + epilogue + stores acknowledged"]
print("%d records; ticks per segment, mean over the recorded waves" % len(t))
for key in sorted(set(zip(t[:, 0] & 0xFFFFFFFF, t[:, 0] >> 32))):
    sel = t[(t[:, 0] & 0xFFFFFFFF == key[0]) & (t[:, 0] >> 32 == key[1])]
    if len(sel) < 8:
        continue
    m = sel[:, 1:6].mean(axis=0)
    print("R %2d grad %d width %4d (%4d waves): " % (key[0] // 2, key[0] & 1, key[1], len(sel)) +
          "  ".join("%s %5.0f (%2.0f %%)" % (nm_, x, 100 * x / m.sum()) for nm_, x in zip(names, m)) + "  | total %6.0f" % m.sum())
