"""Where a tile iteration of match_coarse_kernel spends its cycles: reads the s_memtime stamps of a -DNM_COARSE_STAMPS=1 build
(python tools/build_variant.py stamps nm_match.hip -DNM_COARSE_STAMPS=1; NM_DIAGNOSTIC=1 NM_HIP_LIB=tools/_variants/libnm_hip_stamps.so).
Shares only: the stamps' fences forbid overlaps the product has."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import niftymatch_amd as nm
from niftymatch_amd import synth
dev = torch.device("cuda:0")
n, nA, nB = 16, 12223, 12080
As = [torch.from_numpy(synth.descriptors(2 * k + 1, nA)).to(dev) * 100 for k in range(n)]
Bs = [torch.from_numpy(synth.descriptors(2 * k + 2, nB)).to(dev) * 100 for k in range(n)]
res = [torch.full((nA,), -1, dtype=torch.int32, device=dev) for _ in range(n)]
ws = nm.MatchBatchWorkspace(n, nA, nB, dev)
for _ in range(3):
    nm.sift_match_batch(As, Bs, [nA] * n, [nB] * n, res, 0.8, workspace=ws)
torch.cuda.synchronize()
buf = np.zeros(8 * 64 * 8, np.uint64)
f = nm.lib().nm_debug_coarse_stamps
f.argtypes = [C.c_void_p]; f.restype = C.c_int
assert f(buf.ctypes.data) == 0
t = buf.reshape(8, 64, 8).astype(np.int64)
names = ["0>1 fetchB slots half1", "1>2 fetchA half2", "2>3 fetchB slots half3", "3>4 slot write, vmcnt(0)", "4>5 barrier",
         "5>6 DMA issue, fetchA", "6>7 half4 + fold", "7>0' loop back"]
print("cycles per segment (s_memtime ticks), iterations 4..59, per wave and mean; stamp cost ~40 each")
rows = []
for w in range(8):
    d = np.diff(t[w, 4:60, :], axis=1)                       # 7 in-iteration segments
    back = t[w, 5:61 if False else 60, 0] - t[w, 4:59, 7] if False else (t[w, 5:60, 0] - t[w, 4:59, 7])
    seg = list(d.mean(axis=0)) + [back.mean()]
    rows.append(seg)
    print("wave %d: " % w + " ".join("%7.0f" % x for x in seg) + "  | iteration %7.0f" % (t[w, 5:60, 0] - t[w, 4:59, 0]).mean())
m = np.mean(rows, axis=0)
tot = m.sum()
for nme, x in zip(names, m):
    print("%-28s %7.0f  %5.1f %%" % (nme, x, 100 * x / tot))
print("iteration total %.0f ticks" % tot)
# the distribution of the barrier wait and of the iteration over the first iterations of wave 0
print("wave 0 iterations 0..15 (ticks):", [int(x) for x in (t[0, 1:17, 0] - t[0, 0:16, 0])])
