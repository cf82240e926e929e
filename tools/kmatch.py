"""Time the fused matcher in isolation (events around the MFMA kernel), 12223 x 12080 random descriptors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import niftymatch_amd as nm
from niftymatch_amd import synth
dev = torch.device("cuda:0")
nA, nB = int(os.environ.get("NA", 12223)), int(os.environ.get("NB", 12080))
A = torch.from_numpy(synth.descriptors(1, nA)).to(dev) * 100
B = torch.from_numpy(synth.descriptors(2, nB)).to(dev) * 100
ws = nm.MatchWorkspace(nA, nB, dev)
res = torch.full((nA,), -1, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); e1.record(); torch.cuda.synchronize()
ts = []
for i in range(12):
    nm.profile_events(nm.PROF_MATCH_TOP2, e0, e1)
    nm.sift_match(A, B, 0.8, prior=res, workspace=ws)
    nm.profile_events(nm.PROF_MATCH_TOP2, None, None)
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = sorted(ts[2:])
med = ts[len(ts) // 2]
print("screen", nm.get_match_screen(), "fallback rows", nm.match_fallback_count(ws, nA, nB), "of", nA,
      "second-pass rows", nm.match_second_pass_count(ws, nA, nB) if nm.get_match_screen() == "f16" else "-")
print("top2 kernel median %.1f us  min %.1f us  -> %.1f TFLOP/s (2NM128)" % (
    med * 1e3, ts[0] * 1e3, 256.0 * nA * nB / (med * 1e-3) / 1e12))
