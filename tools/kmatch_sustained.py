"""The fused matcher launched back to back (no host sync in between), as in bench.py's match phase: per-launch time of
match_top2_kernel from the launcher's events, first launches vs steady state (clock / power behaviour under sustained
fp32 MFMA load)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import niftymatch_amd as nm
from niftymatch_amd import synth
dev = torch.device("cuda:0")
nA, nB = 12223, 12080
A = torch.from_numpy(synth.descriptors(1, nA)).to(dev) * 100
B = torch.from_numpy(synth.descriptors(2, nB)).to(dev) * 100
ws = nm.MatchWorkspace(nA, nB, dev)
res = torch.full((nA,), -1, dtype=torch.int32, device=dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
for a, b in ev:
    a.record(); b.record()
torch.cuda.synchronize()
for rep in range(2):
    for i in range(N):
        nm.profile_events(nm.PROF_MATCH_TOP2, ev[i][0], ev[i][1])
        nm.sift_match(A, B, 0.8, prior=res, workspace=ws)
    nm.profile_events(nm.PROF_MATCH_TOP2, None, None)
    torch.cuda.synchronize()
ts = [a.elapsed_time(b) * 1e3 for a, b in ev]
print("launch 0-3: %s us; launches 8-15 avg %.1f; last 16 avg %.1f us -> %.1f TFLOP/s" % (
    [round(t, 1) for t in ts[:4]], sum(ts[8:16]) / 8, sum(ts[-16:]) / 16, 256.0 * nA * nB / (sum(ts[-16:]) / 16 * 1e-6) / 1e12))
