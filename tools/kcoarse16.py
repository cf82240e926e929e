"""match_coarse_kernel of a 16-pair call (one launch), SIFT-like descriptor sets of the headline's sizes: events around the launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import niftymatch_amd as nm
from niftymatch_amd import synth
dev = torch.device("cuda:0")
n = int(os.environ.get("PAIRS", 16))
nA, nB = int(os.environ.get("NA", 12223)), int(os.environ.get("NB", 12080))
As = [torch.from_numpy(synth.descriptors(2 * k + 1, nA)).to(dev) * 100 for k in range(n)]
Bs = [torch.from_numpy(synth.descriptors(2 * k + 2, nB)).to(dev) * 100 for k in range(n)]
res = [torch.full((nA,), -1, dtype=torch.int32, device=dev) for _ in range(n)]
ws = nm.MatchBatchWorkspace(n, nA, nB, dev)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]   # one pair of events per pair of the call:
for a, b in ev:                                                                                         # the first brackets the launch
    a.record(); b.record()
torch.cuda.synchronize()
ts = []
for i in range(40):
    keep = nm.profile_event_pairs(nm.PROF_MATCH_TOP2, ev)
    nm.sift_match_batch(As, Bs, [nA] * n, [nB] * n, res, 0.8, workspace=ws)
    nm.profile_event_pairs(nm.PROF_MATCH_TOP2, [])
    torch.cuda.synchronize()
    ts.append(ev[0][0].elapsed_time(ev[0][1]) * 1e3)
ts = sorted(ts[4:])
print("%s: %d pairs, coarse launch median %.1f us min %.1f -> %.2f us per pair; matches of pair 0: %d" % (
    os.environ.get("NM_HIP_LIB", "product"), n, ts[len(ts) // 2], ts[0], ts[len(ts) // 2] / n, int((res[0] >= 0).sum())))
