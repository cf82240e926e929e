import os, sys
sys.path.insert(0, os.getcwd())
import torch, bench, niftymatch_amd as nm
dev = torch.device("cuda:0")
frames = bench.make_frames(nm, torch, dev, list(range(8)))
ar = [nm.SiftArena(1920, 1080, 16384, device=dev) for _ in range(8)]
for a, f in zip(ar, frames): a.detect_describe(f)
torch.cuda.synchronize()
for p in range(4):
    a0, a1 = ar[2 * p], ar[2 * p + 1]
    nA, nB = int(a0.num_items.item()), int(a1.num_items.item())
    ws = nm.MatchWorkspace(16384, 16384, dev)
    res, _ = nm.sift_match(a0.desc, a1.desc, 0.8, workspace=ws, nA=nA, nB=nB)
    torch.cuda.synchronize()
    dB = a1.desc[:nB]; nb = (dB * dB).sum(1)
    dA = a0.desc[:nA]; na = (dA * dA).sum(1)
    uniqB = torch.unique(dB, dim=0).shape[0]
    print("pair", p, "fallback rows", nm.match_fallback_count(ws, nA, nB), "of", nA, "zero-desc A/B", int((na == 0).sum()), int((nb == 0).sum()),
          "duplicate rows in B", nB - uniqB, "min nb", float(nb.min()))
