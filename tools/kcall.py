"""One 64-frame nm_sift_detect_describe_batch call after the other on one stream (what a round of bench.py's headline does between
its matches): mean time per call over 30 calls. Environment switches of the library (NM_FRAME_SPLIT_DESCRIBE, ...) are read by the
library itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import niftymatch_amd as nm
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = bench.make_frames(nm, torch, dev, list(range(B)))
arenas = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(B)]
for _ in range(4):
    nm.detect_describe_batch(arenas, frames)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    nm.detect_describe_batch(arenas, frames)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / 30 * 1e6
print("call %s: %d frames %.1f us = %.2f us per frame (%d keypoints)" % (" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("NM_")),
                                                                        B, us, us / B, sum(int(a.num_items.item()) for a in arenas)))
