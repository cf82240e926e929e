"""The whole scale-space chain (nm_sift_scale_space_batch: base blur + 6 octaves) of B 1080p frames, alone on the
device: event-timed average, and -- under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` -- the dispatches that
tools/pmc_traffic_total.py sums into HBM bytes per frame.   usage (the interpreter itself after `--`, never this file or env:
the profiler initialises the GPU before the program starts, so any exec hop is forbidden on this pool):
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f --output-format csv -- python3 tools/kpyr_all.py 16 4 [nodog|dogonly]
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w --output-format csv -- python3 tools/kpyr_all.py 16 4 [nodog|dogonly]
and, unprofiled:  python3 tools/kpyr_all.py [B=16] [reps=10]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import niftymatch_amd as nm  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
frames = bench.make_frames(nm, torch, dev, list(range(B)))
arenas = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(B)]
mode = sys.argv[3] if len(sys.argv) > 3 else "all"             # "nodog": the chain the frame driver runs; "dogonly": levels + DoG
dog = mode != "nodog"                                          # planes without the fused gradient planes (108 B/px workload)
grad = mode != "dogonly"
for _ in range(2):
    nm.scale_space_batch(arenas, frames, write_dog=dog, write_grad=grad)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    nm.scale_space_batch(arenas, frames, write_dog=dog, write_grad=grad)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
px = sum((bench.W >> o) * (bench.H >> o) for o in range(6))
print("B %d sequences %d avg %.1f us per sequence = %.2f us per frame; 108 B/px: %.0f GB/s, 144 B/px: %.0f GB/s"
      % (B, reps + 2, ms * 1e3, ms * 1e3 / B, 108.0 * px * B / ms / 1e6, 144.0 * px * B / ms / 1e6))
