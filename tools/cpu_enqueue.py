"""How long does the HOST need to enqueue one bench step (detect calls + matches) compared with the GPU time of the
step? Tells whether the throughput bench is launch-bound on the CPU."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import niftymatch_amd as nm  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
P = int(sys.argv[3]) if len(sys.argv) > 3 else 16
if len(sys.argv) > 5:
    bench.W, bench.H = map(int, sys.argv[5].split("x"))
dev = torch.device("cuda:0")
frames = bench.make_frames(nm, torch, dev, list(range(2 * P)))
arenas = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(2 * P)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
NB = 2 * P // B


T = int(sys.argv[4]) if len(sys.argv) > 4 else 1        # host threads issuing the calls (ctypes drops the GIL)
from concurrent.futures import ThreadPoolExecutor  # noqa: E402
pool = ThreadPoolExecutor(T) if T > 1 else None


def part(t):
    for c in range(t, NB, T):
        with torch.cuda.stream(streams[c % S]):
            nm.detect_describe_batch(arenas[c * B:(c + 1) * B], frames[c * B:(c + 1) * B])


def enqueue():
    if pool is None:
        part(0)
    else:
        list(pool.map(part, range(T)))


for _ in range(3):
    enqueue()
torch.cuda.synchronize()
host, total = [], []
for _ in range(5):
    t0 = time.perf_counter()
    enqueue()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0)
    total.append(t2 - t0)
print("%dx%d threads %d streams %d batch %d: %d frames: host enqueue %.2f ms, until GPU done %.2f ms -> %.0f us/frame host, %.0f us/frame total"
      % (bench.W, bench.H, T, S, B, 2 * P, 1e3 * min(host), 1e3 * min(total), 1e6 * min(host) / (2 * P), 1e6 * min(total) / (2 * P)))
