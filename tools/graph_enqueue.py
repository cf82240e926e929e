"""Host cost and GPU throughput of replaying one captured HIP graph per detect call (vs eager launches, see
cpu_enqueue.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import niftymatch_amd as nm  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 16
P = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda:0")
frames = bench.make_frames(nm, torch, dev, list(range(2 * P)))
arenas = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(2 * P)]
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
NB = 2 * P // B
for c in range(NB):
    nm.detect_describe_batch(arenas[c * B:(c + 1) * B], frames[c * B:(c + 1) * B])
torch.cuda.synchronize()
graphs = []
for c in range(NB):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[c % S]):
        nm.detect_describe_batch(arenas[c * B:(c + 1) * B], frames[c * B:(c + 1) * B])
    graphs.append(g)
torch.cuda.synchronize()


def enqueue():
    for c in range(NB):
        with torch.cuda.stream(streams[c % S]):
            graphs[c].replay()


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
print("GRAPH streams %d batch %d: %d frames: host enqueue %.2f ms, until GPU done %.2f ms -> %.0f us/frame host, %.0f us/frame total"
      % (S, B, 2 * P, 1e3 * min(host), 1e3 * min(total), 1e6 * min(host) / (2 * P), 1e6 * min(total) / (2 * P)))
