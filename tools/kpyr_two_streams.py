"""Does the scale-space chain of one set of frames hide its latency-bound small octaves under another set's? Two sets of B frames:
the chains one after the other on ONE stream against the two chains on TWO streams. Wall time per pair of chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import niftymatch_amd as nm
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = bench.make_frames(nm, torch, dev, list(range(2 * B)))
arenas = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(2 * B)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(two):
    for k in range(2):
        with torch.cuda.stream(s[k if two else 0]):
            nm.scale_space_batch(arenas[k * B:(k + 1) * B], frames[k * B:(k + 1) * B], write_dog=False, write_grad=True)
for two in (False, True, False, True):
    for _ in range(3):
        run(two)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        run(two)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 10 * 1e6
    print("%s: 2 x %d frames %.1f us = %.2f us per frame" % ("two streams" if two else "one stream ", B, us, us / (2 * B)), flush=True)
