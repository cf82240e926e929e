"""Exact-fallback rate of the fused matcher (rows the finalize pass cannot prove from the MFMA candidates) and call time
on Uniform[0,1) descriptors at several sizes, and on the bench's real 1080p SIFT descriptors."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import niftymatch_amd as nm  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)


def run(A, B, label):
    nA, nB = A.shape[0], B.shape[0]
    ws = nm.MatchWorkspace(nA, nB, dev)
    nm.sift_match_shard(A, B, 0, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        nm.sift_match_shard(A, B, 0, workspace=ws)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    fb = nm.match_fallback_count(ws, nA, nB)
    sp = nm.match_second_pass_count(ws, nA, nB) if nm.get_match_screen() == "f16" else 0
    print("%-28s %6d x %6d: second-pass rows %6d (%.3f %%), fallback rows %6d (%.3f %%), %.3f ms per call, %.1f TFLOP/s" % (
        label, nA, nB, sp, 100.0 * sp / nA, fb, 100.0 * fb / nA, ms, 256.0 * nA * nB / ms / 1e9))
    nm.sift_match(A, B, 0.8, workspace=ws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        nm.sift_match(A, B, 0.8, workspace=ws)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    fb = nm.match_fallback_count(ws, nA, nB)
    sp = nm.match_second_pass_count(ws, nA, nB) if nm.get_match_screen() == "f16" else 0
    print("%-28s   ratio-test call: second-pass rows %6d, fallback rows %6d, %.3f ms per call" % ("", sp, fb, ms))


sets = [(torch.rand((n, 128), device=dev, generator=g), torch.rand((n, 128), device=dev, generator=g), "uniform [0,1)")
        for n in (12000, 100000)]
sets.append((torch.rand((100000, 128), device=dev, generator=g), torch.rand((12500, 128), device=dev, generator=g),
             "uniform [0,1) shard"))
frames = bench.make_frames(nm, torch, dev, [0, 1])
ar = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(2)]
nm.detect_describe_batch(ar, frames)
torch.cuda.synchronize()
n0, n1 = int(ar[0].num_items.item()), int(ar[1].num_items.item())
sets.append((ar[0].desc[:n0].contiguous(), ar[1].desc[:n1].contiguous(), "1080p SIFT pair"))
for screen in ("f32", "bf16x3", "f16"):
    nm.set_match_screen(screen)
    for A, B, label in sets:
        run(A, B, screen + " " + label)
