#!/usr/bin/env python3
"""The materialised distance matrix of compute_sift_matches on one 1080p pair's real descriptors (~12k x ~12k): the fp32 MFMA
pass against the exact VALU kernel. Whole call (distance + match) and the distance kernel alone (library profile site)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import niftymatch_amd as nm

dev = torch.device("cuda:0")
W, H, CAP = bench.W, bench.H, bench.CAP
frames = bench.make_frames(nm, torch, dev, [0, 1])
ar = [nm.SiftArena(W, H, CAP, device=dev) for _ in range(2)]
nm.detect_describe_batch(ar, frames)
torch.cuda.synchronize()
nA, nB = int(ar[0].num_items.item()), int(ar[1].num_items.item())
sets = {"sift": (ar[0].desc[:nA].contiguous(), ar[1].desc[:nB].contiguous())}
g = torch.Generator(device=dev).manual_seed(1)
sets["uniform"] = (torch.rand((nA, 128), device=dev, generator=g), torch.rand((nB, 128), device=dev, generator=g))
ws = nm.MatchWorkspace(nA, nB, dev)
D = torch.empty((nA, nB), dtype=torch.float32, device=dev)
res = torch.full((nA,), -1, dtype=torch.int32, device=dev)
lib = nm.lib()
for name, (A, B) in sets.items():
    for mode in ("mfma", "exact"):
        nm.set_distance_mode(mode)
        e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        k = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        for a, b in e + k:
            a.record(); b.record()
        for (a, b), (ka, kb) in zip(e, k):
            nm.profile_events(nm.PROF_DISTANCE, ka, kb)
            a.record()
            assert lib.nm_sift_match_f32(A.data_ptr(), nA, B.data_ptr(), nB, D.data_ptr(), res.data_ptr(), 0.8, ws.buf.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream) == 0
            b.record()
            nm.profile_events(nm.PROF_DISTANCE, None, None)
        torch.cuda.synchronize()
        call = sorted(a.elapsed_time(b) for a, b in e[2:])
        kern = sorted(a.elapsed_time(b) for a, b in k[2:])
        listed, cap = nm.match_distance_listed(ws, nA, nB) if mode == "mfma" else (None, None)
        flops = 256.0 * nA * nB
        print("%-8s %-6s %d x %d  call %.1f us (median)  distance kernel %.1f us = %.1f TFLOP/s (2NM128) = %.3f of 157.3; %.0f GB/s written;"
              " listed blocks %s of cap %s (%d blocks in all)"
              % (name, mode, nA, nB, 1e3 * call[len(call) // 2], 1e3 * kern[len(kern) // 2],
                 flops / (kern[len(kern) // 2] * 1e-3) / 1e12 if mode == "mfma" else 0.0,
                 flops / (kern[len(kern) // 2] * 1e-3) / 1e12 / 157.3 if mode == "mfma" else 0.0,
                 4.0 * nA * nB / (kern[len(kern) // 2] * 1e-3) / 1e9 if mode == "mfma" else 0.0,
                 listed, cap, ((nA + 31) // 32) * ((nB + 31) // 32)), flush=True)
