#!/usr/bin/env python3
"""Manual differential campaign (not collected by pytest): many random frame geometries / contents and matcher shapes,
HIP path vs the CPU oracle, bit for bit. Usage: python tests/soak_campaign.py [frames] [matches] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import helpers as H  # noqa: E402
import niftymatch_amd as nm  # noqa: E402
import oracle_lib as O  # noqa: E402

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n_match = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 12345)
dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0
t0 = time.time()
for it in range(n_frames):
    w = int(rng.integers(24, 400)); h = int(rng.integers(24, 300))
    if rng.random() < 0.5:
        w = (w // 4) * 4 + (0 if rng.random() < 0.8 else 1)        # mostly the packed path, sometimes the generic one
    nb = int(rng.integers(1, 4))
    cap = int(rng.choice([64, 500, 4096]))
    frames = []
    for k in range(nb):
        f = H.blurred_frame(int(rng.integers(1, 1 << 30)), w, h, sigma=float(rng.uniform(0.8, 4.0)))
        f = f * np.float32(rng.choice([1e-4, 0.05, 1.0, 3.0, 100.0]))
        if rng.random() < 0.2:
            f[: h // 3] = 0                                            # flat region
        if rng.random() < 0.2:
            f = np.round(f).astype(np.float32)                         # quantised image: many exact ties / zeros
        frames.append(np.ascontiguousarray(f, dtype=np.float32))
    arenas = [nm.SiftArena(w, h, cap) for _ in range(nb)]
    nm.detect_describe_batch(arenas, [t(f) for f in frames])
    torch.cuda.synchronize()
    for a, f in zip(arenas, frames):
        ref = O.sift_detect_describe(f, cap)
        n = int(a.num_items.item())
        ok = n == ref["n"]
        if ok:
            ok = (np.array_equal(a.kpts[:n].cpu().numpy().view(np.uint32), ref["kpts"].view(np.uint32))
                  and np.array_equal(a.orients[:n].cpu().numpy().view(np.uint32), ref["orient"].view(np.uint32))
                  and np.array_equal(a.desc[:n].cpu().numpy().view(np.uint32), ref["desc"].view(np.uint32)))
        if not ok:
            bad += 1
            print("FRAME MISMATCH", it, w, h, cap, n, ref["n"], flush=True)
        a.close()
print("frames done: %d cases, %d mismatches, %.1fs" % (n_frames, bad, time.time() - t0), flush=True)
t0 = time.time()
for it in range(n_match):
    na = int(rng.integers(1, 3000)); nb = int(rng.integers(1, 3000))
    kind = rng.choice(["uniform", "sparse", "int", "clustered"])
    A = rng.uniform(0, 1, (na, 128)).astype(np.float32)
    B = rng.uniform(0, 1, (nb, 128)).astype(np.float32)
    if kind == "sparse":
        A = np.where(A > 0.7, A, 0).astype(np.float32) * 300; B = np.where(B > 0.7, B, 0).astype(np.float32) * 300
    elif kind == "int":
        A = np.floor(A * 8).astype(np.float32); B = np.floor(B * 8).astype(np.float32)      # massive exact ties
    elif kind == "clustered":
        c = rng.uniform(0, 1, (8, 128)).astype(np.float32)
        A = (c[rng.integers(0, 8, na)] + 1e-3 * A).astype(np.float32); B = (c[rng.integers(0, 8, nb)] + 1e-3 * B).astype(np.float32)
    for _ in range(int(rng.integers(0, 6))):                          # planted duplicates
        B[int(rng.integers(0, nb))] = A[int(rng.integers(0, na))]
    amb = float(rng.choice([0.8, 0.6, 1.0, 1.5]))
    ref, _, (m1, ix, m2) = O.sift_matches(A, B, amb, want_distance=False)
    got, _ = nm.sift_match(t(A), t(B), amb)
    tri = nm.sift_match_shard(t(A), t(B), 3)
    ok = (np.array_equal(got.cpu().numpy(), ref) and np.array_equal(tri[1].cpu().numpy(), ix + 3)
          and np.array_equal(tri[0].cpu().numpy(), m1) and np.array_equal(tri[2].cpu().numpy(), m2))
    if not ok:
        bad += 1
        print("MATCH MISMATCH", it, na, nb, kind, amb, flush=True)
print("matches done: %d cases, total mismatches %d, %.1fs" % (n_match, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
