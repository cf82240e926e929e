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
    if rng.random() < 0.25:                                            # round 4: geometries whose octaves >= 2 take the tail launch
        w = int(rng.integers(256, 900)); h = int(rng.integers(200, 700))   # (tiled and whole-plane items, odd sizes)
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
    nm.set_detect_tall_min(1 if rng.random() < 0.5 else -1)            # round 5: 20-row detection groups forced on half the batches
    nm.detect_describe_batch(arenas, [t(f) for f in frames])
    torch.cuda.synchronize()
    nm.set_detect_tall_min(-1)
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
    if it % 500 == 499:
        print("  ... %d frame batches, %d mismatches so far, %.0fs" % (it + 1, bad, time.time() - t0), flush=True)
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
    if rng.random() < 0.4:                                            # magnitudes around the edges of the fp16 coarse pass
        A = A * np.float32(rng.choice([1e-7, 1e-4, 30.0, 2.0e4, 1.0e6])); B = B * np.float32(rng.choice([1e-7, 1e-4, 30.0, 2.0e4, 1.0e6]))
    for _ in range(int(rng.integers(0, 6))):                          # planted duplicates
        B[int(rng.integers(0, nb))] = A[int(rng.integers(0, na))]
    amb = float(rng.choice([0.8, 0.6, 1.0, 1.5]))
    ref, _, _ = O.sift_matches(A, B, amb, want_distance=False)
    m1, ix, m2 = O.sift_match_shard(A, B, 3)
    got, _ = nm.sift_match(t(A), t(B), amb)
    tri = nm.sift_match_shard(t(A), t(B), 3)
    ok = (np.array_equal(got.cpu().numpy(), ref) and np.array_equal(tri[1].cpu().numpy(), ix)
          and np.array_equal(tri[0].cpu().numpy(), m1) and np.array_equal(tri[2].cpu().numpy(), m2))
    if ok and it % 4 == 0:
        # round 5: the materialised distance on the fp32 MFMA -- EVERY entry within 1e-4 relative of the reference's chain
        # (exact copies exactly 0), on the same random families; and the indexes do not depend on the mode
        _, Dref, _ = O.sift_matches(A, B, amb, want_distance=True)
        got2, D = nm.sift_match(t(A), t(B), amb, want_distance=True)
        try:
            H.assert_distance(nm, D, Dref, "soak distance")
            ok = np.array_equal(got2.cpu().numpy(), ref)
        except AssertionError as e:
            print("   ", str(e)[:200], flush=True)
            ok = False
    if not ok:
        bad += 1
        print("MATCH MISMATCH", it, na, nb, kind, amb, flush=True)
    if it % 500 == 499:
        print("  ... %d matcher cases, %d mismatches so far, %.0fs" % (it + 1, bad, time.time() - t0), flush=True)
print("matches done: %d cases, total mismatches %d, %.1fs" % (n_match, bad, time.time() - t0), flush=True)

# ---- batched matcher: random groups of pairs through nm_sift_match_batch_f32 ----
t0 = time.time()
n_groups = max(5, n_match // 10)
for it in range(n_groups):
    k = int(rng.integers(1, nm.MATCH_MAX_BATCH + 1))
    As, Bs, refs = [], [], []
    for _ in range(k):
        na = int(rng.integers(0, 1500)) if rng.random() < 0.9 else 0
        nb = int(rng.integers(1, 1500))
        if rng.random() < 0.12:                                        # round 5: pairs of several multi-tile segments per workgroup
            na, nb = int(rng.integers(2500, 7000)), int(rng.integers(2500, 7000))   # (segments requested ahead ACROSS pairs)
        A = (rng.uniform(0, 1, (max(na, 1), 128)).astype(np.float32) * np.float32(rng.choice([1, 50, 400])))[:na]
        B = rng.uniform(0, 1, (nb, 128)).astype(np.float32) * np.float32(rng.choice([1, 50, 400]))
        if na and rng.random() < 0.5:
            B[int(rng.integers(0, nb))] = A[int(rng.integers(0, na))]
        As.append(A); Bs.append(B)
    amb = float(rng.choice([0.8, 0.7, 1.2]))
    res = [torch.full((max(len(a), 1),), -5, dtype=torch.int32, device=dev) for a in As]
    nm.sift_match_batch([t(a if len(a) else np.zeros((1, 128), np.float32)) for a in As], [t(b) for b in Bs],
                        [len(a) for a in As], [len(b) for b in Bs], res, amb)
    torch.cuda.synchronize()
    for a, b, r in zip(As, Bs, res):
        if len(a) == 0:
            ok = bool((r == -5).all())
        else:
            ref, _, _ = O.sift_matches(a, b, amb, want_distance=False, prior=np.full(len(a), -5, np.int32))
            ok = np.array_equal(r[:len(a)].cpu().numpy(), ref)
        if not ok:
            bad += 1
            print("BATCH MATCH MISMATCH", it, len(a), len(b), amb, flush=True)
print("batched matches done: %d groups, total mismatches %d, %.1fs" % (n_groups, bad, time.time() - t0), flush=True)

# ---- round 5: the drop-in C++ API's per-octave client loop with lazy counts (nm/lazy_count.h) on random geometries: never
#      looking / looking after every call / eager counts must agree on every count, descriptor and coordinate, and with the oracle
import ctypes as C  # noqa: E402
t0 = time.time()
n_lazy = max(5, n_frames // 20)
for it in range(n_lazy):
    w = int(rng.integers(40, 700)); h = int(rng.integers(40, 500))
    if rng.random() < 0.7:
        w = (w // 4) * 4
    cap = int(rng.choice([64, 500, 8192]))
    f = np.ascontiguousarray(H.blurred_frame(int(rng.integers(1, 1 << 30)), w, h, sigma=float(rng.uniform(1.0, 4.0))))
    ref = O.sift_detect_describe(f, cap)
    n_oct = ref["counts"].shape[0]
    watch = (C.c_int * (4 * n_oct))()
    pend = C.c_int(0)
    n = nm.lib().nm_client_lazy_counts(f.ctypes.data, w, h, cap, watch, n_oct, C.byref(pend))
    ok = n == ref["n"]
    run = 0
    for o in range(n_oct):
        cnt = [int(c) for c in ref["counts"][o]]
        ok = ok and [watch[4 * o + l] for l in range(3)] == cnt
        run = min(cap, run + sum(cnt))
        ok = ok and watch[4 * o + 3] == run
    if not ok:
        bad += 1
        print("LAZY COUNT MISMATCH", it, w, h, cap, n, ref["n"], flush=True)
print("lazy-count client loops done: %d cases, total mismatches %d, %.1fs" % (n_lazy, bad, time.time() - t0), flush=True)

# ---- side stages: front end, warps, blend, RANSAC ----
t0 = time.time()
n_side = max(10, n_frames // 5)
for it in range(n_side):
    w = int(rng.integers(8, 500)); h = int(rng.integers(8, 400))
    cols = int(rng.integers(8, 500)); rows = int(rng.integers(8, 400))
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    a = rng.uniform(-0.3, 0.3); sc = rng.uniform(0.7, 1.4)
    Hm = np.array([[sc * np.cos(a), -sc * np.sin(a), rng.uniform(-40, 40)], [sc * np.sin(a), sc * np.cos(a), rng.uniform(-40, 40)],
                   [rng.uniform(-3e-4, 3e-4), rng.uniform(-3e-4, 3e-4), 1.0]], np.float32)
    ok = True
    for inv in (True, False):
        out, xp, yp = nm.resample_perspective(t(img), cols, rows, t(Hm), inv)
        o_out, o_xp, o_yp = O.resample_perspective(img, cols, rows, Hm, inv)
        ok = ok and np.array_equal(out.cpu().numpy(), o_out) and np.array_equal(xp.cpu().numpy().view(np.uint32), o_xp.view(np.uint32))
    tex = rng.uniform(0, 1, (h, w)).astype(np.float32) if it % 2 else rng.integers(0, 256, (h, w), dtype=np.uint8)
    ok = ok and np.array_equal(nm.resample_undistort(t(tex), xp, yp).cpu().numpy().view(np.uint32),
                               O.resample_undistort(tex, o_xp, o_yp).view(np.uint32))
    ok = ok and np.array_equal(nm.resample_mask(t(tex), xp, yp, 0.3).cpu().numpy(), O.resample_mask(tex, o_xp, o_yp, 0.3))
    cam = np.array([rng.uniform(0.5, 2) * w, rng.uniform(0.5, 2) * w, w / 2 + rng.uniform(-5, 5), h / 2 + rng.uniform(-5, 5)], np.float32)
    dist = rng.uniform(-0.3, 0.3, 3).astype(np.float32)
    u, v = nm.undistort_map(xp, yp, t(cam), t(dist))
    ou, ov = O.undistort_map(o_xp, o_yp, cam, dist)
    ok = ok and np.array_equal(u.cpu().numpy().view(np.uint32), ou.view(np.uint32)) and np.array_equal(v.cpu().numpy().view(np.uint32), ov.view(np.uint32))
    cw, ch = int(rng.integers(16, 600)), int(rng.integers(16, 500))
    canvas = rng.integers(0, 256, (ch, cw, 4), dtype=np.uint8); cwts = (rng.uniform(0, 1, (ch, cw)) * (rng.uniform(0, 1, (ch, cw)) > 0.5)).astype(np.float32)
    mask = (rng.uniform(0, 1, (h, w)) > 0.2).astype(np.float32); wts = rng.uniform(0.01, 2, (h, w)).astype(np.float32)
    tcan, tcw = t(canvas), t(cwts)
    tx, ty = int(rng.integers(-50, 50)), int(rng.integers(-50, 50))
    nm.transform_blend(tcan, tcw, t(img), w + 7, h + 5, t(Hm), tx, ty, t(mask), t(wts))
    ocan, ocw = O.transform_blend(canvas, cwts, img, w + 7, h + 5, Hm, tx, ty, mask, wts)
    ok = ok and np.array_equal(tcan.cpu().numpy(), ocan) and np.array_equal(tcw.cpu().numpy().view(np.uint32), ocw.view(np.uint32))
    ok = ok and np.array_equal(nm.grayscale(t(img)).cpu().numpy().view(np.uint32), O.grayscale(img).view(np.uint32))
    n = int(rng.integers(8, 4000))
    sx = rng.uniform(0, 1920, n).astype(np.float32); sy = rng.uniform(0, 1080, n).astype(np.float32)
    p3 = Hm.astype(np.float64) @ np.stack([sx, sy, np.ones(n)])
    dx = (p3[0] / p3[2]).astype(np.float32); dy = (p3[1] / p3[2]).astype(np.float32)
    outl = rng.random(n) < 0.3
    dx[outl] = rng.uniform(0, 1920, outl.sum()); dy[outl] = rng.uniform(0, 1080, outl.sum())
    for model, ns in ((0, 1), (1, 2), (2, 4)):
        rl = rng.integers(0, n, (int(rng.integers(1, 700)), ns)).astype(np.int32)
        thr = float(rng.uniform(0.5, 9))
        pos, Hb, Ha, inl = nm.ransac(model, t(sx), t(sy), t(dx), t(dy), t(rl), thr)
        pos_r, Hb_r, Ha_r, inl_r = O.ransac(model, sx, sy, dx, dy, rl, thr)
        ok = ok and int(pos.item()) == pos_r and np.array_equal(inl.cpu().numpy(), inl_r)
        ok = ok and np.array_equal(Ha.cpu().numpy().view(np.uint32), Ha_r.view(np.uint32))
    if not ok:
        bad += 1
        print("SIDE MISMATCH", it, w, h, cols, rows, flush=True)
print("side stages done: %d cases, total mismatches %d, %.1fs" % (n_side, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
