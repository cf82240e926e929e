"""The whole client pipeline the reference exists for (video mosaicking), every stage through the C ABI and checked
against the oracle bit for bit: BGRA frame -> warped second view (resample_perspective_transform) -> gray ->
SIFT detect/describe (one batched call) -> brute-force match -> align_points -> RANSAC homography (fixed sample list)
-> transform_blend of both views into one canvas. Also checks that the pipeline actually recovers the known motion."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def test_mosaic_pipeline_end_to_end(nm, oracle, cuda):
    import torch
    w, h = 480, 360
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    # view 0: textured BGRA frame; view 1: the same scene seen through a known homography (inverse warp of view 0)
    g = np.clip(H.blurred_frame(90, w, h, sigma=2.0) * 1.4, 0, 255).astype(np.uint8)
    view0 = np.stack([g, np.roll(g, 3, 1), np.roll(g, 5, 0), np.full_like(g, 255)], -1)
    true_H = np.array([[0.995, 0.02, 9.0], [-0.015, 1.005, -6.0], [1.5e-5, -1e-5, 1.0]], np.float32)
    view1_d, _, _ = nm.resample_perspective(t(view0), w, h, t(true_H), inverse=True)
    view1, _, _ = oracle.resample_perspective(view0, w, h, true_H, inverse=True)
    assert np.array_equal(view1_d.cpu().numpy(), view1)

    # gray + SIFT of both views in ONE batched call
    arenas = [nm.SiftArena(w, h, 8192) for _ in range(2)]
    grays = [nm.grayscale(t(view0)), nm.grayscale(view1_d)]
    nm.detect_describe_batch(arenas, grays)
    torch.cuda.synchronize()
    refs = [oracle.sift_detect_describe(oracle.grayscale(v), 8192) for v in (view0, view1)]
    n = [int(a.num_items.item()) for a in arenas]
    assert n == [r["n"] for r in refs] and min(n) > 300
    for a, r, k in zip(arenas, refs, n):
        assert np.array_equal(a.desc[:k].cpu().numpy().view(np.uint32), r["desc"].view(np.uint32))

    # match view 0 -> view 1, gather matched coordinates
    res, _ = nm.sift_match(arenas[0].desc, arenas[1].desc, 0.8, nA=n[0], nB=n[1])
    ref_m, _, _ = oracle.sift_matches(refs[0]["desc"], refs[1]["desc"], 0.8, want_distance=False)
    assert np.array_equal(res.cpu().numpy(), ref_m) and (ref_m >= 0).sum() > 80
    pts = nm.align_points(arenas[0].x[:n[0]].contiguous(), arenas[0].y[:n[0]].contiguous(),
                          arenas[1].x[:n[1]].contiguous(), arenas[1].y[:n[1]].contiguous(), res)
    pts_r = oracle.align_points(refs[0]["x"], refs[0]["y"], refs[1]["x"], refs[1]["y"], ref_m)
    for a, b in zip(pts, pts_r):
        assert np.array_equal(a.cpu().numpy(), b)

    # RANSAC homography on the matched pairs (same sample list on both sides)
    matched = np.flatnonzero(ref_m >= 0)
    rl = matched[np.random.default_rng(7).integers(0, len(matched), (2048, 4))].astype(np.int32)
    pos, Hb, Ha, inl = nm.ransac(2, *pts, t(rl), 2.0)
    pos_r, Hb_r, Ha_r, inl_r = oracle.ransac(2, *pts_r, rl, 2.0)
    torch.cuda.synchronize()
    assert int(pos.item()) == pos_r and np.array_equal(inl.cpu().numpy(), inl_r)
    assert np.array_equal(Hb.cpu().numpy().view(np.uint32), Hb_r.view(np.uint32))
    # the recovered transform maps view-0 pixels to view-1 pixels: view1(x) = view0(true_H^-1 ... inverse warp), so the
    # point motion view0 -> view1 is true_H itself
    Hn = Hb_r.reshape(3, 3) / Hb_r[8]
    assert inl_r[pos_r] > 0.5 * len(matched)
    np.testing.assert_allclose(Hn, true_H.astype(np.float64), atol=0.6, rtol=0.05)
    np.testing.assert_allclose(Hn[:2, :2], true_H[:2, :2], atol=5e-3)

    # mosaic: view 0 at the canvas origin offset, view 1 warped back by the estimated motion
    cw, ch, tx, ty = 640, 480, 60, 50
    canvas, cwts = np.zeros((ch, cw, 4), np.uint8), np.zeros((ch, cw), np.float32)
    canvas_d, cwts_d = t(canvas), t(cwts)
    mask = np.ones((h, w), np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    wts = (np.minimum(np.minimum(xx, w - 1 - xx), np.minimum(yy, h - 1 - yy)) / 40.0 + 0.05).astype(np.float32)
    eye = np.eye(3, dtype=np.float32)
    for view, M in ((view0, eye), (view1, Hn.astype(np.float32))):
        nm.transform_blend(canvas_d, cwts_d, t(view), w, h, t(M), tx, ty, t(mask), t(wts))
        canvas, cwts = oracle.transform_blend(canvas, cwts, view, w, h, M, tx, ty, mask, wts)
        assert np.array_equal(canvas_d.cpu().numpy(), canvas)
        assert np.array_equal(cwts_d.cpu().numpy().view(np.uint32), cwts.view(np.uint32))
    # where both views contributed, the blend stays close to view 0 (the registration is good to a fraction of a pixel)
    both = cwts[ty:ty + h, tx:tx + w] > wts + 1e-6
    diff = np.abs(canvas[ty:ty + h, tx:tx + w, 0].astype(int) - view0[..., 0].astype(int))
    assert both.mean() > 0.7 and np.median(diff[both]) <= 3
    for a in arenas:
        a.close()


def test_native_client_matches_python_path(nm, cuda):
    """examples/pairs_native.cpp (C ABI + HIP runtime only, built by niftymatch_amd.build) generates the same synthetic
    pair, finds the same keypoints and the same number of matches as the Python-bound path."""
    import json
    import os
    import subprocess
    import torch
    exe = os.path.join(os.path.dirname(nm.LIB_PATH), "pairs_native")
    if not os.path.exists(exe):
        pytest.skip("native example not built")
    out = subprocess.check_output([exe, "2", "2", "1", "640x480"], timeout=240).decode().strip().splitlines()[-1]
    rep = json.loads(out)
    from niftymatch_amd import synth
    taps, r = nm.create_kernel_for_sigma(synth.preblur_sigma(640, 480))
    td = torch.from_numpy(taps).to(cuda)
    arenas = [nm.SiftArena(640, 480, 16384) for _ in range(2)]
    grays = [nm.convolve(torch.from_numpy(synth.noise_frame(s, 640, 480)).to(cuda), td, r) for s in (0, 1)]
    nm.detect_describe_batch(arenas, grays)
    torch.cuda.synchronize()
    n = [int(a.num_items.item()) for a in arenas]
    res, _ = nm.sift_match(arenas[0].desc, arenas[1].desc, 0.8, nA=n[0], nB=n[1])
    assert rep["keypoints_pair0"] == n and rep["matches_pair0"] == int((res >= 0).sum().item())
    assert rep["frame_pairs_per_s"] > 0
