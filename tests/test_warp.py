"""SURVEY.md 8(f) rows N3 (undistortion map + resample) and N4 (perspective warp + mosaicking blend): known answers of
the CPU restatement (oracle/nmo_warp.h), bit-exact GPU parity through the C ABI. Tolerance: 0 (same operation sequence
on both sides; see the header of oracle/nmo_warp.h for what the sampler spec fixes)."""
import numpy as np
import pytest


def _grid(w, h):
    x, y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    return np.ascontiguousarray(x), np.ascontiguousarray(y)


def _homography(rng, w, h, strength=1.0):
    a = rng.uniform(-0.05, 0.05) * strength
    s = 1.0 + rng.uniform(-0.08, 0.08) * strength
    H = np.array([[s * np.cos(a), -s * np.sin(a), rng.uniform(-0.06, 0.06) * w * strength],
                  [s * np.sin(a), s * np.cos(a), rng.uniform(-0.06, 0.06) * h * strength],
                  [rng.uniform(-2e-5, 2e-5) * strength, rng.uniform(-2e-5, 2e-5) * strength, 1.0]], np.float32)
    return H


def test_oracle_sampler_known_answers(oracle):
    # texel centres reproduce the texel; the reference always samples at coordinate + 0.5 (resample.cu:76,93,109)
    t = np.arange(12, dtype=np.float32).reshape(3, 4)
    x, y = _grid(4, 3)
    np.testing.assert_array_equal(oracle.resample_undistort(t, x, y), (t * np.float32(255.9999)).astype(np.float32))
    # half-way between texels: exact average (weights 1/2 are exact in 1.8 fixed point)
    r = oracle.resample_undistort(t, x[:, :3] + 0.5, y[:, :3])
    np.testing.assert_array_equal(r, ((t[:, :3] + t[:, 1:]) * np.float32(0.5) * np.float32(255.9999)).astype(np.float32))
    # border addressing: half a texel outside blends with 0, one texel outside IS 0
    r = oracle.resample_undistort(t, np.array([[-0.5, -1.0, 3.5, 4.0]], np.float32), np.zeros((1, 4), np.float32))
    np.testing.assert_array_equal(r[0], np.array([0.0, 0.0, 1.5, 0.0], np.float32) * np.float32(255.9999))
    # weights are quantised to 1/256: a fraction of 0.3 acts as round(0.3*256)/256 = 77/256
    r = oracle.resample_undistort(t, np.array([[1.3]], np.float32), np.array([[0.0]], np.float32))[0, 0]
    a = np.float32(77.0 / 256.0)
    # frac(1.3f) differs from 0.3 in the last bits but rounds to the same 1/256 step
    assert r == ((np.float32(1) - a) * np.float32(1.0) + a * np.float32(2.0)) * np.float32(255.9999)
    # NaN / inf coordinates fetch 0
    bad = np.array([[np.nan, np.inf, -np.inf]], np.float32)
    assert (oracle.resample_undistort(t, bad, np.zeros((1, 3), np.float32)) == 0).all()
    # 8-bit textures are read normalised: 255 -> 1.0 -> 255.9999
    u8 = np.array([[255, 0], [51, 102]], np.uint8)
    x2, y2 = _grid(2, 2)
    r = oracle.resample_undistort(u8, x2, y2)
    assert r[0, 0] == np.float32(255.9999) and r[0, 1] == 0 and r[1, 0] == np.float32(51) / np.float32(255) * np.float32(255.9999)


def test_oracle_undistort_map_known_answers(oracle):
    x, y = _grid(9, 7)
    cam = np.array([100.0, 120.0, 4.0, 3.0], np.float32)
    # no distortion: identity up to the rounding of (x-c)/f*f+c
    u, v = oracle.undistort_map(x, y, cam, np.zeros(3, np.float32))
    np.testing.assert_allclose(u, x, atol=1e-5)
    np.testing.assert_allclose(v, y, atol=1e-5)
    # the principal point is a fixed point for any k
    u, v = oracle.undistort_map(x, y, cam, np.array([0.3, -0.1, 0.05], np.float32))
    assert u[3, 4] == 4.0 and v[3, 4] == 3.0
    # against the closed form in float64
    xn, yn = (x.astype(np.float64) - 4) / 100, (y.astype(np.float64) - 3) / 120
    r2 = xn * xn + yn * yn
    poly = 1 + 0.3 * r2 - 0.1 * r2 ** 2 + 0.05 * r2 ** 3
    np.testing.assert_allclose(u, xn * poly * 100 + 4, rtol=1e-6, atol=2e-6)
    np.testing.assert_allclose(v, yn * poly * 120 + 3, rtol=1e-6, atol=2e-6)


def test_oracle_perspective_and_blend_known_answers(oracle):
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (12, 16, 4), dtype=np.uint8)
    eye = np.eye(3, dtype=np.float32)
    out, xp, yp = oracle.resample_perspective(frame, 16, 12, eye, inverse=True)
    gx, gy = _grid(16, 12)
    np.testing.assert_array_equal(xp, gx)
    np.testing.assert_array_equal(yp, gy)
    np.testing.assert_array_equal(out, frame)          # (c/255)*255.9999 truncates back to c for every c in 0..255
    # pure integer translation, forward and inverse
    T = np.array([[1, 0, 3], [0, 1, 2], [0, 0, 1]], np.float32)
    fwd, _, _ = oracle.resample_perspective(frame, 16, 12, T, inverse=False)
    np.testing.assert_array_equal(fwd[:10, :13], frame[2:, 3:])
    assert (fwd[10:] == 0).all() and (fwd[:, 13:] == 0).all()              # border texels are 0
    inv, _, _ = oracle.resample_perspective(frame, 16, 12, T, inverse=True)
    np.testing.assert_array_equal(inv[2:, 3:], frame[:10, :13])
    # blend: first frame is copied where the mask is set, alpha 255, weights stored
    canvas = np.zeros((20, 24, 4), np.uint8)
    cw = np.zeros((20, 24), np.float32)
    mask = np.ones((12, 16), np.float32)
    mask[:, :4] = 0
    wts = np.full((12, 16), 0.25, np.float32)
    c1, w1 = oracle.transform_blend(canvas, cw, frame, 16, 12, eye, 5, 6, mask, wts)
    np.testing.assert_array_equal(c1[6:18, 9:21, :3], frame[:, 4:, :3])
    assert (c1[6:18, 9:21, 3] == 255).all() and (c1[6:18, 5:9] == 0).all() and c1[:6].max() == 0
    assert (w1[6:18, 9:21] == 0.25).all() and w1.sum() == np.float32(0.25) * 12 * 12
    # blending the same frame again with a different weight keeps the colours (up to the 255.9999 truncation: <= 1 level)
    c2, w2 = oracle.transform_blend(c1, w1, frame, 16, 12, eye, 5, 6, mask, wts * 3)
    assert (w2[6:18, 9:21] == 1.0).all()
    assert np.abs(c2[6:18, 9:21, :3].astype(int) - frame[:, 4:, :3].astype(int)).max() <= 1
    # canvas clipping: offsets that push the frame off the canvas write nothing outside
    c3, w3 = oracle.transform_blend(canvas, cw, frame, 16, 12, eye, 20, -5, np.ones((12, 16), np.float32), wts)
    assert w3[:7, 20:24].all() and w3.sum() == np.float32(0.25) * 7 * 4


@pytest.mark.gpu
def test_gpu_undistort_and_resample_match_oracle(nm, oracle, cuda):
    import torch
    rng = np.random.default_rng(11)
    for (w, h) in [(1920, 1080), (77, 53)]:
        x, y = _grid(w, h)
        cam = np.array([0.9 * w, 0.95 * w, w / 2 - 3.5, h / 2 + 1.25], np.float32)
        dist = np.array([-0.21, 0.07, -0.012], np.float32)
        u, v = nm.undistort_map(torch.from_numpy(x).to(cuda), torch.from_numpy(y).to(cuda), torch.from_numpy(cam).to(cuda),
                                torch.from_numpy(dist).to(cuda))
        ou, ov = oracle.undistort_map(x, y, cam, dist)
        np.testing.assert_array_equal(u.cpu().numpy().view(np.uint32), ou.view(np.uint32))
        np.testing.assert_array_equal(v.cpu().numpy().view(np.uint32), ov.view(np.uint32))
        # sample a float and an 8-bit texture at those (fractional, partly out-of-range) positions
        tex_f = rng.uniform(0, 1, (h, w)).astype(np.float32)
        tex_u = rng.integers(0, 256, (h, w), dtype=np.uint8)
        ou2 = ou * np.float32(1.1) - np.float32(0.04 * w)          # push some samples outside
        for tex in (tex_f, tex_u):
            got = nm.resample_undistort(torch.from_numpy(tex).to(cuda), torch.from_numpy(ou2).to(cuda), v)
            np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), oracle.resample_undistort(tex, ou2, ov).view(np.uint32))
            gm = nm.resample_mask(torch.from_numpy(tex).to(cuda), torch.from_numpy(ou2).to(cuda), v, 0.4)
            np.testing.assert_array_equal(gm.cpu().numpy(), oracle.resample_mask(tex, ou2, ov, 0.4))
    bad = np.array([[np.nan, np.inf, -np.inf, 1e30]], np.float32)
    got = nm.resample_undistort(torch.from_numpy(tex_f).to(cuda), torch.from_numpy(bad).to(cuda), torch.zeros((1, 4), device=cuda))
    assert (got.cpu().numpy() == 0).all()


@pytest.mark.gpu
def test_gpu_perspective_resample_matches_oracle(nm, oracle, cuda):
    import torch
    rng = np.random.default_rng(12)
    for (w, h, cols, rows) in [(1920, 1080, 1920, 1080), (80, 60, 131, 47)]:
        frame = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        H = _homography(rng, w, h)
        for inverse in (True, False):
            out, xp, yp = nm.resample_perspective(torch.from_numpy(frame).to(cuda), cols, rows, torch.from_numpy(H).to(cuda), inverse)
            o_out, o_xp, o_yp = oracle.resample_perspective(frame, cols, rows, H, inverse)
            np.testing.assert_array_equal(xp.cpu().numpy().view(np.uint32), o_xp.view(np.uint32))
            np.testing.assert_array_equal(yp.cpu().numpy().view(np.uint32), o_yp.view(np.uint32))
            np.testing.assert_array_equal(out.cpu().numpy(), o_out)
            assert (o_out != 0).mean() > 0.5


@pytest.mark.gpu
def test_gpu_mosaic_blend_matches_oracle(nm, oracle, cuda):
    """Three warped frames accumulated into one canvas, as a mosaicking client does (transform_blend is in-place)."""
    import torch
    rng = np.random.default_rng(13)
    fw, fh, cw, ch = 320, 240, 480, 400
    canvas = np.zeros((ch, cw, 4), np.uint8)
    cwts = np.zeros((ch, cw), np.float32)
    t_canvas, t_cwts = torch.from_numpy(canvas).to(cuda), torch.from_numpy(cwts).to(cuda)
    yy, xx = np.mgrid[0:fh, 0:fw]
    wts = (np.minimum(np.minimum(xx, fw - 1 - xx), np.minimum(yy, fh - 1 - yy)) / 64.0 + 0.01).astype(np.float32)
    for k, (tx, ty) in enumerate([(40, 50), (90, 20), (-30, 130)]):
        frame = rng.integers(0, 256, (fh, fw, 4), dtype=np.uint8)
        mask = (rng.uniform(0, 1, (fh, fw)) > 0.1).astype(np.float32) if k != 1 else (rng.integers(0, 256, (fh, fw), dtype=np.uint8))
        H = _homography(rng, fw, fh, 0.5)
        nw, nh = fw + 20, fh + 10
        nm.transform_blend(t_canvas, t_cwts, torch.from_numpy(frame).to(cuda), nw, nh, torch.from_numpy(H).to(cuda), tx, ty,
                           torch.from_numpy(mask).to(cuda), torch.from_numpy(wts).to(cuda))
        canvas, cwts = oracle.transform_blend(canvas, cwts, frame, nw, nh, H, tx, ty, mask, wts)
        np.testing.assert_array_equal(t_cwts.cpu().numpy().view(np.uint32), cwts.view(np.uint32))
        np.testing.assert_array_equal(t_canvas.cpu().numpy(), canvas)
    assert (cwts > 0).mean() > 0.3 and (canvas[..., 3] == 255).sum() == (cwts > 0).sum()
