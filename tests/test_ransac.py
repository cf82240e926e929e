"""RANSAC stage: oracle properties on the CPU, bit-exact GPU parity, C++ API behaviour."""
import numpy as np
import pytest

TRUE_H = np.array([[1.02, 0.03, 12.0], [-0.02, 0.98, -7.0], [1e-5, -2e-5, 1.0]])


def _scene(n=600, outliers=200, seed=0, H=TRUE_H):
    rng = np.random.default_rng(seed)
    sx = rng.uniform(0, 1920, n).astype(np.float32)
    sy = rng.uniform(0, 1080, n).astype(np.float32)
    p = H @ np.stack([sx, sy, np.ones(n)])
    dx = (p[0] / p[2]).astype(np.float32)
    dy = (p[1] / p[2]).astype(np.float32)
    out = rng.choice(n, outliers, replace=False)
    dx[out] = rng.uniform(0, 1920, outliers)
    dy[out] = rng.uniform(0, 1080, outliers)
    sx[7] = sy[7] = dx[7] = dy[7] = -1                       # an unmatched row of align_points
    return sx, sy, dx, dy


def _lists(n, iterations, samples, seed=1):
    return np.random.default_rng(seed).integers(0, n, (iterations, samples)).astype(np.int32)


def test_oracle_ransac_properties(oracle):
    sx, sy, dx, dy = _scene()
    rl = _lists(600, 1500, 4)
    rl[3] = [5, 9, 5, 11]                                     # repeated index: skipped, H stays 0, count 0
    pos, Hb, Ha, inl = oracle.ransac(2, sx, sy, dx, dy, rl, 4.0)
    assert inl[3] == 0 and not Ha[3].any()
    assert inl[pos] == inl.max() and pos == int(np.argmax(inl)) and inl.max() >= 380
    np.testing.assert_allclose(Hb.reshape(3, 3) / Hb[8], TRUE_H, rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose((Hb.reshape(3, 3) / Hb[8])[:2, :2], TRUE_H[:2, :2], atol=2e-3)
    # translation: H = [1 0 tx; 0 1 ty; 0 0 1] of the sampled point, exactly
    T = np.array([[1, 0, 30.5], [0, 1, -12.25], [0, 0, 1.0]])
    sx, sy, dx, dy = _scene(H=T)
    rl = _lists(600, 400, 1)
    pos, Hb, Ha, inl = oracle.ransac(0, sx, sy, dx, dy, rl, 1.0)
    assert Hb.tolist()[:2] == [1.0, 0.0] and abs(Hb[2] - 30.5) < 1e-3 and abs(Hb[5] + 12.25) < 1e-3 and inl.max() >= 390
    i = rl[11, 0]
    assert Ha[11, 2] == dx[i] - sx[i] and Ha[11, 5] == dy[i] - sy[i]
    # similarity: rotation 10 deg, scale 1.1
    c, s = 1.1 * np.cos(np.deg2rad(10)), 1.1 * np.sin(np.deg2rad(10))
    S = np.array([[c, -s, 40.0], [s, c, 25.0], [0, 0, 1.0]])
    sx, sy, dx, dy = _scene(H=S)
    pos, Hb, Ha, inl = oracle.ransac(1, sx, sy, dx, dy, _lists(600, 800, 2), 2.0)
    np.testing.assert_allclose(Hb.reshape(3, 3), S, rtol=1e-3, atol=5e-2)
    assert inl.max() >= 390


@pytest.mark.gpu
@pytest.mark.parametrize("model,samples", [(0, 1), (1, 2), (2, 4)])
def test_gpu_ransac_matches_oracle(nm, oracle, cuda, model, samples):
    import torch
    sx, sy, dx, dy = _scene(n=3000, outliers=1200, seed=3)
    rl = _lists(3000, 4096, samples, seed=4)
    rl[17, :] = rl[17, 0]                                     # a skipped hypothesis (for models with > 1 sample)
    pos_r, Hb_r, Ha_r, inl_r = oracle.ransac(model, sx, sy, dx, dy, rl, 3.0)
    t = lambda a: torch.from_numpy(a).to(cuda)
    pos, Hb, Ha, inl = nm.ransac(model, t(sx), t(sy), t(dx), t(dy), t(rl), 3.0)
    torch.cuda.synchronize()
    assert np.array_equal(inl.cpu().numpy(), inl_r)
    got, ref = Ha.cpu().numpy(), Ha_r
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "hypotheses differ: %d" % (got != ref).any(1).sum()
    assert int(pos.item()) == pos_r and np.array_equal(Hb.cpu().numpy(), Hb_r)


@pytest.mark.gpu
def test_cpp_api_ransac(nm, cuda):
    sx, sy, dx, dy = _scene(n=800, outliers=250, seed=5)
    H = np.zeros(9, np.float32)
    rc = nm.lib().nm_client_ransac(2, sx.ctypes.data, sy.ctypes.data, dx.ctypes.data, dy.ctypes.data, 800, 4.0, 2000, 42,
                                   H.ctypes.data)
    assert rc == 1
    np.testing.assert_allclose(H.reshape(3, 3) / H[8], TRUE_H, rtol=3e-2, atol=3e-2)
    H2 = np.zeros(9, np.float32)
    nm.lib().nm_client_ransac(2, sx.ctypes.data, sy.ctypes.data, dx.ctypes.data, dy.ctypes.data, 800, 4.0, 2000, 42, H2.ctypes.data)
    assert np.array_equal(H, H2)                               # same seed, same answer
    few = np.full(10, -1.0, np.float32); few[:3] = [1, 2, 3]
    assert nm.lib().nm_client_ransac(2, few.ctypes.data, few.ctypes.data, few.ctypes.data, few.ctypes.data, 10, 4.0, 50, 1,
                                     H.ctypes.data) == 0       # fewer than 4 valid points
    assert nm.lib().nm_client_ransac(0, few.ctypes.data, few.ctypes.data, few.ctypes.data, few.ctypes.data, 10, 4.0, 50, 1,
                                     H.ctypes.data) == 1
