"""Element-wise stages either side of the SIFT path (SURVEY.md 8(f), N3 front end + align_points of N1):
oracle known answers on the CPU, bit-exact GPU parity."""
import numpy as np
import pytest


def _bgra(seed, w, h):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 4), dtype=np.uint8)


def test_oracle_image_ops_known_answers(oracle):
    px = np.array([[[10, 200, 30, 77], [255, 255, 255, 0], [0, 0, 0, 255]]], np.uint8)      # B, G, R, A
    g = oracle.grayscale(px)
    np.testing.assert_allclose(g[0], [0.07 * 10 + 0.72 * 200 + 0.21 * 30, 255.0, 0.0], rtol=1e-6)   # bgra_2_gray.cu:16
    assert oracle.extract_channel(px, 1)[0].tolist() == [200.0, 255.0, 0.0]
    assert (oracle.extract_channel(px, 7) == -7.0).all()                                   # other channels: untouched
    put = oracle.put_channel(px, np.array([[5.9, 6.1, 7.0]], np.float32), 2)
    assert put[0, :, 2].tolist() == [5, 6, 7] and (put[..., :2] == px[..., :2]).all()
    assert oracle.put_channel(px, np.zeros((1, 3), np.float32), 3)[0, :, 3].tolist() == [255, 255, 255]
    assert oracle.set_alpha(px, 9)[0, :, 3].tolist() == [9, 9, 9]
    src = np.array([[0.0, 1.9, 127.5, 200.0, 255.0]], np.float32)
    assert oracle.cast_f32_u8(src, 0)[0].tolist() == [0, 1, 127, 200, 255]
    assert oracle.cast_f32_u8(src, 128)[0].tolist() == [0, 1, 127, 128, 128]              # saturation (cast.cu:17-19)
    big = _bgra(0, 10, 7)
    assert np.array_equal(oracle.downsample2_u8x4(big, 5, 3), big[0:6:2, 0:10:2])
    sx, sy = np.arange(4, dtype=np.float32), np.arange(4, dtype=np.float32) + 10
    dx, dy = np.arange(6, dtype=np.float32) * 2, np.arange(6, dtype=np.float32) * 3
    csx, csy, cdx, cdy = oracle.align_points(sx, sy, dx, dy, np.array([5, -1, 0, 2], np.int32))
    assert csx.tolist() == [0, -1, 2, 3] and csy.tolist() == [10, -1, 12, 13]
    assert cdx.tolist() == [10, -1, 0, 4] and cdy.tolist() == [15, -1, 0, 6]


@pytest.mark.gpu
def test_gpu_image_ops_match_oracle(nm, oracle, cuda):
    import torch
    for (w, h) in [(1920, 1080), (61, 45)]:
        b = _bgra(w, w, h)
        tb = torch.from_numpy(b).to(cuda)
        assert np.array_equal(nm.grayscale(tb).cpu().numpy(), oracle.grayscale(b))
        for c in range(4):
            assert np.array_equal(nm.extract_channel(tb, c).cpu().numpy(), oracle.extract_channel(b, c))
        plane = np.random.default_rng(1).uniform(0, 255.99, (h, w)).astype(np.float32)
        tp = torch.from_numpy(plane).to(cuda)
        for c in range(4):
            assert np.array_equal(nm.put_channel(tb, tp, c).cpu().numpy(), oracle.put_channel(b, plane, c))
        assert np.array_equal(nm.set_alpha(tb, 17).cpu().numpy(), oracle.set_alpha(b, 17))
        for mv in (0, 200):
            assert np.array_equal(nm.cast_f32_u8(tp, mv).cpu().numpy(), oracle.cast_f32_u8(plane, mv))
        assert np.array_equal(nm.downsample2_u8x4(tb, w // 2, h // 2).cpu().numpy(), oracle.downsample2_u8x4(b, w // 2, h // 2))
    rng = np.random.default_rng(3)
    n, m = 5000, 4000
    sx, sy = rng.uniform(0, 1920, n).astype(np.float32), rng.uniform(0, 1080, n).astype(np.float32)
    dx, dy = rng.uniform(0, 1920, m).astype(np.float32), rng.uniform(0, 1080, m).astype(np.float32)
    mt = rng.integers(-1, m, n).astype(np.int32)
    got = nm.align_points(*[torch.from_numpy(a).to(cuda) for a in (sx, sy, dx, dy, mt)])
    for gte, ref in zip(got, oracle.align_points(sx, sy, dx, dy, mt)):
        assert np.array_equal(gte.cpu().numpy(), ref)


@pytest.mark.gpu
def test_bgra_frame_to_matches_end_to_end(nm, oracle, cuda):
    """Camera-style input: BGRA uint8 frame pair -> gray -> detect/describe -> match -> aligned coordinates."""
    import torch
    import helpers as H
    w, h = 320, 240
    frames = []
    for s in (40, 41):
        g = np.clip(H.blurred_frame(s, w, h, sigma=3.0) * 1.6, 0, 255).astype(np.uint8)
        frames.append(np.stack([g, g, g, np.full_like(g, 255)], -1))
    arenas, refs = [], []
    for b in frames:
        gray = nm.grayscale(torch.from_numpy(b).to(cuda))
        a = nm.SiftArena(w, h, 4096)
        a.detect_describe(gray)
        arenas.append(a)
        refs.append(oracle.sift_detect_describe(oracle.grayscale(b), 4096))
    torch.cuda.synchronize()
    n0, n1 = int(arenas[0].num_items.item()), int(arenas[1].num_items.item())
    assert (n0, n1) == (refs[0]["n"], refs[1]["n"]) and n0 > 50
    res, _ = nm.sift_match(arenas[0].desc, arenas[1].desc, 0.8, nA=n0, nB=n1)
    ref, _, _ = oracle.sift_matches(refs[0]["desc"], refs[1]["desc"], 0.8, want_distance=False)
    assert np.array_equal(res.cpu().numpy(), ref)
    got = nm.align_points(arenas[0].x[:n0].contiguous(), arenas[0].y[:n0].contiguous(), arenas[1].x[:n1].contiguous(),
                          arenas[1].y[:n1].contiguous(), res)
    want = oracle.align_points(refs[0]["x"], refs[0]["y"], refs[1]["x"], refs[1]["y"], ref)
    for a, b in zip(got, want):
        assert np.array_equal(a.cpu().numpy(), b)
