"""GPU parity, stage by stage: every C-ABI launcher against the CPU oracle on the same seeded inputs. Bit-exact."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _t(a, cuda):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _eq(got, ref, what):
    got = got.detach().cpu().numpy() if hasattr(got, "detach") else np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    same = got.view(np.uint32) == ref.view(np.uint32) if got.dtype == np.float32 else got == ref
    assert same.all(), "%s: %d of %d elements differ (max abs diff %g)" % (
        what, (~same).sum(), same.size, np.nanmax(np.abs(got.astype(np.float64) - ref.astype(np.float64))))


@pytest.mark.parametrize("wh", [(200, 150), (640, 480), (60, 33), (135, 67), (61, 45), (3, 2)])
def test_convolve_all_sift_kernels(nm, oracle, cuda, wh):
    w, h = wh
    img = H.synth.noise_frame(7, w, h)
    p = oracle.sift_params(1920, 1080)
    sigmas = [p.base_smooth] + list(p.sigmas)[:5] + [3.0, 4.0, 0.6]     # radii 7,5,7,8,10,13,12,16 + generic 3
    for s in sigmas:
        taps, r = oracle.create_kernel_for_sigma(s)
        t2, r2 = nm.create_kernel_for_sigma(s)
        assert r2 == r and np.array_equal(t2, taps)
        ref, refbuf = oracle.convolve(img, taps, r)
        out, buf = nm.convolve(_t(img, cuda), _t(taps, cuda), r, want_buffer=True)
        _eq(out, ref, "convolve sigma=%g" % s)
        _eq(buf, refbuf, "convolve row pass sigma=%g" % s)


def test_downsample_subtract_gradient(nm, oracle, cuda):
    a = H.blurred_frame(1, 270, 135)
    b = H.blurred_frame(2, 270, 135)
    _eq(nm.downsample2(_t(a, cuda), 135, 67), oracle.downsample2(a, 135, 67), "downsample")
    _eq(nm.subtract(_t(a, cuda), _t(b, cuda)), oracle.subtract(a, b), "subtract")
    _eq(nm.gradient(_t(a, cuda)), oracle.gradient(a), "gradient")
    ramp = np.tile(np.arange(64, dtype=np.float32) * 2.0, (48, 1))       # dy = 0, dx > 0: theta = (float)(2 pi)
    g = nm.gradient(_t(ramp, cuda)).cpu().numpy()
    _eq(g, oracle.gradient(ramp), "gradient ramp")
    assert g[10, 10, 1] == np.float32(2 * np.pi) and g[0, 0, 0] == 0 and g[0, 0, 1] == 0
    flat = np.full((40, 40), 3.0, np.float32)
    assert not nm.gradient(_t(flat, cuda)).cpu().numpy().any()


def test_fast_sqrt_is_correctly_rounded_everywhere(nm):
    """The packed Gaussian kernel's gradient uses rsq + two exact-residual corrections instead of the IEEE expansion:
    exhaustive comparison over every float in [2^-96, 2^96) and 0."""
    assert nm.selftest_sqrt() == 0


def test_descriptor_weight_fast_form_equals_the_spec_sequence_everywhere(nm):
    """The descriptor kernel evaluates (float)exp(t / 8) on voting samples by a division-free form (ln 2 reduction + Taylor
    degree 10) and falls back to the spec's binary64 sequence when a binary32 rounding boundary is near: every float t in
    [0, 12.875] is compared on the device."""
    bad, near, n = nm.selftest_expw()
    assert bad == 0
    assert n == int(np.float32(12.875).view(np.uint32)) + 1
    assert near < n / 2 ** 13            # the fallback stays rare (2^-15 expected)


def test_orientation_hoisted_arithmetic_equals_the_expressions_it_replaces(nm):
    """frame_orient_kernel divides by 3 in binary32 (exact-residual form), tests the window in binary32 and divides r2 by a
    keypoint's 2 sigma^2 through a once-refined reciprocal: all 2^32 floats for the two one-operand forms (every window radius),
    2^32 pseudo-random pairs of its guarded domain for the division, each against the plain expression on the device."""
    bad3, rejected, badc, badd, nd = nm.selftest_orient()
    assert bad3 == 0 and badc == 0 and badd == 0
    assert rejected == 2 ** 24 + 1          # NaNs (2^24 - 2), the two infinities, -0: these take the binary64 expression
    assert nd == 8192 * 256 * 2048


def _octave(oracle, w, h, seed):
    lv0 = H.blurred_frame(seed, w, h, sigma=2.0)
    return oracle.octave_pyramid(lv0, 1920, 1080)


def _extreme_level0(w, h):
    """Level-0 plane whose differences span the whole float range: decades of decay down to denormals (the gradient's
    fast sqrt / division leave their proven domain there and must fall back), exact zeros, a huge plateau, a negative
    zero, and isolated spikes next to flat areas (dx = 0 or dy = 0 exactly)."""
    yy, xx = np.mgrid[0:h, 0:w]
    img = (200.0 * np.power(10.0, -(xx + 0.7 * yy) / 6.0)).astype(np.float32)        # reaches 0 via denormals
    img[h // 2:, : w // 3] = 0.0
    img[: h // 4, w // 2: w // 2 + 40] = np.float32(3e30)
    img[h // 4: h // 4 + 3, w // 2: w // 2 + 40] = np.float32(1e-30)
    img[5, 5] = -0.0
    img[h - 20, w - 20] = 77.0
    img[h - 40: h - 30, w - 60: w - 50] = np.float32(1e-20)
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("kind", ["blurred", "extreme"])
def test_octave_pyramid_fused(nm, oracle, cuda, kind):
    import torch
    w, h = 320, 200
    if kind == "blurred":
        levels, dogs, grad = _octave(oracle, w, h, 3)
    else:
        levels, dogs, grad = oracle.octave_pyramid(_extreme_level0(w, h), 1920, 1080)
        g = np.asarray(grad)
        assert ((g[..., 0] > 0) & (g[..., 0] < 1e-16)).any() and (g[..., 0] > 1e20).any()      # both fallback ends occur
    arena = nm.SiftArena(w, h, 1024)
    n = w * h

    def view(ptr, count):
        # wrap arena memory for read-back
        buf = torch.empty(count, dtype=torch.float32, device=cuda)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        assert hip.hipMemcpy(buf.data_ptr(), ptr, count * 4, 3) == 0
        return buf

    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    lv0 = _t(levels[0], cuda)
    assert hip.hipMemcpy(arena.level_ptr(0), lv0.data_ptr(), n * 4, 3) == 0
    arena.octave_pyramid(w, h)
    torch.cuda.synchronize()
    for l in range(1, 6):
        _eq(view(arena.level_ptr(l), n).reshape(h, w), levels[l], "level %d" % l)
    for d in range(5):
        _eq(view(arena.dog_ptr(d), n).reshape(h, w), dogs[d], "dog %d" % d)
    _eq(view(arena.grad_ptr(), 6 * n).reshape(3, h, w, 2), grad, "grad")
    arena.close()


@pytest.mark.parametrize("wh", [(320, 200), (135, 67)])
def test_keypoints_orientations_descriptors(nm, oracle, cuda, wh):
    w, h = wh
    levels, dogs, grad = _octave(oracle, w, h, 5)
    p = oracle.sift_params(1920, 1080)
    xper = 2.0
    total = 0
    for lvl in range(3):
        ref = oracle.find_keypoints(dogs[lvl + 1], dogs[lvl], dogs[lvl + 2], 0.0, 10.0, xper, p.sigma_0, 3, lvl)
        got = nm.find_keypoints(_t(dogs[lvl + 1], cuda), _t(dogs[lvl], cuda), _t(dogs[lvl + 2], cuda), 0.0, 10.0, xper,
                                p.sigma_0, 3, lvl)
        _eq(got, ref, "dense keypoint map level %d" % lvl)
        kp_ref = oracle.compact_keypoints(ref)
        kp = nm.compact_keypoints(got)
        _eq(kp, kp_ref, "compaction level %d" % lvl)
        total += len(kp_ref)
        if len(kp_ref) == 0:
            continue
        ori_ref = oracle.detect_orientations(kp_ref, grad, w, h, 1.5, xper)
        ori = nm.detect_orientations(kp, _t(grad, cuda), w, h, 1.5, xper)
        _eq(ori, ori_ref, "orientations level %d" % lvl)
        d_ref, x_ref, y_ref = oracle.compute_sift_descriptors(kp_ref, ori_ref, grad, w, h, 3, xper)
        d, x, y = nm.compute_sift_descriptors(kp, ori, _t(grad, cuda), w, h, 3, xper)
        _eq(d, d_ref, "descriptors level %d" % lvl)
        _eq(x, x_ref, "x")
        _eq(y, y_ref, "y")
    assert total > 20


def test_orientations_outside_the_hoisted_forms_domain(nm, oracle, cuda):
    """The orientation kernel runs a keypoint's votes in their hoisted form only inside a guarded domain (2 sigma_w^2 in
    [2^-20, 2^20], coordinates 0 or >= 2^-50 and below 2^22, quotient below 87) and the 3-tap mean's binary32 division only for
    finite sums: keypoints and gradients outside either -- tiny and huge scales, a denormal-range coordinate, infinite
    magnitudes, orientations beyond 2 pi (bins 37.. take `%`) and exactly 2 pi (bin 36 -> 0) -- must still equal the oracle bit
    for bit, beside ordinary keypoints on the same planes."""
    w, h = 320, 200
    levels, dogs, grad = oracle.octave_pyramid(_extreme_level0(w, h), 1920, 1080)
    grad = np.array(grad, dtype=np.float32, copy=True)
    assert np.isinf(grad[..., 0]).any()                       # the 3e30 plateau's edge: dx^2 overflows
    two_pi = np.float32(2 * 3.14159265358979323846)
    grad[0, 60:70, 100:130, 1] = two_pi                       # bin 36
    grad[0, 70:80, 100:130, 1] = np.float32(7.5)              # bin 42 -> 6 through the generic remainder
    grad[0, 60:80, 100:130, 0] = np.float32(3.0)
    xper = 2.0
    rng = np.random.default_rng(11)
    rows = []
    for _ in range(200):                                      # ordinary keypoints all over the plane (borders included)
        rows.append((rng.uniform(0, w - 1) * xper, rng.uniform(0, h - 1) * xper, rng.uniform(1.0, 4.5) * xper, 0.0))
    for x, y in ((110.3, 66.2), (120.9, 74.5), (105.0, 79.0), (162.0, 12.0), (170.5, 52.0)):
        for sc in (1e-4, 3e-4, 1.6, 2.7, 1e5, 1e6):          # tiny / ordinary / huge scales on the special regions
            rows.append((x * xper, y * xper, sc * xper, 0.0))
    rows.append((1e-30 * xper, 40.0 * xper, 2.0 * xper, 0.0))     # denormal-range coordinates
    rows.append((55.0 * xper, 3e-25 * xper, 2.0 * xper, 0.0))
    rows.append((0.0, 0.0, 2.0 * xper, 0.0))
    kp = np.asarray(rows, dtype=np.float32)
    ref = oracle.detect_orientations(kp, grad, w, h, 1.5, xper)
    got = nm.detect_orientations(_t(kp, cuda), _t(grad, cuda), w, h, 1.5, xper)
    _eq(got, ref, "orientations outside the hoisted forms' domain")
    assert (np.asarray(ref)[:, 0] >= 0).sum() > 50


def test_masked_keypoints(nm, oracle, cuda):
    w, h = 160, 120
    levels, dogs, grad = _octave(oracle, w, h, 9)
    p = oracle.sift_params(640, 480)
    for xper, mw, mh in ((1.0, w, h), (2.0, 2 * w, 2 * h)):
        mask = np.zeros((mh, mw), np.float32)
        mask[mh // 4: 3 * mh // 4, mw // 3:] = 1.0
        ref = oracle.find_keypoints(dogs[2], dogs[1], dogs[3], 0.0, 10.0, xper, p.sigma_0, 3, 1, mask=mask)
        got = nm.find_keypoints(_t(dogs[2], cuda), _t(dogs[1], cuda), _t(dogs[3], cuda), 0.0, 10.0, xper, p.sigma_0, 3,
                                1, mask=_t(mask, cuda))
        _eq(got, ref, "masked map xper=%g" % xper)
        full = oracle.find_keypoints(dogs[2], dogs[1], dogs[3], 0.0, 10.0, xper, p.sigma_0, 3, 1)
        assert 0 < (ref[..., 3] >= 0).sum() < (full[..., 3] >= 0).sum()
