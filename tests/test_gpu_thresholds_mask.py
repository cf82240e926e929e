"""The reference's run-time knobs on the HIP path with NON-default values: SiftParams::_peak_threshold / _edge_threshold
(public fields, sift/siftparams.h:97-98, passed per call, siftfunctions.cu:123-125) and the mask variant
(siftfunctions.cu:65-98, keypoint.cu:204-224). With peak != 0 both branches of the double gate `c >= 0.8 peak` /
`c <= 0.8 peak` (keypoint.cu:195-196) are live and |v| > peak (keypoint.cu:168) rejects refined points; edge moves the
(e+1)^2/e test (:169). Frame driver (nm_sift_arena_set_params / _set_mask + nm_sift_detect_describe[_batch]) and the fused
per-octave launcher nm_find_keypoints3_f32, on the 640x480 and 1080p bench frames, bit-exact against the oracle."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from test_gpu_stages import _eq, _t

pytestmark = pytest.mark.gpu

COMBOS = [(0.5, 5.0), (0.5, 20.0), (2.0, 5.0), (2.0, 20.0), (-0.25, 10.0)]


def _mask(w, h):
    """ROI mask with soft edges: 1 inside an off-centre rectangle, a 0.5 band (rejected: the fetch must be >= 1) and 0."""
    m = np.zeros((h, w), np.float32)
    m[h // 5: 4 * h // 5, w // 4: 7 * w // 8] = 1.0
    m[h // 2: h // 2 + 9, :] = 0.5
    m[:, w // 2: w // 2 + 3] = 0.0
    return m


def _check(a, ref, what):
    n = int(a.num_items.item())
    assert n == ref["n"], (what, n, ref["n"])
    _eq(a.kpts[:n], ref["kpts"], "keypoints " + what)
    _eq(a.orients[:n], ref["orient"], "orientations " + what)
    _eq(a.desc[:n], ref["desc"], "descriptors " + what)
    _eq(a.x[:n], ref["x"], "x " + what)


@pytest.mark.parametrize("wh,cap", [((640, 480), 8192), ((1920, 1080), 16384)])
def test_frame_driver_thresholds_and_mask(nm, oracle, cuda, wh, cap):
    import torch
    w, h = wh
    f = H.blurred_frame(0, w, h)
    d = _t(f, cuda)
    base = oracle.sift_detect_describe(f, cap)
    a = nm.SiftArena(w, h, cap, device=cuda)
    counts = {}
    for peak, edge in COMBOS:
        ref = oracle.sift_detect_describe(f, cap, peak=peak, edge=edge)
        a.set_params(peak, edge)
        a.detect_describe(d)
        torch.cuda.synchronize()
        _check(a, ref, "peak=%g edge=%g %dx%d" % (peak, edge, w, h))
        counts[(peak, edge)] = ref["n"]
    # the knobs bite, in the direction the reference's tests imply
    assert 0 < counts[(2.0, 5.0)] < counts[(0.5, 5.0)] < base["n"]
    assert counts[(0.5, 5.0)] < counts[(0.5, 20.0)] and counts[(2.0, 5.0)] <= counts[(2.0, 20.0)]
    # mask, alone and together with non-default thresholds
    m = _mask(w, h)
    tm = _t(m, cuda)
    for peak, edge in ((0.0, 10.0), (0.5, 20.0)):
        ref = oracle.sift_detect_describe(f, cap, peak=peak, edge=edge, mask=m)
        a.set_params(peak, edge)
        a.set_mask(tm)
        a.detect_describe(d)
        torch.cuda.synchronize()
        _check(a, ref, "masked peak=%g edge=%g" % (peak, edge))
        assert 0 < ref["n"] < counts.get((peak, edge), base["n"])
        k = ref["kpts"]
        # the mask is tested at the integer pixel of its octave, the keypoint is reported at its refined position: nearly
        # all (not all) reported positions lie inside the ROI
        assert (m[np.clip(k[:, 1].astype(int), 0, h - 1), np.clip(k[:, 0].astype(int), 0, w - 1)] >= 0.5).mean() > 0.9
    # removing the mask and restoring the defaults gives the default result again
    a.set_mask(None)
    a.set_params()
    a.detect_describe(d)
    torch.cuda.synchronize()
    _check(a, base, "defaults restored")
    a.close()


def test_batched_call_masks_per_frame_and_params_must_agree(nm, oracle, cuda):
    import torch
    w, h, cap = 640, 480, 8192
    fr = [H.blurred_frame(s, w, h) for s in (0, 1, 2)]
    dev = [_t(f, cuda) for f in fr]
    arenas = [nm.SiftArena(w, h, cap, device=cuda) for _ in fr]
    m = _mask(w, h)
    tm = _t(m, cuda)
    for a in arenas:
        a.set_params(0.5, 20.0)
    arenas[1].set_mask(tm)                              # only the middle frame is masked
    nm.detect_describe_batch(arenas, dev)
    torch.cuda.synchronize()
    for k, a in enumerate(arenas):
        ref = oracle.sift_detect_describe(fr[k], cap, peak=0.5, edge=20.0, mask=m if k == 1 else None)
        _check(a, ref, "batched frame %d" % k)
    arenas[2].set_params(0.5, 10.0)                     # thresholds are launch arguments: one pair per call
    with pytest.raises(nm.NmError):
        nm.detect_describe_batch(arenas, dev)
    with pytest.raises(nm.NmError):
        arenas[0].set_params(0.0, 0.0)                  # edge threshold must be positive
    with pytest.raises(nm.NmError):
        arenas[0].set_mask(torch.zeros((h, w + 1), dtype=torch.float32, device=cuda))
    for a in arenas:
        a.close()


@pytest.mark.parametrize("wh", [(640, 480), (1920, 1080)])
def test_find_keypoints3_nondefault_thresholds(nm, oracle, cuda, wh):
    """nm_find_keypoints3_f32 (what compute_keypoints[_with_mask] of the drop-in C++ API launch) on octave 0 of the bench
    frame: every level, every threshold pair, with and without the mask."""
    import torch
    w, h = wh
    p = oracle.sift_params(w, h)
    base = oracle.convolve(H.blurred_frame(0, w, h), *oracle.create_kernel_for_sigma(p.base_smooth))[0]
    levels, dogs, grad = oracle.octave_pyramid(base, w, h, want_grad=False)
    tdog = [_t(x, cuda) for x in dogs]
    ptrs = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    m = _mask(w, h)
    tm = _t(m, cuda)
    for peak, edge in COMBOS[:4]:
        for mask in (None, m):
            ref = [oracle.find_keypoints(dogs[l + 1], dogs[l], dogs[l + 2], peak, edge, 1.0, p.sigma_0, 3, l, mask=mask)
                   for l in range(3)]
            dense = [torch.full((h, w, 4), 7.0, dtype=torch.float32, device=cuda) for _ in range(3)]
            rc = nm.lib().nm_find_keypoints3_f32(ptrs(tdog), tm.data_ptr() if mask is not None else None,
                                                 w if mask is not None else 0, h if mask is not None else 0, w, h, peak,
                                                 edge, 1.0, p.sigma_0, 3, ptrs(dense), None)
            assert rc == 0
            torch.cuda.synchronize()
            for l in range(3):
                _eq(dense[l], ref[l], "dense map level %d peak=%g edge=%g mask=%s" % (l, peak, edge, mask is not None))
        # the single-level launcher under the same thresholds
        got = nm.find_keypoints(tdog[2], tdog[1], tdog[3], peak, edge, 1.0, p.sigma_0, 3, 1)
        _eq(got, oracle.find_keypoints(dogs[2], dogs[1], dogs[3], peak, edge, 1.0, p.sigma_0, 3, 1), "single level")
