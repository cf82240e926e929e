"""GPU parity of the fused per-octave launchers behind the drop-in C++ API (compute_dog / compute_gradients /
compute_keypoints[_with_mask] / compute_orientations / compute_descriptors in one launch each) and of the reference-style
client loop at the bench's 1080p size. Bit-exact against the oracle and against the single-level launchers."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from test_gpu_stages import _eq, _t

pytestmark = pytest.mark.gpu


def _ptrs(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def _octave(oracle, w, h, seed):
    p = oracle.sift_params(w, h)
    base = oracle.convolve(H.blurred_frame(seed, w, h), *oracle.create_kernel_for_sigma(p.base_smooth))[0]
    levels, dogs, grad = oracle.octave_pyramid(base, w, h)
    return p, levels, dogs, grad


@pytest.mark.parametrize("wh", [(320, 240), (250, 187), (1000, 60), (37, 41)])
def test_find_keypoints3_and_compact3(nm, oracle, cuda, wh):
    import torch
    w, h = wh
    p, levels, dogs, grad = _octave(oracle, w, h, 3)
    xper = 2.0
    mask = np.ones((2 * h + 1, 2 * w), np.float32)
    mask[: h // 2, :] = 0.0
    mask[:, -w // 3:] = 0.5
    tdog = [_t(d, cuda) for d in dogs]
    for m in (None, mask):
        ref = [oracle.find_keypoints(dogs[l + 1], dogs[l], dogs[l + 2], p.peak_threshold, p.edge_threshold, xper, p.sigma_0,
                                     3, l, mask=m) for l in range(3)]
        # maps start full of garbage: the fused launcher must write every pixel of the region itself
        dense = [torch.full((h + 2, w, 4), 7.0, dtype=torch.float32, device=cuda) for _ in range(3)]
        tm = _t(m, cuda) if m is not None else None
        rc = nm.lib().nm_find_keypoints3_f32(_ptrs(tdog), tm.data_ptr() if tm is not None else None,
                                             m.shape[1] if m is not None else 0, m.shape[0] if m is not None else 0, w, h,
                                             p.peak_threshold, p.edge_threshold, xper, p.sigma_0, 3, _ptrs(dense), None)
        assert rc == 0
        torch.cuda.synchronize()
        for l in range(3):
            _eq(dense[l][:h], ref[l], "dense map level %d mask=%s" % (l, m is not None))
            assert bool((dense[l][h:] == 7.0).all()), "wrote past the region"
        # batched compaction of the three maps
        n = w * h
        out = [torch.full((n, 4), -1.0, dtype=torch.float32, device=cuda) for _ in range(3)]
        cnt = torch.zeros(3, dtype=torch.int32, device=cuda)
        ws = torch.empty(nm.lib().nm_compact3_workspace_bytes(n) + 16, dtype=torch.uint8, device=cuda)
        assert nm.lib().nm_compact_keypoints3(_ptrs(dense), n, _ptrs(out), cnt.data_ptr(), ws.data_ptr(), None) == 0
        torch.cuda.synchronize()
        for l in range(3):
            want = oracle.compact_keypoints(ref[l])
            assert int(cnt[l]) == len(want)
            _eq(out[l][: len(want)], want, "collated level %d" % l)
        # the detection launch that also resets what lies behind the region (compute_keypoints' per-octave reset): entries
        # [w h, reset_end) become -1, everything behind reset_end stays
        dense2 = [torch.full((h + 40, w, 4), 7.0, dtype=torch.float32, device=cuda) for _ in range(3)]
        ends = (C.c_size_t * 3)(w * h + 5, 0, (h + 33) * w + 1)
        rc = nm.lib().nm_find_keypoints3_reset_f32(_ptrs(tdog), tm.data_ptr() if tm is not None else None,
                                                   m.shape[1] if m is not None else 0, m.shape[0] if m is not None else 0, w, h,
                                                   p.peak_threshold, p.edge_threshold, xper, p.sigma_0, 3, _ptrs(dense2), ends, None)
        assert rc == 0
        torch.cuda.synchronize()
        for l in range(3):
            _eq(dense2[l][:h], ref[l], "dense map level %d with reset" % l)
            flat = dense2[l].reshape(-1, 4)
            end = max(int(ends[l]), w * h)
            assert bool((flat[w * h:end] == -1.0).all()) and bool((flat[end:] == 7.0).all()), l


def test_dog_and_gradient_batches(nm, oracle, cuda):
    import torch
    w, h = 270, 135
    planes = [H.blurred_frame(s, w, h) for s in range(6)]
    tp = [_t(x, cuda) for x in planes]
    out = [torch.empty_like(tp[0]) for _ in range(5)]
    assert nm.lib().nm_subtract_batch_f32(5, _ptrs(tp[1:]), _ptrs(tp[:5]), _ptrs(out), w, h, None) == 0
    g = [torch.empty((h, w, 2), dtype=torch.float32, device=cuda) for _ in range(3)]
    assert nm.lib().nm_gradient_batch_f32(3, _ptrs(tp[1:4]), _ptrs(g), w, h, None) == 0
    torch.cuda.synchronize()
    for i in range(5):
        _eq(out[i], oracle.subtract(planes[i + 1], planes[i]), "dog %d" % i)
    for i in range(3):
        _eq(g[i], oracle.gradient(planes[i + 1]), "gradient %d" % i)


def test_orientation_and_descriptor_level_batches(nm, oracle, cuda):
    import torch
    w, h = 320, 240
    p, levels, dogs, grad = _octave(oracle, w, h, 5)
    lists = [oracle.compact_keypoints(oracle.find_keypoints(dogs[l + 1], dogs[l], dogs[l + 2], 0.0, 10.0, 1.0, p.sigma_0, 3, l))
             for l in range(3)]
    assert all(len(k) > 0 for k in lists)
    tg = _t(grad, cuda)
    tk = [_t(k, cuda) for k in lists]
    n = (C.c_int * 3)(*[len(k) for k in lists])
    ori = [torch.full((len(k), 2), 9.0, dtype=torch.float32, device=cuda) for k in lists]
    assert nm.lib().nm_detect_orientations_levels(3, _ptrs(tk), n, tg.data_ptr(), w, h, 1.5, 1.0, _ptrs(ori), None) == 0
    desc = [torch.zeros((len(k), 128), dtype=torch.float32, device=cuda) for k in lists]
    xs = [torch.zeros(len(k), dtype=torch.float32, device=cuda) for k in lists]
    ys = [torch.zeros(len(k), dtype=torch.float32, device=cuda) for k in lists]
    assert nm.lib().nm_compute_sift_descriptors_levels(3, _ptrs(tk), _ptrs(ori), n, tg.data_ptr(), w, h, 3, 1.0, _ptrs(desc),
                                                       _ptrs(xs), _ptrs(ys), None) == 0
    torch.cuda.synchronize()
    for l in range(3):
        ro = oracle.detect_orientations(lists[l], grad, w, h, 1.5, 1.0)
        _eq(ori[l], ro, "orientations level %d" % l)
        rd, rx, ry = oracle.compute_sift_descriptors(lists[l], ro, grad, w, h, 3, 1.0)
        _eq(desc[l], rd, "descriptors level %d" % l)
        _eq(xs[l], rx, "x level %d" % l)


def test_cpp_api_client_loop_1080p(nm, oracle, cuda):
    """The reference-style client loop (SiftParams / PyramidData / SiftData + the per-octave compute_* calls) on the
    bench's 1080p frame: every descriptor equals the oracle's."""
    frame = H.blurred_frame(0, 1920, 1080)
    cap = 16384
    ref = oracle.sift_detect_describe(frame, cap)
    desc = np.zeros((cap, 128), np.float32)
    x = np.zeros(cap, np.float32)
    y = np.zeros(cap, np.float32)
    n = nm.lib().nm_client_detect_describe(frame.ctypes.data, 1920, 1080, cap, desc.ctypes.data, x.ctypes.data,
                                           y.ctypes.data)
    assert n == ref["n"] and n > 10000
    _eq(desc[:n], ref["desc"], "C++ API descriptors at 1080p")
    _eq(x[:n], ref["x"], "x")
    _eq(y[:n], ref["y"], "y")


def test_cpp_api_pair_loop_is_stateless_and_matches(nm, oracle, cuda):
    """nm_client_pair_loop re-uses one PyramidData / two SiftData over many frames (dirty-region bookkeeping of the dense
    maps, grow-only workspaces): after several repetitions the counts and the matches are still the oracle's."""
    import torch
    import bench
    f = bench.make_frames(nm, torch, cuda, [0, 1])
    r0 = oracle.sift_detect_describe(H.blurred_frame(0, 1920, 1080), 16384)
    r1 = oracle.sift_detect_describe(H.blurred_frame(1, 1920, 1080), 16384)
    m, _, _ = oracle.sift_matches(r0["desc"], r1["desc"], 0.8, want_distance=False)
    for wd in (0, 1):
        out = (C.c_int * 3)()
        us = nm.lib().nm_client_pair_loop(f[0].data_ptr(), f[1].data_ptr(), 1920, 1080, 16384, 3, wd, out)
        assert us > 0
        assert (out[0], out[1], out[2]) == (r0["n"], r1["n"], int((m >= 0).sum()))


def test_pyramiddata_and_siftdata_value_semantics(nm, oracle, cuda):
    """PyramidData owned a raw pinned pointer with implicit copy operations (double hipHostFree, writes into freed
    pinned memory). Copies, assignments, moves and std::vector growth now behave like the reference's thrust-based members
    (sift/pyramidata.h:60-110, sift/siftdata.h:25-40): every variant runs the frame and must reproduce the oracle's count."""
    w, h, cap = 320, 240, 4096
    f = H.blurred_frame(5, w, h)
    ref = oracle.sift_detect_describe(f, cap)
    n = nm.lib().nm_client_copy_semantics(np.ascontiguousarray(f).ctypes.data, w, h, cap)
    assert n == ref["n"] and n > 100


@pytest.mark.parametrize("wh,cap", [((640, 480), 8192), ((320, 240), 300), ((1920, 1080), 16384)])
def test_lazy_counts_show_the_reference_observable_state(nm, oracle, cuda, wh, cap):
    """nm/lazy_count.h: the per-octave client loop no longer synchronises per octave -- the keypoint counts stay on the device
    and _orientations[l].size() / SiftData::_num_items resolve at their first host-visible read (the reference reads them back
    once per level, sift/pyramidata.cu:84-91, siftfunctions.cu:165-178). A client that looks after every call must see the
    numbers the eager path (one synchronisation per octave) shows, a client that never looks must end with the same
    descriptors, and both must equal the oracle; cap = 300 clips in the middle of a level ON THE DEVICE."""
    w, h = wh
    f = np.ascontiguousarray(H.blurred_frame(7, w, h))
    ref = oracle.sift_detect_describe(f, cap)
    n_oct = ref["counts"].shape[0]
    watch = (C.c_int * (4 * n_oct))()
    pending = C.c_int(0)
    n = nm.lib().nm_client_lazy_counts(f.ctypes.data, w, h, cap, watch, n_oct, C.byref(pending))
    assert n == ref["n"] and (n == cap or cap > 5000)
    assert pending.value >= 4 * n_oct, "the lazy path was not taken"      # 3 sizes + 1 item count per octave
    # what the curious client saw: the level sizes are the oracle's accepted counts per (octave, level) -- an empty level ends
    # its octave -- and the running item count is their clipped running sum
    run = 0
    for o in range(n_oct):
        cnt = [int(c) for c in ref["counts"][o]]
        for l in range(3):
            assert watch[4 * o + l] == cnt[l], (o, l)
        run = min(cap, run + sum(cnt))
        assert watch[4 * o + 3] == run, o
