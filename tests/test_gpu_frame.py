"""GPU parity of the whole detect/describe path: frame driver (C ABI) and the C++ API client loop vs the oracle."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from test_gpu_stages import _eq, _t

pytestmark = pytest.mark.gpu


def _run_arena(nm, cuda, frame, capacity):
    import torch
    h, w = frame.shape
    arena = nm.SiftArena(w, h, capacity)
    arena.detect_describe(_t(frame, cuda))
    torch.cuda.synchronize()
    n = int(arena.num_items.item())
    out = dict(n=n, desc=arena.desc[:n].cpu().numpy(), x=arena.x[:n].cpu().numpy(), y=arena.y[:n].cpu().numpy(),
               kpts=arena.kpts[:n].cpu().numpy(), orient=arena.orients[:n].cpu().numpy())
    arena.close()
    return out


@pytest.mark.parametrize("wh,cap", [((128, 96), 2048), ((250, 187), 4096), ((37, 41), 256), ((640, 480), 16384),
                                    ((1920, 1080), 16384)])
def test_frame_driver_matches_oracle(nm, oracle, cuda, wh, cap):
    w, h = wh
    frame = H.blurred_frame(0, w, h)
    ref = oracle.sift_detect_describe(frame, cap)
    got = _run_arena(nm, cuda, frame, cap)
    assert got["n"] == ref["n"]
    _eq(got["kpts"], ref["kpts"], "keypoints (x,y,sigma,level), output order")
    _eq(got["orient"], ref["orient"], "orientations")
    _eq(got["x"], ref["x"], "x")
    _eq(got["y"], ref["y"], "y")
    _eq(got["desc"], ref["desc"], "descriptors")


def test_frame_driver_capacity_clipping(nm, oracle, cuda):
    frame = H.blurred_frame(3, 320, 240)
    full = oracle.sift_detect_describe(frame, 16384)
    cap = full["n"] // 2 + 3
    ref = oracle.sift_detect_describe(frame, cap)
    got = _run_arena(nm, cuda, frame, cap)
    assert ref["n"] == cap == got["n"]
    _eq(got["desc"], ref["desc"], "descriptors under capacity (Q13)")
    _eq(got["desc"], full["desc"][:cap], "prefix property")


def test_frame_driver_reuse_is_stateless(nm, oracle, cuda):
    import torch
    a = H.blurred_frame(11, 256, 192)
    b = H.blurred_frame(12, 256, 192)
    arena = nm.SiftArena(256, 192, 4096)
    outs = []
    for f in (a, b, a):
        arena.detect_describe(_t(f, cuda))
        torch.cuda.synchronize()
        n = int(arena.num_items.item())
        outs.append((n, arena.desc[:n].cpu().numpy().copy()))
    arena.close()
    assert outs[0][0] == outs[2][0] and np.array_equal(outs[0][1], outs[2][1])
    ref = oracle.sift_detect_describe(b, 4096)
    assert outs[1][0] == ref["n"]
    _eq(outs[1][1], ref["desc"], "second frame after reuse")


@pytest.mark.parametrize("wh,n", [((256, 192), 2), ((250, 187), 3), ((1920, 1080), 2), ((320, 240), 4)])
def test_frame_batch_equals_single_frames(nm, oracle, cuda, wh, n):
    """nm_sift_detect_describe_batch: one launch sequence for n frames; every frame's outputs equal the oracle's (and
    therefore the single-frame driver's). Includes a width that is not a multiple of 4 (per-frame fallback kernels)."""
    import torch
    w, h = wh
    frames = [H.blurred_frame(20 + i, w, h) for i in range(n)]
    arenas = [nm.SiftArena(w, h, 16384) for _ in range(n)]
    for rep in range(2):                                   # second pass: arenas and side streams are reused
        order = list(range(n)) if rep == 0 else list(reversed(range(n)))
        nm.detect_describe_batch(arenas, [_t(frames[i], cuda) for i in order])
        torch.cuda.synchronize()
        for a, i in zip(arenas, order):
            ref = oracle.sift_detect_describe(frames[i], 16384) if (w * h <= 320 * 240 or i == 0) else None
            cnt = int(a.num_items.item())
            if ref is not None:
                assert cnt == ref["n"]
                _eq(a.kpts[:cnt], ref["kpts"], "batch keypoints frame %d" % i)
                _eq(a.orients[:cnt], ref["orient"], "batch orientations frame %d" % i)
                _eq(a.desc[:cnt], ref["desc"], "batch descriptors frame %d" % i)
            else:                                           # 1080p: compare the other frames with the single-frame driver
                single = _run_arena(nm, cuda, frames[i], 16384)
                assert cnt == single["n"]
                _eq(a.desc[:cnt], single["desc"], "batch vs single descriptors frame %d" % i)
                _eq(a.x[:cnt], single["x"], "batch vs single x frame %d" % i)
    for a in arenas:
        a.close()


@pytest.mark.parametrize("wh,n", [((250, 187), 3), ((320, 240), 4), ((37, 41), 2), ((1920, 1080), 2), ((523, 21), 2)])
def test_tall_detection_groups_give_the_same_frames(nm, oracle, cuda, wh, n):
    """Round 5: batched detection launches with thousands of unit groups take 20-row groups instead of 5-row ones
    (nm_sift_set_detect_tall_min). Forced on here for every launch: heights that 20 does not divide, planes lower than one
    group, widths that 256 does not divide -- the same keypoints, orientations and descriptors as the oracle's, and as the
    5-row form's bit for bit."""
    import torch
    w, h = wh
    frames = [H.blurred_frame(40 + i, w, h) for i in range(n)]
    outs = {}
    for tall_min in (1, 2 ** 31 - 1):
        prev = nm.set_detect_tall_min(tall_min)
        try:
            arenas = [nm.SiftArena(w, h, 16384) for _ in range(n)]
            nm.detect_describe_batch(arenas, [_t(f, cuda) for f in frames])
            torch.cuda.synchronize()
            outs[tall_min] = [(int(a.num_items.item()), a.kpts[:int(a.num_items.item())].cpu().numpy(),
                               a.orients[:int(a.num_items.item())].cpu().numpy(), a.desc[:int(a.num_items.item())].cpu().numpy())
                              for a in arenas]
            for a in arenas:
                a.close()
        finally:
            nm.set_detect_tall_min(prev)
    nm.set_detect_tall_min(-1)                             # the default, whatever an earlier failure left
    for i in range(n):
        t, f = outs[1][i], outs[2 ** 31 - 1][i]
        assert t[0] == f[0]
        for k, what in ((1, "keypoints"), (2, "orientations"), (3, "descriptors")):
            _eq(t[k], f[k], "%s, frame %d, 20-row against 5-row groups" % (what, i))
        if w * h <= 320 * 240 or i == 0:
            ref = oracle.sift_detect_describe(frames[i], 16384)
            assert t[0] == ref["n"]
            _eq(t[1], ref["kpts"], "keypoints frame %d, 20-row groups" % i)
            _eq(t[3], ref["desc"], "descriptors frame %d, 20-row groups" % i)


def test_frame_batch_rejects_bad_arguments(nm, cuda):
    import torch
    a, b = nm.SiftArena(64, 48, 256), nm.SiftArena(128, 96, 256)
    f64 = torch.zeros((48, 64), device=cuda)
    f128 = torch.zeros((96, 128), device=cuda)
    with pytest.raises(nm.NmError):
        nm.detect_describe_batch([a, b], [f64, f128])          # different geometry
    with pytest.raises(nm.NmError):
        nm.detect_describe_batch([a, a], [f64, f64])           # the same arena twice
    with pytest.raises(nm.NmError):
        nm.detect_describe_batch([a] * (nm.SIFT_MAX_BATCH + 1), [f64] * (nm.SIFT_MAX_BATCH + 1))   # too many
    a.close(); b.close()


def test_cpp_api_client_loop(nm, oracle, cuda):
    """The reference-style client (SiftParams/PyramidData/SiftData + compute_*) gives the same answer."""
    frame = H.blurred_frame(0, 640, 480)
    cap = 4096
    ref = oracle.sift_detect_describe(frame, cap)
    desc = np.zeros((cap, 128), np.float32)
    x = np.zeros(cap, np.float32)
    y = np.zeros(cap, np.float32)
    n = nm.lib().nm_client_detect_describe(frame.ctypes.data, 640, 480, cap, desc.ctypes.data, x.ctypes.data,
                                           y.ctypes.data)
    assert n == ref["n"]
    _eq(desc[:n], ref["desc"], "C++ API descriptors")
    _eq(x[:n], ref["x"], "C++ API x")
    _eq(y[:n], ref["y"], "C++ API y")


def test_flat_and_tiny_frames_have_no_keypoints(nm, oracle, cuda):
    import torch
    for shape in ((64, 64), (5, 4)):
        flat = np.full(shape, 37.0, np.float32)
        ref = oracle.sift_detect_describe(flat, 64)
        got = _run_arena(nm, cuda, flat, 64)
        assert got["n"] == ref["n"]
        if ref["n"]:
            _eq(got["desc"], ref["desc"], "descriptors of the zero-padding border extrema")
    z = np.zeros((48, 40), np.float32)
    assert _run_arena(nm, cuda, z, 16)["n"] == 0 == oracle.sift_detect_describe(z, 16)["n"]


def test_empty_level_ends_the_octave_q9(nm, oracle, cuda):
    """SURVEY Q9 (sift/siftfunctions.cu:145,160): keypoints of the levels after the first empty level of an octave are
    dropped. A single blob whose scale puts it in level 1 or 2 of octave 0 (level 0 empty) must vanish."""
    yy, xx = np.mgrid[0:128, 0:128].astype(np.float64)
    hits = 0
    for sb in (3.1, 3.3, 3.5, 3.7):
        img = (200.0 * np.exp(-((xx - 64.3) ** 2 + (yy - 63.6) ** 2) / (2 * sb * sb))).astype(np.float32)
        ref = oracle.sift_detect_describe(img, 64)
        got = _run_arena(nm, cuda, img, 64)
        assert got["n"] == ref["n"]
        _eq(got["kpts"], ref["kpts"], "keypoints sb=%g" % sb)
        p = oracle.sift_params(128, 128)
        levels, dogs, _ = oracle.octave_pyramid(oracle.convolve(img, *oracle.create_kernel_for_sigma(p.base_smooth))[0], 128, 128)
        per_level = [len(oracle.compact_keypoints(oracle.find_keypoints(dogs[l + 1], dogs[l], dogs[l + 2], 0.0, 10.0, 1.0,
                                                                      p.sigma_0, 3, l))) for l in range(3)]
        if per_level[0] == 0 and sum(per_level) > 0:
            hits += 1
            assert ref["counts"][0].tolist() == [0, 0, 0]      # found by the stage, dropped by the orchestration
    assert hits >= 1, "no blob exercised the empty-level rule; adjust the scales"


def test_cpp_api_match_with_and_without_distance(nm, oracle, cuda):
    """compute_sift_matches through the C++ SiftData API (nm_client_match), incl. the materialised distance matrix."""
    A = H.synth.descriptors(11, 700)
    B = H.synth.descriptors(12, 450)
    ref, Dref, _ = oracle.sift_matches(A, B, 0.8)
    res = np.full(700, -1, np.int32)
    D = np.zeros((700, 450), np.float32)
    assert nm.lib().nm_client_match(A.ctypes.data, 700, B.ctypes.data, 450, D.ctypes.data, res.ctypes.data, 0.8) == 0
    assert np.array_equal(res, ref)
    H.assert_distance(nm, D, Dref, "distance matrix via the C++ API (default: fp32 MFMA pass, 1e-4 relative)")
    before = nm.get_distance_mode()
    try:                                                  # the exact kernel behind the switch: bit for bit
        nm.set_distance_mode("exact")
        assert nm.lib().nm_client_match(A.ctypes.data, 700, B.ctypes.data, 450, D.ctypes.data, res.ctypes.data, 0.8) == 0
        assert np.array_equal(res, ref)
        _eq(D, Dref, "distance matrix via the C++ API, exact kernel")
    finally:
        nm.set_distance_mode(before)
    res2 = np.full(700, -1, np.int32)
    assert nm.lib().nm_client_match(A.ctypes.data, 700, B.ctypes.data, 450, None, res2.ctypes.data, 0.8) == 0
    assert np.array_equal(res2, ref)


@pytest.mark.parametrize("wh", [(520, 44), (256, 9), (1000, 12)])
def test_dense_candidates_take_several_refinement_passes(nm, oracle, cuda, wh):
    """detect_stage_kernel lists the candidates of a 4 x 256-pixel unit and refines them 256 at a time. Real scale spaces
    put ~10 candidates into a unit; crafted DoG planes (noise in the searched planes, almost nothing in their neighbours:
    22 % of the pixels are strict extrema) put 400-900 there, so the refinement runs 2-4 passes per unit with the accepted
    counts carried between passes and the ordered compaction inside every (row, level) group crossing pass boundaries.
    Both forms of the kernel against the oracle: the dense maps of the API and the compact ordered lists of the frame
    driver (incl. the empty-level and capacity rules of the orchestration)."""
    import torch
    w, h = wh
    rng = np.random.default_rng(w * 1000 + h)
    big = lambda: (rng.uniform(-40, 40, (h, w))).astype(np.float32)
    small = lambda: (rng.uniform(-1e-3, 1e-3, (h, w))).astype(np.float32)
    dogs = [small(), big(), small(), big(), small()]
    p = oracle.sift_params(1920, 1080)
    dense_ref = [oracle.find_keypoints(dogs[l + 1], dogs[l], dogs[l + 2], p.peak_threshold, p.edge_threshold, 2.0, p.sigma_0, 3, l)
                 for l in range(3)]
    lists_ref = [oracle.compact_keypoints(d) for d in dense_ref]
    def extrema(cur, dn, up):            # strict 26-neighbour extrema with the sign gating of keypoint.cu:195-196, interior only
        c = cur[1:-1, 1:-1]
        nb = [pl[1 + dy: pl.shape[0] - 1 + dy, 1 + dx: pl.shape[1] - 1 + dx] for pl in (cur, dn, up) for dy in (-1, 0, 1)
              for dx in (-1, 0, 1) if not (pl is cur and dy == 0 and dx == 0)]
        hi, lo = np.maximum.reduce(nb), np.minimum.reduce(nb)
        return ((c >= 0) & (c > hi)) | ((c <= 0) & (c < lo))
    cand = np.zeros((h - 2, w - 2), np.int32)
    for l in range(3):
        cand += extrema(dogs[l + 1], dogs[l], dogs[l + 2])
    per_unit = max(int(cand[r0: r0 + 4, c0: c0 + 256].sum()) for r0 in range(0, h, 4) for c0 in range(0, w, 256))
    assert per_unit > 300, "the planes are not dense enough to need several refinement passes (%d per unit)" % per_unit
    tdog = [_t(d, cuda) for d in dogs]
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    dense = [torch.full((h, w, 4), 5.0, dtype=torch.float32, device=cuda) for _ in range(3)]
    assert nm.lib().nm_find_keypoints3_f32(arr(tdog), None, 0, 0, w, h, p.peak_threshold, p.edge_threshold, 2.0, p.sigma_0, 3,
                                           arr(dense), None) == 0
    torch.cuda.synchronize()
    for l in range(3):
        _eq(dense[l], dense_ref[l], "dense map level %d" % l)
    for cap in (1 << 20, len(lists_ref[0]) + 7):                         # unclipped, and clipped inside level 1
        cap = min(cap, 3 * w * h)
        out = torch.full((cap, 4), -3.0, dtype=torch.float32, device=cuda)
        cnt = torch.zeros(3, dtype=torch.int32, device=cuda)
        ws = torch.empty(nm.lib().nm_find_keypoints3_compact_workspace_bytes(w, h), dtype=torch.uint8, device=cuda)
        assert nm.lib().nm_find_keypoints3_compact_f32(arr(tdog), w, h, p.peak_threshold, p.edge_threshold, 2.0, p.sigma_0, 3,
                                                       cap, out.data_ptr(), cnt.data_ptr(), ws.data_ptr(), None) == 0
        torch.cuda.synchronize()
        want, room, live = [], cap, True
        for l in range(3):
            n = len(lists_ref[l]) if live else 0
            if n == 0:
                live = False
            n = min(n, room)
            room -= n
            want.append(lists_ref[l][:n])
        assert cnt.cpu().tolist() == [len(x) for x in want]
        _eq(out[: sum(len(x) for x in want)], np.concatenate(want), "compact ordered lists, capacity %d" % cap)


def test_frame_driver_with_materialised_dog_planes(nm, oracle, cuda, tmp_path):
    """NM_FRAME_DOG=1 (read once per process, so a child process): the frame driver writes the DoG planes and detection
    reads them, as in round 1, instead of forming them from the Gaussian levels. Same keypoints and descriptors."""
    import os
    import subprocess
    import sys
    frames = {"a": H.blurred_frame(3, 320, 200), "b": H.blurred_frame(4, 250, 187)}
    np.savez(tmp_path / "frames.npz", **frames)
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import niftymatch_amd as nm\n"
        "z = np.load(%r)\n"
        "out = {}\n"
        "for k in z.files:\n"
        "    f = z[k]; a = nm.SiftArena(f.shape[1], f.shape[0], 4096)\n"
        "    a.detect_describe(torch.from_numpy(f).cuda()); torch.cuda.synchronize()\n"
        "    n = int(a.num_items.item())\n"
        "    out[k + '_kpts'] = a.kpts[:n].cpu().numpy(); out[k + '_desc'] = a.desc[:n].cpu().numpy()\n"
        "np.savez(%r, **out)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "frames.npz"), str(tmp_path / "out.npz"))
    env = dict(os.environ, NM_FRAME_DOG="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(tmp_path / "out.npz")
    for k, f in frames.items():
        ref = oracle.sift_detect_describe(f, 4096)
        _eq(got[k + "_kpts"], ref["kpts"], "keypoints, frame %s, DoG planes materialised" % k)
        _eq(got[k + "_desc"], ref["desc"], "descriptors, frame %s, DoG planes materialised" % k)
