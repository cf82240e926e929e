"""The octave-tail launch of the frame driver (csrc/nm_tail.hip): the octaves >= 2 of a one- or two-frame call run as ONE
persistent launch (+ one scan launch) instead of 8 launches per octave. Reference orchestration it stands for:
sift/siftfunctions.cu:42-181 per octave, pyramidata.cu:84-91 (ordered compaction). Every output must be bit-identical to the
per-octave launches (NM_FRAME_TAIL=0) and to the oracle, for any geometry, under concurrency, and call after call."""
import os
import threading

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
CAP = 4096


def _arena(nm, cuda, w, h, tail, cap=CAP):
    old = os.environ.get("NM_FRAME_TAIL")
    if tail is None:
        os.environ.pop("NM_FRAME_TAIL", None)
    else:
        os.environ["NM_FRAME_TAIL"] = str(tail)
    try:
        return nm.SiftArena(w, h, cap, device=cuda)
    finally:
        if old is None:
            os.environ.pop("NM_FRAME_TAIL", None)
        else:
            os.environ["NM_FRAME_TAIL"] = old


def _out(a):
    n = int(a.num_items.item())
    return n, [t[:n].cpu().numpy().view(np.uint32).copy() for t in (a.kpts, a.orients, a.x, a.y, a.desc)]


def _same(o0, o1):
    return o0[0] == o1[0] and all(np.array_equal(x, y) for x, y in zip(o0[1], o1[1]))


@pytest.mark.parametrize("w,h", [(256, 192), (320, 240), (400, 300), (510, 250), (640, 480), (257, 131), (1916, 1076), (1280, 720)])
def test_tail_equals_per_octave_launches(nm, cuda, w, h):
    import torch
    assert nm.lib().nm_sift_tail_plan(w, h, 2, None, 0, None) > 0, "geometry must take the tail path"
    a0, a1 = _arena(nm, cuda, w, h, 0, 16384), _arena(nm, cuda, w, h, None, 16384)
    assert nm.lib().nm_sift_arena_tail_segments(a0._h) == 0 and nm.lib().nm_sift_arena_tail_segments(a1._h) > 0
    for seed in (3, 4):
        f = torch.from_numpy(H.blurred_frame(seed, w, h, sigma=2.5)).to(cuda)
        a0.detect_describe(f); a1.detect_describe(f)
        torch.cuda.synchronize()
        o0, o1 = _out(a0), _out(a1)
        assert o0[0] > 50 and _same(o0, o1), (w, h, seed, o0[0], o1[0])
    a0.close(); a1.close()


@pytest.mark.parametrize("T", [1, 2, 3])
def test_tail_first_octave_choices_against_the_oracle(nm, oracle, cuda, T):
    import torch
    w, h = 640, 480
    a = _arena(nm, cuda, w, h, T, 8192)
    frame = H.blurred_frame(11, w, h, sigma=3.0)
    ref = oracle.sift_detect_describe(frame, 8192)
    a.detect_describe(torch.from_numpy(frame).to(cuda))
    torch.cuda.synchronize()
    n = int(a.num_items.item())
    assert n == ref["n"] and n > 1000
    for got, want in ((a.kpts, ref["kpts"]), (a.orients, ref["orient"]), (a.x, ref["x"]), (a.y, ref["y"]), (a.desc, ref["desc"])):
        assert np.array_equal(got[:n].cpu().numpy().view(np.uint32), want.view(np.uint32))
    a.close()


def test_capacity_clipping_inside_the_tail_octaves(nm, oracle, cuda):
    """siftfunctions.cu:165-169: the capacity runs out in the middle of a tail octave's level."""
    import torch
    w, h = 640, 480
    frame = H.blurred_frame(12, w, h, sigma=3.0)
    full = oracle.sift_detect_describe(frame, 8192)
    cap = full["n"] - 40                              # the last 40 keypoints belong to the small octaves
    ref = oracle.sift_detect_describe(frame, cap)
    a = _arena(nm, cuda, w, h, None, cap)
    a.detect_describe(torch.from_numpy(frame).to(cuda))
    torch.cuda.synchronize()
    n = int(a.num_items.item())
    assert n == ref["n"] == cap
    assert np.array_equal(a.desc[:n].cpu().numpy().view(np.uint32), ref["desc"].view(np.uint32))
    assert np.array_equal(a.kpts[:n].cpu().numpy().view(np.uint32), ref["kpts"].view(np.uint32))
    a.close()


def test_two_frame_calls_and_repeated_calls_leave_clean_state(nm, cuda):
    """The launch cleans its own state words (no memset): 60 calls in a row, one- and two-frame calls alternating on the same
    arenas, always the same answers."""
    import torch
    w, h = 480, 360
    ars = [_arena(nm, cuda, w, h, None) for _ in range(2)]
    ref = _arena(nm, cuda, w, h, 0)
    frames = [torch.from_numpy(H.blurred_frame(s, w, h, sigma=2.5)).to(cuda) for s in (20, 21)]
    want = []
    for f in frames:
        ref.detect_describe(f)
        torch.cuda.synchronize()
        want.append(_out(ref))
    for it in range(30):
        nm.detect_describe_batch(ars, frames)
        ars[1].detect_describe(frames[0])             # the second arena as the FIRST (state-lending) arena of a call
        if it % 10 == 9:
            torch.cuda.synchronize()
            assert _same(_out(ars[0]), want[0]) and _same(_out(ars[1]), want[0])
            nm.detect_describe_batch(ars, frames)
            torch.cuda.synchronize()
            assert _same(_out(ars[0]), want[0]) and _same(_out(ars[1]), want[1])
    for a in ars + [ref]:
        a.close()


def test_many_tail_launches_side_by_side(nm, cuda):
    """Eight host threads, eight streams, single-frame calls back to back: up to eight persistent tail launches share the chip
    with each other and with everything else the calls launch. The ticket order makes each of them complete whatever is
    resident; every result equals the sequential one."""
    import torch
    w, h = 640, 480
    n_thr, reps = 8, 12
    frames = [torch.from_numpy(H.blurred_frame(30 + k, w, h, sigma=3.0)).to(cuda) for k in range(n_thr)]
    ref = _arena(nm, cuda, w, h, 0, 8192)
    want = []
    for f in frames:
        ref.detect_describe(f)
        torch.cuda.synchronize()
        want.append(_out(ref))
    ars = [_arena(nm, cuda, w, h, None, 8192) for _ in range(n_thr)]
    streams = [torch.cuda.Stream(device=cuda) for _ in range(n_thr)]
    errs = []

    def work(k):
        try:
            torch.cuda.set_device(cuda)
            with torch.cuda.stream(streams[k]):
                for r in range(reps):
                    ars[k].detect_describe(frames[(k + r) % n_thr])
                streams[k].synchronize()
        except Exception as e:                        # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(n_thr)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for k in range(n_thr):
        assert _same(_out(ars[k]), want[(k + reps - 1) % n_thr]), k
    for a in ars + [ref]:
        a.close()


def test_launch_count_of_a_single_frame_call(nm, cuda):
    """A 1080p frame: 23 launches with the tail (51 without): the plan says what the tail launch covers."""
    import ctypes as C
    seg = (C.c_int * (8 * 40))()
    info = (C.c_int * 4)()
    n = nm.lib().nm_sift_tail_plan(1920, 1080, 2, seg, 40, info)
    assert n > 0 and info[3] == 4                     # octaves 2..5 in one launch
    launches = 1 + 2 * 5 + 2 * 3 + 2 + 2 * 2          # base blur, 2 octaves x 5 levels, 2 x (detect, scan, gather), tail + scan, 2 x describe
    a = nm.SiftArena(1920, 1080, CAP, device=cuda)
    assert nm.lib().nm_sift_arena_launches_per_call(a._h, 1) == launches == 23
    assert nm.lib().nm_sift_arena_launches_per_call(a._h, 2) == 23
    assert nm.lib().nm_sift_arena_launches_per_call(a._h, 16) == 1 + 6 * 8 + 2
    a.close()


def test_a_failed_tail_launch_is_reported_and_the_next_call_is_correct(nm, cuda):
    """ADVICE r4 (nm_tail.hip): a wait that times out sets a sticky error word and the rest of the launch drains without working.
    The word must not outlive its launch and the call must not look successful: d_num_items reads -1, the status query says 1,
    and the NEXT call on the same arenas is bit-identical to a clean arena's. (The timeout itself takes ~2 s of spinning to
    provoke; the test hook sets the word as the timed-out wait would.)"""
    import torch
    w, h = 640, 480
    frames = [torch.from_numpy(H.blurred_frame(s, w, h, sigma=3.0)).to(cuda) for s in (40, 41)]
    ref = _arena(nm, cuda, w, h, 0, 8192)
    want = []
    for f in frames:
        ref.detect_describe(f)
        torch.cuda.synchronize()
        want.append(_out(ref))
    ars = [_arena(nm, cuda, w, h, None, 8192) for _ in range(2)]
    assert ars[0].tail_status() == 0
    for n_frames in (1, 2):
        ars[0].tail_inject_error()
        if n_frames == 1:
            ars[0].detect_describe(frames[0])
        else:
            nm.detect_describe_batch(ars, frames)
        torch.cuda.synchronize()
        assert ars[0].tail_status() == 1
        for a in ars[:n_frames]:
            assert int(a.num_items.item()) == -1
        # the call after the failed one: clean state, correct results, status back to 0
        nm.detect_describe_batch(ars, frames)
        torch.cuda.synchronize()
        assert ars[0].tail_status() == 0
        assert _same(_out(ars[0]), want[0]) and _same(_out(ars[1]), want[1])
        ars[1].detect_describe(frames[0])
        torch.cuda.synchronize()
        assert ars[1].tail_status() == 0 and _same(_out(ars[1]), want[0])
    for a in ars + [ref]:
        a.close()


def test_arenas_with_different_tail_plans_take_the_per_octave_launches(nm, cuda):
    """ADVICE r4 (nm_frame.hip): the first arena's plan is paired with every arena's own plane table, so arenas created under
    different NM_FRAME_TAIL settings must not share a tail launch."""
    import torch
    w, h = 640, 480
    frames = [torch.from_numpy(H.blurred_frame(s, w, h, sigma=3.0)).to(cuda) for s in (42, 43)]
    ref = _arena(nm, cuda, w, h, 0, 8192)
    want = []
    for f in frames:
        ref.detect_describe(f)
        torch.cuda.synchronize()
        want.append(_out(ref))
    ars = [_arena(nm, cuda, w, h, 2, 8192), _arena(nm, cuda, w, h, 3, 8192)]
    nm.detect_describe_batch(ars, frames)
    torch.cuda.synchronize()
    assert _same(_out(ars[0]), want[0]) and _same(_out(ars[1]), want[1])
    for a in ars + [ref]:
        a.close()
