"""The frame driver's contract: every launch goes to the caller's stream, nothing allocates or synchronises. So
(a) independent arenas on independent streams may run concurrently, and (b) a whole frame can be captured into a HIP
graph and replayed. Both must reproduce the oracle bit for bit."""
import numpy as np
import pytest

import helpers as H
from test_gpu_stages import _eq, _t

pytestmark = pytest.mark.gpu


def test_concurrent_arenas_on_streams(nm, oracle, cuda):
    import torch
    w, h, cap = 320, 240, 4096
    frames = [H.blurred_frame(60 + i, w, h, sigma=3.0) for i in range(6)]
    refs = [oracle.sift_detect_describe(f, cap) for f in frames]
    dev = [_t(f, cuda) for f in frames]
    arenas = [nm.SiftArena(w, h, cap) for _ in frames]
    streams = [torch.cuda.Stream() for _ in frames]
    torch.cuda.synchronize()
    for rep in range(3):
        for a, f, s in zip(arenas, dev, streams):
            with torch.cuda.stream(s):
                a.detect_describe(f)
    torch.cuda.synchronize()
    for a, r in zip(arenas, refs):
        n = int(a.num_items.item())
        assert n == r["n"]
        _eq(a.desc[:n], r["desc"], "descriptors under stream concurrency")
        _eq(a.kpts[:n], r["kpts"], "keypoints under stream concurrency")


def test_frame_and_match_capture_into_hip_graph(nm, oracle, cuda):
    import torch
    w, h, cap = 640, 480, 8192
    f0, f1 = H.blurred_frame(0, w, h), H.blurred_frame(1, w, h)
    r0, r1 = oracle.sift_detect_describe(f0, cap), oracle.sift_detect_describe(f1, cap)
    a0, a1 = nm.SiftArena(w, h, cap), nm.SiftArena(w, h, cap)
    d0, d1 = _t(f0, cuda), _t(f1, cuda)
    ws = nm.MatchWorkspace(cap, cap, cuda)
    res = torch.full((cap,), -1, dtype=torch.int32, device=cuda)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                       # warm-up outside capture (module load, attributes)
        a0.detect_describe(d0); a1.detect_describe(d1)
        nm.sift_match(a0.desc, a1.desc, 0.8, prior=res, workspace=ws, nA=r0["n"], nB=r1["n"])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        a0.detect_describe(d0)
        a1.detect_describe(d1)
        nm.sift_match(a0.desc, a1.desc, 0.8, prior=res, workspace=ws, nA=r0["n"], nB=r1["n"])
    for a in (a0, a1):                               # wipe the outputs, then replay
        a.desc.zero_(); a.num_items.zero_()
    res.fill_(-1)
    g.replay()
    torch.cuda.synchronize()
    assert int(a0.num_items.item()) == r0["n"] and int(a1.num_items.item()) == r1["n"]
    _eq(a0.desc[:r0["n"]], r0["desc"], "descriptors from graph replay")
    ref, _, _ = oracle.sift_matches(r0["desc"], r1["desc"], 0.8, want_distance=False)
    assert np.array_equal(res[:r0["n"]].cpu().numpy(), ref)
    # replay with a different frame in the same input buffer: the graph is data-independent
    f2 = H.blurred_frame(2, w, h)
    d0.copy_(_t(f2, cuda))
    g.replay()
    torch.cuda.synchronize()
    r2 = oracle.sift_detect_describe(f2, cap)
    n2 = int(a0.num_items.item())
    assert n2 == r2["n"]
    _eq(a0.desc[:n2], r2["desc"], "second frame through the same graph")


def test_detect_match_graph_replays_on_different_frames(nm, oracle, cuda):
    """One HIP graph over batched detect+describe of a pair AND the device-sized match (nm_sift_match_batch_dev_f32 reads
    the keypoint counts the frame driver left on the device): replayed on three different frame pairs, whose keypoint
    counts all differ, it must reproduce the oracle's descriptors and matches (siftfunctions.cu:100-181 -> :15-40;
    match.cu:83-117). The host-sized matcher entry cannot do this: its sizes are baked into the captured launches."""
    import torch
    w, h, cap = 640, 480, 8192
    seeds = [(0, 1), (2, 3), (5, 4)]
    fr = {s: H.blurred_frame(s, w, h) for p in seeds for s in p}
    fr[4] = np.roll(fr[5], (2, 3), axis=(0, 1)).copy()          # a shifted copy: many true matches
    ref = {s: oracle.sift_detect_describe(f, cap) for s, f in fr.items()}
    assert len({ref[s]["n"] for s in fr}) == len(fr), "the frames must differ in their keypoint counts"
    a = [nm.SiftArena(w, h, cap), nm.SiftArena(w, h, cap)]
    d = [_t(fr[0], cuda), _t(fr[1], cuda)]
    res = torch.full((cap,), -1, dtype=torch.int32, device=cuda)
    ws = nm.MatchBatchDevWorkspace(1, cap, cap, cuda)
    s = torch.cuda.Stream()

    def enqueue():
        nm.detect_describe_batch(a, d)
        nm.sift_match_batch_dev([a[0].desc], [a[0].num_items], [a[1].desc], [a[1].num_items], [res], 0.8, workspace=ws)
    with torch.cuda.stream(s):
        enqueue()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        enqueue()
    for p in seeds:
        d[0].copy_(_t(fr[p[0]], cuda)); d[1].copy_(_t(fr[p[1]], cuda))
        for x in a:
            x.desc.zero_(); x.num_items.zero_()
        res.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        r0, r1 = ref[p[0]], ref[p[1]]
        assert int(a[0].num_items.item()) == r0["n"] and int(a[1].num_items.item()) == r1["n"]
        _eq(a[0].desc[:r0["n"]], r0["desc"], "descriptors of frame %d from graph replay" % p[0])
        _eq(a[1].desc[:r1["n"]], r1["desc"], "descriptors of frame %d from graph replay" % p[1])
        want, _, _ = oracle.sift_matches(r0["desc"], r1["desc"], 0.8, want_distance=False)
        assert np.array_equal(res[:r0["n"]].cpu().numpy(), want), p
        assert bool((res[r0["n"]:] == -1).all())
    assert (want >= 0).mean() > 0.3


def test_batched_pair_call_captures_into_hip_graph(nm, oracle, cuda):
    """nm_sift_detect_describe_batch forks onto the arenas' side streams with events only: capturable, replayable."""
    import torch
    w, h, cap = 320, 240, 4096
    f = [H.blurred_frame(70 + i, w, h) for i in range(3)]
    r = [oracle.sift_detect_describe(x, cap) for x in f]
    a = [nm.SiftArena(w, h, cap) for _ in range(2)]
    d = [_t(f[0], cuda), _t(f[1], cuda)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        nm.detect_describe_batch(a, d)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        nm.detect_describe_batch(a, d)
    d[1].copy_(_t(f[2], cuda))
    for x in a:
        x.desc.zero_(); x.num_items.zero_()
    g.replay()
    torch.cuda.synchronize()
    for x, ref in zip(a, (r[0], r[2])):
        n = int(x.num_items.item())
        assert n == ref["n"]
        _eq(x.desc[:n], ref["desc"], "batched call through a HIP graph")


def test_many_frame_call_captures_into_hip_graph(nm, oracle, cuda):
    """A call of 32 frames (no octave tail: the per-octave launches over all frames, detection on the side stream) captures into a
    HIP graph and replays, on other frames, to the oracle's results."""
    import torch
    w, h, cap, n = 160, 120, 1024, 32
    f = [H.blurred_frame(300 + i, w, h) for i in range(n + 2)]
    a = [nm.SiftArena(w, h, cap) for _ in range(n)]
    d = [_t(f[i], cuda) for i in range(n)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        nm.detect_describe_batch(a, d)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        nm.detect_describe_batch(a, d)
    d[0].copy_(_t(f[n], cuda)); d[n - 1].copy_(_t(f[n + 1], cuda))        # one frame of either half replaced
    for x in a:
        x.desc.zero_(); x.num_items.zero_()
    g.replay()
    torch.cuda.synchronize()
    for i, src in ((0, n), (1, 1), (n // 2 - 1, n // 2 - 1), (n // 2, n // 2), (n - 2, n - 2), (n - 1, n + 1)):
        ref = oracle.sift_detect_describe(f[src], cap)
        k = int(a[i].num_items.item())
        assert k == ref["n"] and k > 0
        _eq(a[i].desc[:k], ref["desc"], "frame %d of a 32-frame call through a HIP graph" % i)


def test_calls_enqueued_from_several_host_threads(nm, oracle, cuda):
    """bench.py issues its detect calls from a thread pool (the ABI is re-entrant; an arena belongs to one call at a
    time): results stay bit-exact."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    w, h, cap = 256, 192, 4096
    frames = [H.blurred_frame(80 + i, w, h) for i in range(8)]
    refs = [oracle.sift_detect_describe(x, cap) for x in frames]
    dev = [_t(x, cuda) for x in frames]
    arenas = [nm.SiftArena(w, h, cap) for _ in frames]
    streams = [torch.cuda.Stream() for _ in range(4)]
    torch.cuda.synchronize()

    def work(t):
        for rep in range(3):
            with torch.cuda.stream(streams[t]):
                nm.detect_describe_batch(arenas[2 * t:2 * t + 2], dev[2 * t:2 * t + 2])

    with ThreadPoolExecutor(4) as pool:
        list(pool.map(work, range(4)))
    torch.cuda.synchronize()
    for x, ref in zip(arenas, refs):
        n = int(x.num_items.item())
        assert n == ref["n"]
        _eq(x.desc[:n], ref["desc"], "descriptors with multi-threaded enqueue")
