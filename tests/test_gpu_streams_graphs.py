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
