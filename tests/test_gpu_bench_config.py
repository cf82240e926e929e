"""Parity on what bench.py times (BASELINE configs[2] / configs[3]): many-frame 1080p nm_sift_detect_describe_batch calls (16
frames per call here; the bench's own 64-frame calls -- 64 pointers per kernarg block, blockIdx.y = 64 detect grids -- in
test_detect_256_loop_of_the_bench) and 16-pair nm_sift_match_batch_f32 calls on the
real, un-normalised ~12k x ~12k SIFT descriptors of those frames (the MFMA selector + margin logic on real data).
Reference semantics: sift/siftfunctions.cu:100-181 (detect/describe orchestration), kernels/match.cu:83-117 (scan)."""
import numpy as np
import pytest

import helpers as H
from test_gpu_stages import _eq

pytestmark = pytest.mark.gpu

W, H_, CAP = 1920, 1080, 16384
N = 16


@pytest.fixture(scope="module")
def bench16(nm, cuda):
    """The bench's own frames (bench.make_frames: device-side noise + HIP pre-blur), one 16-frame call, exactly as
    bench.py issues it; a second identical call checks that re-use of the arenas changes nothing."""
    import torch

    import bench
    frames = bench.make_frames(nm, torch, cuda, list(range(N)))
    arenas = [nm.SiftArena(W, H_, CAP, device=cuda) for _ in range(N)]
    nm.detect_describe_batch(arenas, frames)
    torch.cuda.synchronize()
    first = [(int(a.num_items.item()), a.desc.clone()) for a in arenas]
    nm.detect_describe_batch(arenas, frames)
    torch.cuda.synchronize()
    yield frames, arenas, first
    for a in arenas:
        a.close()


def test_bench_frames_equal_the_oracle_blur(nm, oracle, cuda, bench16):
    frames, _, _ = bench16
    for s in (0, 1, 15):
        _eq(frames[s], H.blurred_frame(s, W, H_), "bench frame %d (device noise + HIP blur) vs oracle blur" % s)


def test_detect_batch_16x1080p(nm, oracle, cuda, bench16):
    """Frames 0, 1 and 15 of the 16-frame call against the oracle (every output, bit for bit); the other 13 against the
    single-frame driver; the repeated call is identical."""
    import torch
    frames, arenas, first = bench16
    counts = [int(a.num_items.item()) for a in arenas]
    for f, (n0, d0) in enumerate(first):
        assert counts[f] == n0 and torch.equal(arenas[f].desc[:n0], d0[:n0]), "second call differs for frame %d" % f
    for f in (0, 1, 15):
        ref = oracle.sift_detect_describe(H.blurred_frame(f, W, H_), CAP)
        a = arenas[f]
        assert counts[f] == ref["n"] and ref["n"] > 10000
        _eq(a.kpts[:counts[f]], ref["kpts"], "keypoints frame %d" % f)
        _eq(a.orients[:counts[f]], ref["orient"], "orientations frame %d" % f)
        _eq(a.x[:counts[f]], ref["x"], "x frame %d" % f)
        _eq(a.y[:counts[f]], ref["y"], "y frame %d" % f)
        _eq(a.desc[:counts[f]], ref["desc"], "descriptors frame %d" % f)
    single = nm.SiftArena(W, H_, CAP, device=cuda)
    for f in range(N):
        if f in (0, 1, 15):
            continue
        single.detect_describe(frames[f])
        torch.cuda.synchronize()
        n = int(single.num_items.item())
        assert n == counts[f]
        for name in ("kpts", "orients", "x", "y", "desc"):
            assert torch.equal(getattr(single, name)[:n], getattr(arenas[f], name)[:n]), (name, f)
    single.close()


def test_match_batch_16_pairs_on_real_descriptors(nm, oracle, cuda, bench16):
    """16 pairs in ONE nm_sift_match_batch_f32 call, as bench.py issues them: the 8 frame pairs (2i, 2i+1) and their
    reverses (2i+1, 2i). Pairs 0 and 9 (= frames (0,1) and (3,2)) against the oracle's full scan; all 16 against single
    nm_sift_match_f32 calls; pair 0 also with the materialised distance matrix against the oracle's."""
    import torch
    _, arenas, _ = bench16
    cnt = [int(a.num_items.item()) for a in arenas]
    pairs = [(2 * i, 2 * i + 1) for i in range(8)] + [(2 * i + 1, 2 * i) for i in range(8)]
    results = [torch.full((CAP,), -1, dtype=torch.int32, device=cuda) for _ in pairs]
    ws = nm.MatchBatchWorkspace(16, CAP, CAP, cuda)
    nm.sift_match_batch([arenas[a].desc for a, _ in pairs], [arenas[b].desc for _, b in pairs],
                        [cnt[a] for a, _ in pairs], [cnt[b] for _, b in pairs], results, 0.8, workspace=ws)
    torch.cuda.synchronize()
    ws1 = nm.MatchWorkspace(CAP, CAP, cuda)
    for k, (a, b) in enumerate(pairs):
        single, _ = nm.sift_match(arenas[a].desc, arenas[b].desc, 0.8, workspace=ws1, nA=cnt[a], nB=cnt[b])
        assert torch.equal(single[:cnt[a]], results[k][:cnt[a]]), "pair %d: batched != single call" % k
        assert bool((results[k][cnt[a]:] == -1).all())
    for k in (0, 9):
        a, b = pairs[k]
        A = arenas[a].desc[:cnt[a]].cpu().numpy()
        B = arenas[b].desc[:cnt[b]].cpu().numpy()
        ref, Dref, _ = oracle.sift_matches(A, B, 0.8, want_distance=(k == 0))
        m1, ix, m2 = oracle.sift_match_shard(A, B, 0)
        assert np.array_equal(results[k][:cnt[a]].cpu().numpy(), ref), "pair %d vs oracle" % k
        t = nm.sift_match_shard(arenas[a].desc[:cnt[a]], arenas[b].desc[:cnt[b]], 0, workspace=ws1)
        _eq(t[0], m1, "min1 pair %d" % k)
        assert np.array_equal(t[1].cpu().numpy(), ix)
        _eq(t[2], m2, "min2 pair %d" % k)
        if k == 0:
            before = nm.get_distance_mode()
            try:
                for mode in ("mfma", "exact"):            # fp32 MFMA pass: every entry within 1e-4 relative; exact kernel: bit for bit
                    nm.set_distance_mode(mode)
                    res, D = nm.sift_match(arenas[a].desc, arenas[b].desc, 0.8, want_distance=True, workspace=ws1,
                                           nA=cnt[a], nB=cnt[b])
                    torch.cuda.synchronize()
                    assert np.array_equal(res[:cnt[a]].cpu().numpy(), ref)
                    H.assert_distance(nm, D, Dref, "materialised 12k x 12k distance matrix (%s)" % mode)
                    if mode == "mfma":                    # real SIFT descriptors: the listed blocks fit the list by a wide margin
                        listed, cap = nm.match_distance_listed(ws1, cnt[a], cnt[b])
                        assert 0 <= listed < cap // 4, (listed, cap)
                    del D
            finally:
                nm.set_distance_mode(before)
    # the last call on ws1 was the shard call of pair 9: real SIFT descriptors rarely need the exact fallback
    assert 0 <= nm.match_fallback_count(ws1, cnt[pairs[9][0]], cnt[pairs[9][1]]) < 200


def test_match_batch_dev_16_pairs_on_real_descriptors(nm, oracle, cuda, bench16):
    """nm_sift_match_batch_dev_f32: the same 16 pairs with the set sizes read from the frame driver's d_num_items ON THE
    DEVICE (what bench.py's timed loop issues: no host read-back between detect and match; the reference's flow is
    siftfunctions.cu:165-178 -> :15-40). Must equal the host-sized batched call on every pair and the oracle's scan
    (match.cu:83-117) on pairs 0 and 9; rows past the real size stay untouched. Both MFMA screens."""
    import torch
    _, arenas, _ = bench16
    cnt = [int(a.num_items.item()) for a in arenas]
    pairs = [(2 * i, 2 * i + 1) for i in range(8)] + [(2 * i + 1, 2 * i) for i in range(8)]
    host = [torch.full((CAP,), -1, dtype=torch.int32, device=cuda) for _ in pairs]
    nm.sift_match_batch([arenas[a].desc for a, _ in pairs], [arenas[b].desc for _, b in pairs],
                        [cnt[a] for a, _ in pairs], [cnt[b] for _, b in pairs], host, 0.8,
                        workspace=nm.MatchBatchWorkspace(16, CAP, CAP, cuda))
    before = nm.get_match_screen()
    ws = nm.MatchBatchDevWorkspace(16, CAP, CAP, cuda)
    try:
        for screen in ("f16", "bf16x3", "f32"):
            nm.set_match_screen(screen)
            dev = [torch.full((CAP,), -5, dtype=torch.int32, device=cuda) for _ in pairs]
            nm.sift_match_batch_dev([arenas[a].desc for a, _ in pairs], [arenas[a].num_items for a, _ in pairs],
                                    [arenas[b].desc for _, b in pairs], [arenas[b].num_items for _, b in pairs],
                                    dev, 0.8, workspace=ws)
            torch.cuda.synchronize()
            for k, (a, b) in enumerate(pairs):
                assert torch.equal(dev[k][:cnt[a]], host[k][:cnt[a]]), "pair %d (%s): device-sized != host-sized" % (k, screen)
                assert bool((dev[k][cnt[a]:] == -5).all()), "pair %d: rows past nA were written" % k
            for k in (0, 9):
                a, b = pairs[k]
                ref, _, _ = oracle.sift_matches(arenas[a].desc[:cnt[a]].cpu().numpy(), arenas[b].desc[:cnt[b]].cpu().numpy(),
                                                0.8, want_distance=False, prior=np.full(cnt[a], -5, np.int32))
                assert np.array_equal(dev[k][:cnt[a]].cpu().numpy(), ref), "pair %d (%s) vs oracle" % (k, screen)
    finally:
        nm.set_match_screen(before)


def test_detect_256_loop_of_the_bench(nm, oracle, cuda):
    """BASELINE configs[3] exactly as bench.detect_256 runs it (sift/siftfunctions.cu:100-181 per frame): 256 distinct 1080p
    frames in four 64-frame calls (bench.py's default call size since round 4: NM_SIFT_MAX_BATCH frames per launch sequence)
    re-using the same 64 arenas four times per pass, several passes.
    The first, a middle and the last frame against the oracle (every output, bit for bit); all 256 keypoint counts and
    descriptor checksums against single-frame calls."""
    import torch

    import bench
    streams = [torch.cuda.Stream(device=cuda) for _ in range(4)]
    arenas = [nm.SiftArena(W, H_, CAP, device=cuda) for _ in range(64)]
    frames = bench.make_frames(nm, torch, cuda, list(range(256)))
    counts = torch.zeros(256, dtype=torch.int32, device=cuda)
    sums = torch.zeros(256, dtype=torch.int64, device=cuda)
    keep = {0: None, 137: None, 255: None}
    rows = torch.arange(CAP, device=cuda)

    def after_call(c, b, e, ar):                 # runs on the call's stream, right behind it
        for k, a in enumerate(ar):
            counts[b + k] = a.num_items[0]
            live = (rows < a.num_items[0]).long()       # rows past the count hold earlier frames' descriptors
            sums[b + k] = (a.desc.view(torch.int32).long().sum(dim=1) * live).sum()     # bit patterns: exact, order-free
            if b + k in keep:
                keep[b + k] = (a.kpts.clone(), a.orients.clone(), a.x.clone(), a.y.clone(), a.desc.clone())

    out = bench.detect_256(nm, torch, None, cuda, cuda, 0, 1, arenas, streams, bench.parse_args([]).batch, passes=2, frames=frames,
                           after_call=after_call)
    torch.cuda.synchronize()
    assert out["frames_per_s"] > 0 and out["frames_per_s_sustained"] > 0 and len(out["ms_per_pass"]) == 2
    counts = counts.cpu().numpy()
    assert out["keypoints_total"] == int(counts.sum()) and counts.min() > 10000
    for f, (kpts, orients, x, y, desc) in keep.items():
        ref = oracle.sift_detect_describe(H.blurred_frame(f, W, H_), CAP)
        n = int(counts[f])
        assert n == ref["n"]
        _eq(kpts[:n], ref["kpts"], "keypoints frame %d" % f)
        _eq(orients[:n], ref["orient"], "orientations frame %d" % f)
        _eq(x[:n], ref["x"], "x frame %d" % f)
        _eq(y[:n], ref["y"], "y frame %d" % f)
        _eq(desc[:n], ref["desc"], "descriptors frame %d" % f)
    single = nm.SiftArena(W, H_, CAP, device=cuda)
    sums = sums.cpu().numpy()
    for f in range(256):
        single.detect_describe(frames[f])
        n = int(single.num_items.item())
        assert n == counts[f], "frame %d: %d keypoints in the 256-frame loop, %d alone" % (f, counts[f], n)
        assert int(single.desc[:n].view(torch.int32).long().sum().item()) == int(sums[f]), "frame %d: descriptor checksum" % f
    single.close()
    for a in arenas:
        a.close()
