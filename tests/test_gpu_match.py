"""GPU parity of the matcher: fused MFMA path, exact API building blocks, shard + merge, edge cases (KAT-12)."""
import numpy as np
import pytest

import helpers as H
from test_gpu_stages import _eq, _t

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["f16", "bf16x3", "f32"])
def screen(request, nm):
    """Every test of this module runs with all MFMA screens of the fused matcher (nm_sift_match_set_screen): the two-stage
    one (fp16 coarse pass + split-bf16 second pass, default), the split-bf16 one and the fp32 one. The expected results are
    the same -- the oracle's."""
    before, dbefore = nm.get_match_screen(), nm.get_distance_mode()
    nm.set_match_screen(request.param)
    # how a requested `distance` is filled rides along: the MFMA pass (default) under two screens, the exact kernel under one
    nm.set_distance_mode("exact" if request.param == "bf16x3" else "mfma")
    yield request.param
    nm.set_match_screen(before)
    nm.set_distance_mode(dbefore)


def _match(nm, cuda, A, B, amb=0.8, want_distance=False, prior=None):
    import torch
    pr = None if prior is None else _t(np.asarray(prior, np.int32), cuda)
    res, D = nm.sift_match(_t(A, cuda), _t(B, cuda), amb, want_distance=want_distance, prior=pr)
    torch.cuda.synchronize()
    return res.cpu().numpy(), (None if D is None else D.cpu().numpy())


@pytest.mark.parametrize("na,nb", [(1024, 1024), (1000, 777), (300, 2500), (5, 3), (257, 129)])
def test_fused_match_random(nm, oracle, cuda, na, nb):
    A = H.synth.descriptors(1, na)
    B = H.synth.descriptors(2, nb)
    ref, Dref, _ = oracle.sift_matches(A, B, 0.8)
    got, D = _match(nm, cuda, A, B, want_distance=True)
    assert np.array_equal(got, ref)
    H.assert_distance(nm, D, Dref, "distance matrix")
    got2, _ = _match(nm, cuda, A, B, want_distance=False)
    assert np.array_equal(got2, ref)
    assert (ref >= 0).sum() >= 0


def test_fused_match_sift_descriptors(nm, oracle, cuda):
    """Real (un-normalised, Q12) descriptors of two frames, incl. a shifted copy: true matches, cancellation hazard."""
    f0 = H.blurred_frame(0, 640, 480)
    f1 = np.roll(f0, (2, 3), axis=(0, 1)).copy()
    a = oracle.sift_detect_describe(f0, 8192)["desc"]
    b = oracle.sift_detect_describe(f1, 8192)["desc"]
    ref, _, _ = oracle.sift_matches(a, b, 0.8, want_distance=False)
    got, _ = _match(nm, cuda, a, b)
    assert np.array_equal(got, ref)
    assert (ref >= 0).mean() > 0.3


def test_match_hard_near_duplicates(nm, oracle, cuda):
    rng = np.random.default_rng(5)
    A = H.synth.descriptors(3, 2000)
    B = H.synth.descriptors(4, 2000)
    idx = rng.choice(2000, 200, replace=False)
    B[idx] = A[idx] + (1e-3 * rng.standard_normal((200, 128))).astype(np.float32)
    B[7] = A[9]                      # exact duplicate: min1 = 0, min2 > 0 -> ratio 0 -> matched
    ref, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False)
    got, _ = _match(nm, cuda, A, B)
    assert np.array_equal(got, ref)
    assert ref[9] == 7 and (ref[idx] == idx).mean() > 0.95


def test_match_kat12_edge_cases(nm, oracle, cuda):
    base = H.synth.descriptors(8, 6)
    # two exact duplicates of A[0] in B: min2 == 0 -> result left untouched (prior value)
    B = base.copy(); A = base[:2].copy()
    B[3] = A[0]; B[5] = A[0]
    prior = np.array([42, 17], np.int32)
    ref, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False, prior=prior)
    got, _ = _match(nm, cuda, A, B, prior=prior)
    assert np.array_equal(got, ref) and ref[0] == 42
    # M = 1: min2 stays 2139095040.0f -> ratio ~ 0 -> index 0
    ref, _, _ = oracle.sift_matches(A, base[4:5], 0.8, want_distance=False)
    got, _ = _match(nm, cuda, A, base[4:5])
    assert np.array_equal(got, ref) and (ref == 0).all()
    # tie for the minimum: lowest index wins, ratio 1 -> -1 at 0.8, matched at ambiguity 1.5
    B = base.copy(); B[4] = B[1]; A = (B[1:2] + 0.01).astype(np.float32)
    for amb in (0.8, 1.5):
        ref, _, _ = oracle.sift_matches(A, B, amb, want_distance=False)
        got, _ = _match(nm, cuda, A, B, amb=amb)
        assert np.array_equal(got, ref)
    assert ref[0] == 1


def test_api_building_blocks(nm, oracle, cuda):
    import torch
    A = H.synth.descriptors(1, 333)
    B = H.synth.descriptors(2, 200)
    At = nm.transpose(_t(A, cuda))
    _eq(At, oracle.transpose(A), "transpose")
    Dt = nm.bf_distance(At, _t(B, cuda))
    Dt_ref = oracle.bf_distance(oracle.transpose(A), B)
    _eq(Dt, Dt_ref, "bf_distance (transposed layout)")
    D = nm.transpose(Dt)
    res = nm.get_sift_matches(D, 0.8)
    torch.cuda.synchronize()
    ref = oracle.get_sift_matches(oracle.transpose(Dt_ref), 0.8)
    assert np.array_equal(res.cpu().numpy(), ref)
    # buffer_width > cols and a prior that must survive min2 <= 0
    M = np.zeros((3, 8), np.float32); M[0, :4] = [5, 1, 3, 1]; M[1, :4] = [0, 0, 2, 2]; M[2, :4] = [4, 9, 8, 7]
    ref = oracle.get_sift_matches(M, 0.8, prior=np.array([9, 9, 9], np.int32), cols=4)
    got = nm.get_sift_matches(_t(M, cuda), 0.8, prior=_t(np.array([9, 9, 9], np.int32), cuda), cols=4)
    assert np.array_equal(got.cpu().numpy(), ref) and ref[1] == 9


def test_shard_and_merge_equals_unsharded(nm, oracle, cuda):
    import torch
    A = H.synth.descriptors(1, 1500)
    B = H.synth.descriptors(2, 2100)
    B[100] = B[1900]                                     # cross-shard tie: the lower global index must win
    ref, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False)
    m1r, ixr, m2r = oracle.sift_match_shard(A, B, 0)
    bounds = [0, 700, 1400, 2100]
    m1s, ixs, m2s = [], [], []
    for g in range(3):
        m1, ix, m2 = nm.sift_match_shard(_t(A, cuda), _t(B[bounds[g]:bounds[g + 1]], cuda), bounds[g])
        m1s.append(m1); ixs.append(ix); m2s.append(m2)
    res = nm.sift_match_merge(torch.stack(m1s), torch.stack(ixs), torch.stack(m2s), 0.8)
    torch.cuda.synchronize()
    assert np.array_equal(res.cpu().numpy(), ref)
    one = nm.sift_match_shard(_t(A, cuda), _t(B, cuda), 0)
    _eq(one[0], m1r, "min1"); assert np.array_equal(one[1].cpu().numpy(), ixr); _eq(one[2], m2r, "min2")


def test_full_size_properties_12k(nm, cuda):
    """BASELINE config-3 size (12k x 12k): size-independent properties instead of an oracle run."""
    import torch
    A = _t(H.synth.descriptors(1, 12223), cuda)
    perm = torch.randperm(12223, device=cuda, generator=torch.Generator(device=cuda).manual_seed(0))
    B = A[perm].contiguous()
    res, _ = nm.sift_match(A, B, 0.8)                    # every row has an exact duplicate in B: distance 0, ratio 0
    inv = torch.empty_like(perm); inv[perm] = torch.arange(12223, device=cuda)
    assert torch.equal(res.long(), inv)
    res2, _ = nm.sift_match(A, B, 0.8)                   # idempotent / run-to-run deterministic
    assert torch.equal(res, res2)
    m1, ix, m2 = nm.sift_match_shard(A, B, 0)
    assert float(m1.abs().max()) == 0.0 and torch.equal(ix.long(), inv) and bool((m2 > 0).all())


def test_config5_shard_scale_100k_by_12k5(nm, cuda):
    """BASELINE config 5, one rank's share: 100 000 queries against a 12 500-row shard (3.2e11 flop). Properties:
    every shard row planted into A is found at distance 0 with the right GLOBAL index; run-to-run identical."""
    import torch
    nA, nB, off = 100_000, 12_500, 37_500                   # rank 3 of 8
    g = torch.Generator(device=cuda).manual_seed(1)
    A = torch.rand((nA, 128), device=cuda, generator=g)
    B = torch.rand((nB, 128), device=cuda, generator=g)
    rows = torch.randperm(nA, device=cuda, generator=g)[:nB]
    A[rows] = B                                             # query rows[j] is an exact copy of shard row j
    ws = nm.MatchWorkspace(nA, nB, cuda)
    m1, ix, m2 = nm.sift_match_shard(A, B, off, workspace=ws)
    torch.cuda.synchronize()
    assert float(m1[rows].abs().max()) == 0.0
    assert torch.equal(ix[rows].long(), torch.arange(nB, device=cuda) + off)
    assert bool((m2 > 0).all()) and bool((m1 <= m2).all()) and int(ix.min()) >= off and int(ix.max()) < off + nB
    m1b, ixb, m2b = nm.sift_match_shard(A, B, off, workspace=ws)
    assert torch.equal(m1, m1b) and torch.equal(ix, ixb) and torch.equal(m2, m2b)
    # spot-check 64 random queries against a direct fp64 evaluation
    q = torch.randint(0, nA, (64,), device=cuda, generator=g)
    d = ((A[q].double()[:, None, :] - B.double()[None]) ** 2).sum(-1)
    top = d.topk(2, dim=1, largest=False)
    assert torch.equal(ix[q].long() - off, top.indices[:, 0])
    assert torch.allclose(m1[q].double(), top.values[:, 0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(m2[q].double(), top.values[:, 1], rtol=1e-5)


def test_config5_full_size_eight_shards_merge_equals_the_unsharded_call(nm, cuda):
    """BASELINE configs[4] at FULL size on one GPU: all-pairs 100 000 x 100 000, candidates cut into 8 shards of 12 500 rows as
    8 ranks would hold them. The 8 shard calls (offsets 0 ... 87 500) merged -- shard-major triples through
    nm_sift_match_merge_f32 AND the rank-major packed layout an ncclAllGather leaves, through nm_sift_match_merge_packed_f32
    (what nm_sift_match_allgather_f32 merges) -- must equal the single unsharded nm_sift_match_f32 call index for index.
    Planted cross-shard duplicates: the lowest GLOBAL index wins a tie (match.cu:94-105), and with two exact copies min2 = 0
    leaves the prior untouched (match.cu:107-116). 64 rows are checked against a binary64 evaluation of all 100 000 candidates."""
    import torch
    n, n_shards, amb = 100_000, 8, 0.8
    per = n // n_shards
    g = torch.Generator(device=cuda).manual_seed(5)
    A = torch.rand((n, 128), device=cuda, generator=g)
    B = torch.rand((n, 128), device=cuda, generator=g)
    # (a) candidate 90 001 (shard 7) is an exact copy of candidate 13 000 (shard 1); query 7 sits 1e-3 off both: two equal
    #     minima in different shards, lower global index must win, and the ratio (= 1) only passes an ambiguity > 1
    B[90_001] = B[13_000]
    A[7] = B[13_000] + 1e-3
    # (b) query 11 IS candidate 50 000 (shard 4) and candidate 99 999 (shard 7) is a copy of it too: min1 = min2 = 0 across shards
    B[99_999] = B[50_000]
    A[11] = B[50_000]
    # (c) query 13 is an exact copy of candidate 0 only: distance 0 at global index 0, the clamped-min2 corner (match.cu:91)
    A[13] = B[0]
    ws_full = nm.MatchWorkspace(n, n, cuda)
    ws_shard = nm.MatchWorkspace(n, per, cuda)
    for ambiguity in (amb, 1.5):
        prior = torch.full((n,), -7, dtype=torch.int32, device=cuda)
        want, _ = nm.sift_match(A, B, ambiguity, prior=prior.clone(), workspace=ws_full)
        m1s, ixs, m2s, blocks = [], [], [], []
        for r in range(n_shards):
            m1, ix, m2 = nm.sift_match_shard(A, B[r * per:(r + 1) * per], r * per, workspace=ws_shard)
            m1s.append(m1); ixs.append(ix); m2s.append(m2)
            blocks.append(torch.stack([m1.view(torch.int32), ix, m2.view(torch.int32)]))
        got = nm.sift_match_merge(torch.stack(m1s), torch.stack(ixs), torch.stack(m2s), ambiguity, prior=prior.clone())
        packed = torch.stack(blocks).contiguous()                 # (rank, 3, nA): the all-gather's receive buffer
        got_p = prior.clone()
        assert nm.lib().nm_sift_match_merge_packed_f32(packed.data_ptr(), n_shards, n, got_p.data_ptr(), ambiguity, None) == 0
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(got_p, want), ambiguity
        assert int(want[11]) == -7                                # min2 == 0: result left untouched
        assert int(want[13]) == 0                                 # unique exact copy at global index 0
        assert int(want[7]) == (13_000 if ambiguity > 1 else -1)  # cross-shard tie: lowest global index, ratio exactly 1
        assert int((want >= 0).sum()) > 0 if ambiguity > 1 else True
    # 64 random rows + the planted ones against binary64 over ALL candidates (the last loop's ambiguity is 1.5)
    q = torch.cat([torch.randint(0, n, (64,), device=cuda, generator=g), torch.tensor([7, 13], device=cuda)])
    d = torch.empty((q.numel(), n), dtype=torch.float64, device=cuda)
    for c0 in range(0, n, 12_500):
        d[:, c0:c0 + 12_500] = ((A[q].double()[:, None, :] - B[c0:c0 + 12_500].double()[None]) ** 2).sum(-1)
    top = d.topk(2, dim=1, largest=False)
    ratio_ok = top.values[:, 0] / top.values[:, 1] < 1.5 * (1 - 1e-6)
    exp = torch.where(ratio_ok, top.indices[:, 0], torch.full_like(top.indices[:, 0], -1))
    clear = (top.values[:, 0] / top.values[:, 1] - 1.5).abs() > 1e-5      # rows whose ratio test binary64 decides safely
    tie = (top.values[:, 1] - top.values[:, 0]) < 1e-9 * top.values[:, 1]  # exact ties: lowest index, topk's pick is arbitrary
    sel = clear & ~tie
    assert torch.equal(want[q].long()[sel], exp[sel])


def test_match_random_shapes_sweep(nm, oracle, cuda):
    """Ragged sizes around every tiling boundary of the fused kernel (256 queries / 128 candidates / chunking)."""
    rng = np.random.default_rng(2026)
    shapes = [(1, 1), (1, 2), (2, 1), (255, 127), (256, 128), (257, 129), (511, 385), (33, 1025)]
    shapes += [(int(rng.integers(1, 900)), int(rng.integers(1, 900))) for _ in range(6)]
    for na, nb in shapes:
        A = H.synth.descriptors(100 + na, na)
        B = H.synth.descriptors(200 + nb, nb)
        ref, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False)
        got, _ = _match(nm, cuda, A, B)
        assert np.array_equal(got, ref), (na, nb)
        m1, ix, m2 = oracle.sift_match_shard(A, B, 5)        # min2 unclamped: +inf for a one-row shard
        t = nm.sift_match_shard(_t(A, cuda), _t(B, cuda), 5)
        assert np.array_equal(t[1].cpu().numpy(), ix), (na, nb)
        assert np.array_equal(t[0].cpu().numpy(), m1) and np.array_equal(t[2].cpu().numpy(), m2), (na, nb)


def test_match_empty_sets_are_noops(nm, cuda):
    import torch
    A = _t(H.synth.descriptors(1, 8), cuda)
    prior = torch.full((8,), 5, dtype=torch.int32, device=cuda)
    ws = nm.MatchWorkspace(8, 8, cuda)
    res, _ = nm.sift_match(A, A, 0.8, prior=prior, workspace=ws, nA=0, nB=8)
    res, _ = nm.sift_match(A, A, 0.8, prior=prior, workspace=ws, nA=8, nB=0)
    torch.cuda.synchronize()
    assert bool((res == 5).all())


def test_fallback_rate_and_forced_fallback(nm, oracle, cuda):
    """The exact fallback must be rare on ordinary data and must fire (and give the exact answer) on near-ties that
    the MFMA formulation cannot resolve."""
    import torch
    A = H.synth.descriptors(1, 4000) * 100
    B = H.synth.descriptors(2, 3000) * 100
    ws = nm.MatchWorkspace(4000, 3000, cuda)
    res, _ = nm.sift_match(_t(A, cuda), _t(B, cuda), 0.8, workspace=ws)
    assert nm.match_fallback_count(ws, 4000, 3000) < 0.05 * 4000
    ref, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False)
    assert np.array_equal(res.cpu().numpy(), ref)
    # near-ties: every query has three candidates whose distances differ by ~1 ulp of the accumulated sum
    rng = np.random.default_rng(0)
    A = rng.uniform(0, 255, (300, 128)).astype(np.float32)
    B = rng.uniform(0, 255, (900, 128)).astype(np.float32)
    for i in range(300):
        for c in range(3):
            v = A[i].copy()
            v[(7 * i + c) % 128] += np.float32(60.0)              # three candidates at (almost) the same distance 3600
            v[(11 * i + 5 * c) % 128] += np.float32(1e-3 * c)
            B[3 * i + c] = v
    ws = nm.MatchWorkspace(300, 900, cuda)
    m1, ix, m2 = nm.sift_match_shard(_t(A, cuda), _t(B, cuda), 0, workspace=ws)
    assert nm.match_fallback_count(ws, 300, 900) > 100
    m1r, ixr, m2r = oracle.sift_match_shard(A, B, 0)
    assert np.array_equal(ix.cpu().numpy(), ixr)
    assert np.array_equal(m1.cpu().numpy(), m1r) and np.array_equal(m2.cpu().numpy(), m2r)
    # a handful of unprovable rows among ordinary ones (the few-rows path of the fallback: <= 24 listed rows): three
    # near-tied candidates inside one candidate tile (a segment reports only its best two with their indices), the
    # groups spread over the candidate slices, plus an exact duplicate of a tied candidate (lowest index must win)
    for n_hard in (1, 7, 24):
        A = rng.uniform(0, 255, (500, 128)).astype(np.float32)
        B = rng.uniform(0, 255, (2900, 128)).astype(np.float32)
        for t in range(n_hard):
            i = 17 * t + 3
            for c2, jj in enumerate((100 * t + 5, 100 * t + 6, 100 * t + 7)):
                v = A[i].copy()
                v[(7 * t + c2) % 128] += np.float32(60.0)
                v[(11 * t + 5 * c2) % 128] += np.float32(1e-3 * c2)
                B[jj] = v
        B[2000] = B[5]                                             # duplicate of a tied candidate of query 3 (t = 0)
        ws = nm.MatchWorkspace(500, 2900, cuda)
        m1, ix, m2 = nm.sift_match_shard(_t(A, cuda), _t(B, cuda), 0, workspace=ws)
        assert n_hard <= nm.match_fallback_count(ws, 500, 2900) <= 24, n_hard
        m1r, ixr, m2r = oracle.sift_match_shard(A, B, 0)
        assert np.array_equal(ix.cpu().numpy(), ixr)
        assert np.array_equal(m1.cpu().numpy(), m1r) and np.array_equal(m2.cpu().numpy(), m2r)


def test_oversized_workspace_is_valid_for_smaller_calls(nm, oracle, cuda):
    """A workspace sized for (16384, 16384) must serve any smaller call (bench.py does exactly that): the chunk count of
    the grid plan depends on the actual sizes, so the workspace bound has to cover the worst plan."""
    import torch
    ws = nm.MatchWorkspace(16384, 16384, cuda)
    for na, nb in [(300, 2500), (5000, 140), (16384, 130), (2591, 2600), (1, 16384)]:
        A = H.synth.descriptors(na + 1, na)
        B = H.synth.descriptors(nb + 2, nb)
        ref, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False)
        res, _ = nm.sift_match(_t(A, cuda), _t(B, cuda), 0.8, workspace=ws)
        torch.cuda.synchronize()
        assert np.array_equal(res.cpu().numpy(), ref), (na, nb)


def test_match_batch_equals_single_matches(nm, oracle, cuda):
    """nm_sift_match_batch_f32: norms / finalize / fallback once for all pairs, the MFMA kernel once per pair. Pairs of
    different sizes incl. an empty one, near-ties that take the fallback, prior values that must survive."""
    import torch
    rng = np.random.default_rng(5)
    shapes = [(700, 900), (1, 1), (1300, 257), (0, 50), (2049, 1500), (300, 4000)]
    As = [H.synth.descriptors(300 + k, max(na, 1))[:na] for k, (na, nb) in enumerate(shapes)]
    Bs = [H.synth.descriptors(400 + k, max(nb, 1))[:nb] for k, (na, nb) in enumerate(shapes)]
    Bs[0][7] = As[0][3]; Bs[0][800] = As[0][3]              # duplicate candidates: tie on the lowest index
    Bs[4][10] = Bs[4][11]                                   # equal best and second best for some queries
    tA = [_t(a if len(a) else np.zeros((1, 128), np.float32), cuda) for a in As]
    tB = [_t(b if len(b) else np.zeros((1, 128), np.float32), cuda) for b in Bs]
    results = [torch.full((max(na, 1),), -9, dtype=torch.int32, device=cuda) for na, _ in shapes]
    nm.sift_match_batch(tA, tB, [s[0] for s in shapes], [s[1] for s in shapes], results, 0.8)
    torch.cuda.synchronize()
    for k, (na, nb) in enumerate(shapes):
        if na == 0 or nb == 0:
            assert bool((results[k] == -9).all())
            continue
        ref, _, _ = oracle.sift_matches(As[k], Bs[k], 0.8, want_distance=False, prior=np.full(na, -9, np.int32))
        assert np.array_equal(results[k][:na].cpu().numpy(), ref), (k, na, nb)
    # more pairs than the limit is refused
    many = [tA[0]] * (nm.MATCH_MAX_BATCH + 1)
    with pytest.raises(nm.NmError):
        nm.sift_match_batch(many, many, [1] * len(many), [1] * len(many), [results[0]] * len(many))


def test_match_adversarial_rounding(nm, oracle, cuda):
    """Inputs built to drive the MFMA formulation |a|^2 + |b|^2 - 2 a.b to its worst-case rounding behaviour, where a
    probabilistic error margin is not enough (matcher theorem): products that all round the same
    way, massive cancellation, near-ties closer than the formulation can resolve. Whatever route a row takes (MFMA
    candidates proven by the bound, or the exact fallback), (min1, index, min2) must equal the oracle's bit for bit."""
    import torch
    rng = np.random.default_rng(77)
    ulp = np.float32(2.0 ** -23)
    cases = {}
    # 1. constant vectors: every one of the 128 products of a row pair is the same number and rounds the same way
    ca = (1.0 + rng.integers(0, 4096, 600) * 2.0 ** -12).astype(np.float32)
    cb = (1.0 + rng.integers(0, 4096, 900) * 2.0 ** -12).astype(np.float32)
    cases["constant vectors"] = (np.repeat(ca[:, None], 128, 1), np.repeat(cb[:, None], 128, 1))
    # 2. powers of two plus one ulp: the exact products end in ...01 x ...01 patterns that never round to even
    ea = rng.integers(-3, 4, (500, 128)); eb = rng.integers(-3, 4, (700, 128))
    cases["2^e (1 + ulp)"] = ((2.0 ** ea * (1.0 + ulp)).astype(np.float32), (2.0 ** eb * (1.0 + 3 * ulp)).astype(np.float32))
    # 3. cancellation: norms ~1.3e8, distances ~1e-2 .. 1e2 -- the formulation's absolute error exceeds every distance
    base = np.full((1, 128), 1000.0, np.float32)
    cases["large offset, tiny distances"] = ((base + rng.standard_normal((400, 128)) * 0.05).astype(np.float32),
                                             (base + rng.standard_normal((1100, 128)) * 0.05).astype(np.float32))
    # 4. ties everywhere: all candidates identical, and a block of exact copies of the queries
    A4 = H.synth.descriptors(5, 300); B4 = np.repeat(H.synth.descriptors(6, 1), 513, 0)
    cases["all candidates equal"] = (A4, B4)
    B5 = np.concatenate([A4[::-1], A4[::-1], H.synth.descriptors(9, 200)])       # every query twice in B: min2 == 0
    cases["duplicated exact copies"] = (A4, B5)
    # 5. one-ulp ladders: candidate j differs from the query in one coordinate by j ulps -> distances 0, tiny, tiny, ...
    A6 = (H.synth.descriptors(7, 64) * 255).astype(np.float32)
    B6 = np.repeat(A6, 9, 0)
    for j in range(B6.shape[0]):
        B6[j, (5 * j) % 128] = np.nextafter(B6[j, (5 * j) % 128], np.float32(1e9), dtype=np.float32) if j % 9 else B6[j, (5 * j) % 128]
        for _ in range(j % 9 - 1 if j % 9 > 1 else 0):
            B6[j, (5 * j) % 128] = np.nextafter(B6[j, (5 * j) % 128], np.float32(1e9), dtype=np.float32)
    cases["one-ulp ladders"] = (A6, B6)
    # 6. bf16x3 screen: every element is 2^e (1 + 2^-9 + 2^-17 - 2^-23): its two-piece bf16 split leaves the largest
    #    residual (~2^-17 relative), with the same sign in every element of every row, so the screen's dot products are
    #    all off in the same direction by ~2^-16 relative (4e-3 absolute here) while the candidates of a query differ by
    #    ~1e-6: only the proven bound (-> exact fallback) can get these rows right
    bad = np.float32(1.0 + 2.0 ** -9 + 2.0 ** -17 - 2.0 ** -23)
    A7 = (bad * 2.0 ** rng.integers(-1, 2, (200, 128))).astype(np.float32)
    B7 = np.repeat(A7, 4, 0)
    for j in range(B7.shape[0]):
        B7[j, (3 * j) % 128] += np.float32((1 + j % 4) * 2.0 ** -11)
    cases["aligned bf16 split residuals"] = (A7, np.concatenate([B7, (bad * 2.0 ** rng.integers(-1, 2, (300, 128))).astype(np.float32)]))
    # 7. two-stage screen: every element is 2^e (1 + 2^-11 - 2^-23), the value just below the midpoint of two fp16 numbers:
    #    the fp16 images lose ~2^-11 relative in EVERY element, all in the same direction, so the coarse pass's dot
    #    products are off by ~1e-3 relative while the candidates of a query differ by ~1e-7: only the residual-norm bound
    #    (-> second pass) can get these rows right
    bad16 = np.float32(1.0 + 2.0 ** -11 - 2.0 ** -23)
    A8 = (bad16 * 2.0 ** rng.integers(-1, 2, (200, 128))).astype(np.float32)
    B8 = np.repeat(A8, 4, 0)
    for j in range(B8.shape[0]):
        B8[j, (3 * j) % 128] += np.float32((1 + j % 4) * 2.0 ** -12)
    cases["aligned fp16 rounding residuals"] = (A8, np.concatenate([B8, (bad16 * 2.0 ** rng.integers(-1, 2, (300, 128))).astype(np.float32)]))
    # 8. magnitudes at and beyond the edges of the fp16 range: elements below its smallest subnormal (the images are all
    #    zeros), in its subnormal range, and beyond its largest number (2 |x| > 65504: outside the coarse pass's domain)
    D8a, D8b = H.synth.descriptors(21, 300), H.synth.descriptors(22, 700)
    for nm_, sc in (("below fp16 subnormals", 1e-9), ("fp16 subnormal range", 3e-6), ("beyond the fp16 range", 4.0e4),
                    ("mixed magnitudes", None)):
        if sc is None:
            sa = (10.0 ** rng.uniform(-7, 4.7, (300, 1))).astype(np.float32); sb = (10.0 ** rng.uniform(-7, 4.7, (700, 1))).astype(np.float32)
            cases[nm_] = (D8a * sa, D8b * sb)
        else:
            cases[nm_] = (D8a * np.float32(sc), D8b * np.float32(sc))
    for name, (A, B) in cases.items():
        A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.float32)
        prior = np.full(len(A), -7, np.int32)
        ref, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False, prior=prior)
        m1r, ixr, m2r = oracle.sift_match_shard(A, B, 0)
        got, _ = _match(nm, cuda, A, B, prior=prior)
        assert np.array_equal(got, ref), name
        ws = nm.MatchWorkspace(len(A), len(B), cuda)
        m1, ix, m2 = nm.sift_match_shard(_t(A, cuda), _t(B, cuda), 0, workspace=ws)
        torch.cuda.synchronize()
        assert np.array_equal(ix.cpu().numpy(), ixr), name
        assert np.array_equal(m1.cpu().numpy().view(np.uint32), m1r.view(np.uint32)), name
        assert np.array_equal(m2.cpu().numpy().view(np.uint32), m2r.view(np.uint32)), name
        if name == "large offset, tiny distances":        # the bound must refuse to trust the MFMA pass here
            assert nm.match_fallback_count(ws, len(A), len(B)) == len(A)
        if nm.get_match_screen() == "f16" and name in ("aligned fp16 rounding residuals", "beyond the fp16 range"):
            assert nm.match_second_pass_count(ws, len(A), len(B)) == len(A), name     # the coarse pass must not be trusted


def test_shard_with_empty_candidate_shard(nm, oracle, cuda):
    """world > nB or uneven tiny sets: a rank's shard can be empty. Its triple must be neutral for the merge."""
    import torch
    A = H.synth.descriptors(1, 50)
    B = H.synth.descriptors(2, 3)
    ref, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False)
    m1s, ixs, m2s = [], [], []
    for b, e in [(0, 0), (0, 2), (2, 2), (2, 3)]:                       # shards 0 and 2 are empty
        Bs = B[b:e] if e > b else np.zeros((1, 128), np.float32)
        t = nm.sift_match_shard(_t(A, cuda), _t(Bs, cuda)[: e - b], b)
        m1s.append(t[0]); ixs.append(t[1]); m2s.append(t[2])
    res = nm.sift_match_merge(torch.stack(m1s), torch.stack(ixs), torch.stack(m2s), 0.8)
    torch.cuda.synchronize()
    assert np.array_equal(res.cpu().numpy(), ref)


def test_native_allgather_entry_and_packed_merge(nm, oracle, cuda):
    """nm_sift_match_allgather_f32 (native multi-GPU entry). One rank: shard + merge without a communicator must equal
    the unsharded match. Several ranks are emulated by laying the per-rank (min1, idx, min2) blocks out exactly as
    ncclAllGather would (rank-major) and running the merge step on it: ascending rank order = lowest global index wins
    a cross-shard tie (Q14, match.cu:94-105)."""
    import torch
    A = H.synth.descriptors(1, 1500)
    B = H.synth.descriptors(2, 2100)
    B[100] = B[1900]                                     # cross-shard duplicate: the lower global index must win
    A[7] = B[1900] + np.float32(1e-3)
    ref, _, _ = oracle.sift_matches(A, B, 1.5, want_distance=False)      # ambiguity 1.5: the tie (ratio 1) is reported
    tA, tB = _t(A, cuda), _t(B, cuda)
    lib = nm.lib()
    res = torch.full((1500,), -1, dtype=torch.int32, device=cuda)
    ws = torch.empty(lib.nm_sift_match_allgather_workspace_bytes(1500, 2100, 1), dtype=torch.uint8, device=cuda)
    assert lib.nm_sift_match_allgather_f32(tA.data_ptr(), 1500, tB.data_ptr(), 2100, 0, 1, res.data_ptr(), 1.5,
                                           ws.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(res.cpu().numpy(), ref) and ref[7] == 100
    # more than one rank without a communicator is refused, not silently wrong
    assert lib.nm_sift_match_allgather_f32(tA.data_ptr(), 1500, tB.data_ptr(), 2100, 0, 2, res.data_ptr(), 1.5,
                                           ws.data_ptr(), None, None) != 0
    bounds = [0, 700, 700, 1400, 2100]                   # four "ranks", the second one with an empty shard
    blocks = []
    for g in range(4):
        b, e = bounds[g], bounds[g + 1]
        m1, ix, m2 = nm.sift_match_shard(tA, tB[b:e], b)
        blocks.append(torch.stack([m1.view(torch.int32), ix, m2.view(torch.int32)]))
    packed = torch.stack(blocks).contiguous()            # (rank, 3, nA) = the all-gather's receive buffer
    res2 = torch.full((1500,), -1, dtype=torch.int32, device=cuda)
    assert lib.nm_sift_match_merge_packed_f32(packed.data_ptr(), 4, 1500, res2.data_ptr(), 1.5, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(res2.cpu().numpy(), ref)


def test_match_batch_dev_sizes_sweep(nm, oracle, cuda):
    """Device-sized entry on ragged sizes under one capacity: the plan is made on the device for the real sizes (plain
    order for small sets, XCD-grouped for large ones), sizes above the capacity are clipped, empty sets and negative
    counts make the pair a no-op (prior untouched), near-ties still reach the exact fallback."""
    import torch
    cap_a, cap_b = 4096, 5000
    shapes = [(700, 900), (1, 1), (1300, 257), (0, 50), (4096, 5000), (300, 4000), (50, 0), (4096, 130), (2049, 1500)]
    rng = np.random.default_rng(11)
    n = len(shapes)
    As = [H.synth.descriptors(500 + k, cap_a) for k in range(n)]
    Bs = [H.synth.descriptors(600 + k, cap_b) for k in range(n)]
    Bs[0][7] = As[0][3]; Bs[0][800] = As[0][3]              # duplicate candidates: tie on the lowest index
    for i in range(40):                                     # near-ties closer than the screen can resolve -> fallback
        for c in range(3):
            v = As[8][i].copy()
            v[(7 * i + c) % 128] += np.float32(0.25)
            v[(11 * i + 5 * c) % 128] += np.float32(1e-6 * c)
            Bs[8][3 * i + c] = v
    tA = [_t(a, cuda) for a in As]
    tB = [_t(b, cuda) for b in Bs]
    sizes_a = [s[0] for s in shapes]; sizes_b = [s[1] for s in shapes]
    dA = _t(np.array(sizes_a, np.int32), cuda)
    dB = _t(np.array(sizes_b, np.int32), cuda)
    results = [torch.full((cap_a,), -9, dtype=torch.int32, device=cuda) for _ in shapes]
    ws = nm.MatchBatchDevWorkspace(n, cap_a, cap_b, cuda)
    nm.sift_match_batch_dev(tA, [dA[k:k + 1] for k in range(n)], tB, [dB[k:k + 1] for k in range(n)], results, 0.8,
                            workspace=ws, capA=cap_a, capB=cap_b)
    torch.cuda.synchronize()
    for k, (na, nb) in enumerate(shapes):
        if na == 0 or nb == 0:
            assert bool((results[k] == -9).all()), k
            continue
        ref, _, _ = oracle.sift_matches(As[k][:na], Bs[k][:nb], 0.8, want_distance=False, prior=np.full(na, -9, np.int32))
        assert np.array_equal(results[k][:na].cpu().numpy(), ref), (k, na, nb)
        assert bool((results[k][na:] == -9).all()), (k, na, nb)
    # the SAME call with other sizes in the same device ints (what a HIP graph replay sees): clipping and negatives
    dA.copy_(_t(np.array([cap_a + 77, 5, -3, 10, 1000, 4096, 9, 77, 256], np.int32), cuda))
    dB.copy_(_t(np.array([cap_b + 1, 3, 10, -1, 5000, 128, 129, 5000, 128], np.int32), cuda))
    for r in results:
        r.fill_(-9)
    nm.sift_match_batch_dev(tA, [dA[k:k + 1] for k in range(n)], tB, [dB[k:k + 1] for k in range(n)], results, 0.8,
                            workspace=ws, capA=cap_a, capB=cap_b)
    torch.cuda.synchronize()
    sa = np.clip(dA.cpu().numpy(), 0, cap_a); sb = np.clip(dB.cpu().numpy(), 0, cap_b)
    for k in range(n):
        na, nb = int(sa[k]), int(sb[k])
        if na == 0 or nb == 0:
            assert bool((results[k] == -9).all()), k
            continue
        ref, _, _ = oracle.sift_matches(As[k][:na], Bs[k][:nb], 0.8, want_distance=False, prior=np.full(na, -9, np.int32))
        assert np.array_equal(results[k][:na].cpu().numpy(), ref), (k, na, nb)
        assert bool((results[k][na:] == -9).all()), (k, na, nb)
    assert nm.lib().nm_sift_match_batch_dev_f32(n, None, None, None, None, cap_a, cap_b, None, 0.8, None, None) != 0


def test_match_outside_the_comfortable_domain(nm, oracle, cuda):
    """Exact set_matches semantics (match.cu:88-116) where round 2 deviated: distances above 2139095040 (min_2 is
    overwritten at every replacement, :97, so the initial value only survives while the minimum sits at candidate 0),
    M = 1, and non-finite inputs (a NaN in candidate 0 turns every row into -1; a NaN elsewhere is skipped by the scan's
    comparisons). Fused matcher (both screens), materialised distance + get_sift_matches, shard triples and the merge for
    several shard splits incl. empty shards: everything equal to the oracle, whose scan is checked against a literal
    transcription of the reference loop on the CPU (tests/test_oracle_match_semantics.py)."""
    import torch
    from test_oracle_match_semantics import _cases
    u32 = lambda a: np.ascontiguousarray(a).view(np.uint32)
    for name, (A, B) in _cases().items():
        for amb in (0.8, 1.5):
            prior = np.full(len(A), -7, np.int32)
            ref, Dref, _ = oracle.sift_matches(A, B, amb, want_distance=True, prior=prior)
            got, D = _match(nm, cuda, A, B, amb=amb, want_distance=True, prior=prior)
            assert np.array_equal(got, ref), (name, amb)
            if nm.get_distance_mode() == "exact":
                nan = np.isnan(Dref)                     # a NaN's sign / payload is not specified; everything else bit for bit
                assert np.array_equal(np.isnan(D), nan) and np.array_equal(u32(D)[~nan], u32(Dref)[~nan]), name
            else:
                H.assert_distance(nm, D, Dref, name)
            res = nm.get_sift_matches(_t(Dref, cuda), amb, prior=_t(prior, cuda))
            assert np.array_equal(res.cpu().numpy(), ref), (name, amb, "get_sift_matches")
        m1r, ixr, m2r = oracle.sift_match_shard(A, B, 0)
        m1, ix, m2 = nm.sift_match_shard(_t(A, cuda), _t(B, cuda), 0)
        assert np.array_equal(ix.cpu().numpy(), ixr), name
        assert np.array_equal(u32(m1.cpu().numpy()) & 0x7fffffff, u32(m1r) & 0x7fffffff), name     # NaN: sign bit is free
        assert np.array_equal(u32(m2.cpu().numpy()), u32(m2r)), name
        ref, _, _ = oracle.sift_matches(A, B, 1.5, want_distance=False, prior=prior)
        for fr in ([0, 0.34, 1.0], [0, 0, 0.5, 0.5, 1.0], [0, 0.04, 1.0]):
            bounds = [int(round(f * len(B))) for f in fr]
            tr = []
            for b, e in zip(bounds[:-1], bounds[1:]):
                Bs = B[b:e] if e > b else np.zeros((1, 128), np.float32)
                tr.append(nm.sift_match_shard(_t(A, cuda), _t(Bs, cuda)[: e - b], b))
            res = nm.sift_match_merge(torch.stack([t[0] for t in tr]), torch.stack([t[1] for t in tr]),
                                      torch.stack([t[2] for t in tr]), 1.5, prior=_t(prior, cuda))
            torch.cuda.synchronize()
            assert np.array_equal(res.cpu().numpy(), ref), (name, bounds)


def test_nonfinite_pair_does_not_disturb_its_batch(nm, oracle, cuda):
    """A pair with NaN descriptors inside batched calls (host- and device-sized): that pair follows the reference's scan
    (through the exact fallback), the other pairs of the call are unaffected."""
    import torch
    rng = np.random.default_rng(3)
    As = [H.synth.descriptors(700 + k, 600) for k in range(3)]
    Bs = [H.synth.descriptors(800 + k, 900) for k in range(3)]
    Bs[1][0, 3] = np.nan                      # every row of pair 1 -> -1
    As[2][5, 7] = np.inf; Bs[2][17, 0] = np.nan
    tA = [_t(a, cuda) for a in As]; tB = [_t(b, cuda) for b in Bs]
    refs = [oracle.sift_matches(As[k], Bs[k], 0.8, want_distance=False, prior=np.full(600, -9, np.int32))[0] for k in range(3)]
    assert (refs[1] == -1).all() and (refs[0] >= 0).any()
    results = [torch.full((600,), -9, dtype=torch.int32, device=cuda) for _ in range(3)]
    nm.sift_match_batch(tA, tB, [600] * 3, [900] * 3, results, 0.8)
    torch.cuda.synchronize()
    for k in range(3):
        assert np.array_equal(results[k].cpu().numpy(), refs[k]), k
    results = [torch.full((600,), -9, dtype=torch.int32, device=cuda) for _ in range(3)]
    nA = _t(np.array([600, 600, 600], np.int32), cuda); nB = _t(np.array([900, 900, 900], np.int32), cuda)
    nm.sift_match_batch_dev(tA, [nA[k:k + 1] for k in range(3)], tB, [nB[k:k + 1] for k in range(3)], results, 0.8)
    torch.cuda.synchronize()
    for k in range(3):
        assert np.array_equal(results[k].cpu().numpy(), refs[k]), k


def test_match_phases_equal_the_whole_call(nm, oracle, cuda):
    """nm_sift_match_batch_dev_phases_f32: PREP, SCREEN and FINISH issued as three calls (on one stream here) give what the
    whole call gives; a phase mask outside 1..7 is refused."""
    import torch
    A = [_t(H.synth.descriptors(900 + k, 800), cuda) for k in range(2)]
    B = [_t(H.synth.descriptors(950 + k, 1100), cuda) for k in range(2)]
    nA = _t(np.array([800, 333], np.int32), cuda); nB = _t(np.array([1100, 999], np.int32), cuda)
    args = (A, [nA[k:k + 1] for k in range(2)], B, [nB[k:k + 1] for k in range(2)])
    whole = [torch.full((800,), -3, dtype=torch.int32, device=cuda) for _ in range(2)]
    ws = nm.MatchBatchDevWorkspace(2, 800, 1100, cuda)
    nm.sift_match_batch_dev(*args, whole, 0.8, workspace=ws)
    parts = [torch.full((800,), -3, dtype=torch.int32, device=cuda) for _ in range(2)]
    ws2 = nm.MatchBatchDevWorkspace(2, 800, 1100, cuda)
    for ph in (nm.MATCH_PHASE_PREP, nm.MATCH_PHASE_SCREEN, nm.MATCH_PHASE_FINISH):
        nm.sift_match_batch_dev(*args, parts, 0.8, workspace=ws2, phases=ph)
    torch.cuda.synchronize()
    for k, (na, nb) in enumerate(((800, 1100), (333, 999))):
        assert torch.equal(whole[k], parts[k])
        ref, _, _ = oracle.sift_matches(A[k][:na].cpu().numpy(), B[k][:nb].cpu().numpy(), 0.8, want_distance=False,
                                        prior=np.full(na, -3, np.int32))
        assert np.array_equal(whole[k][:na].cpu().numpy(), ref)
    with pytest.raises(nm.NmError):
        nm.sift_match_batch_dev(*args, parts, 0.8, workspace=ws2, phases=8)


@pytest.mark.parametrize("instruction,screen", [(1, 2), (0, 1)], ids=["f16_coarse", "bf16x3"])
def test_mfma_rounding_model_is_inside_what_the_screens_budget(nm, cuda, instruction, screen):
    """The premise under the matcher's exactness (protects kernels/match.cu:83-117): the error of the
    fp32 accumulation INSIDE v_mfma_f32_32x32x16_{bf16,f16} is a hardware model, not IEEE arithmetic. nm_selftest_mfma_model
    measures it on THIS device -- layout, directed cases, 2^20 random instructions, and the screens' own chains (bf16 norm
    k-slot instruction with C = 0, then 8 f16 or 24 bf16 instructions into the same accumulator) on adversarial rows incl.
    fp16-subnormal operands alone and mixed with normal ones (a pipe that flushed them would read ~0.25 here) -- against
    binary64. Asserted: at most HALF of what screen_err_coeff budgets for the accumulation (nm_sift_match_accum_budget)."""
    r = nm.selftest_mfma_model(instruction, n_random=1 << 20, n_chains=8192)
    budget = nm.match_accum_budget(screen)
    u = 2.0 ** -24
    assert r["layout_mismatches"] == 0
    assert r["instructions"] >= (1 << 20) and r["chain_launches"] >= 8192
    # the dot product is formed before C is added, and the final rounding is to nearest (model H)
    assert r["c1_plus_16_small_ulp"] == 4 and r["c1_plus_one_small_ulp"] == 1 and r["c2p24_plus_16"] == 16
    # one instruction, random operands, against the model H (u |D| + 7 u (pmax_lo + pmax_hi)): MI355X reads 1.7-1.9 (the
    # three-way sum of the halves and C loses up to an ulp); a chain of n instructions is then off by at most
    # model_ratio (n u + 7 u) 1.01 (sqrt na + sqrt nb)^2 -- which must stay below HALF the budget
    n_instr = 9 if instruction == 1 else 25
    assert 0 < r["model_ratio"] <= 2.0, r
    assert r["model_ratio"] * (n_instr + 7) * u * 1.01 <= 0.5 * budget, (r, budget)
    # the chains as issued: measured coefficient <= half of the budgeted one, subnormal operands included
    assert 0 < r["chain_coeff"] <= 0.5 * budget, (r, budget)
    assert r["chain_coeff_subnormal"] <= 0.5 * budget, (r, budget)
    # same-half truncation (what distinguishes H from exact-then-round): the 0.992-ulp product is cut away
    assert r["same_half_truncation_ulp"] == 0


def test_distance_on_the_mfma_every_entry_within_1e_4(nm, oracle, cuda, screen):
    """nm_sift_match_f32's `distance` filled by the fp32 MFMA pass (distance_mfma_kernel; reference: match.cu:14-80 through
    siftfunctions.cu:28-34): EVERY entry within 1e-4 relative of the reference's chain, on the families that stress the bound --
    uniform rows (all-positive: the centring is what keeps them off the list), real SIFT descriptors with true matches, near
    duplicates at 1e-3 ... 1e-7, exact copies (distance exactly 0), one-hot rows, magnitudes 1e6 and 1e-6, rows that differ in
    one element, ragged sizes around the 128-row / 256-column / 32-column tile edges, and too few rows for the pass's scratch
    (exact kernel). The match indexes are the oracle's in every case."""
    import torch
    if screen == "bf16x3":
        pytest.skip("this screen's parametrisation runs the exact distance kernel")
    rng = np.random.default_rng(77)

    def run(A, B, what, amb=0.8, expect_listed=None):
        ref, Dref, _ = oracle.sift_matches(A, B, amb)
        ws = nm.MatchWorkspace(len(A), len(B), cuda)
        res, D = nm.sift_match(_t(A, cuda), _t(B, cuda), amb, want_distance=True, workspace=ws)
        torch.cuda.synchronize()
        assert np.array_equal(res.cpu().numpy(), ref), what
        H.assert_distance(nm, D, Dref, what)
        listed, cap = nm.match_distance_listed(ws, len(A), len(B))
        if expect_listed is not None:
            assert expect_listed(listed, cap), (what, listed, cap)
        return listed, cap, D.cpu().numpy(), Dref

    A, B = H.synth.descriptors(1, 1000), H.synth.descriptors(2, 777)
    n_blocks = ((1000 + 31) // 32) * ((777 + 31) // 32)
    run(A, B, "uniform", expect_listed=lambda n, cap: 0 <= n < n_blocks // 4)
    # real descriptors, second frame = shifted copy: thousands of true matches (near-duplicates of large norm)
    f0 = H.blurred_frame(0, 640, 480)
    f1 = np.roll(f0, (2, 3), axis=(0, 1)).copy()
    a = oracle.sift_detect_describe(f0, 8192)["desc"]
    b = oracle.sift_detect_describe(f1, 8192)["desc"]
    run(a, b, "sift + shifted copy", expect_listed=lambda n, cap: 0 < n <= cap)
    # near-duplicates over five decades + exact copies: the listed entries must come out of the reference's own chain
    A = H.synth.descriptors(3, 700) * 50
    B = H.synth.descriptors(4, 900) * 50
    for q, eps in enumerate((1e-3, 1e-4, 1e-5, 1e-6, 1e-7)):
        B[100 * q:100 * q + 60] = A[100 * q:100 * q + 60] * np.float32(1 + eps)
    B[600:650] = A[600:650]                                 # exact copies: distance exactly 0
    B[700] = B[701]
    listed, cap, D, Dref = run(A, B, "near duplicates", amb=1.5, expect_listed=lambda n, cap: 0 < n <= cap)
    assert (D[600:650, 600:650].diagonal() == 0).all() and (Dref[600:650, 600:650].diagonal() == 0).all()
    d = np.arange(60)
    assert np.array_equal(D[d, d].view(np.uint32), Dref[d, d].view(np.uint32))      # listed entries: bit-equal to the chain
    # scale families and structured rows
    run((A * 2e4).astype(np.float32), (B * 2e4).astype(np.float32), "magnitude 1e6", amb=1.5)
    run((A * 2e-8).astype(np.float32), (B * 2e-8).astype(np.float32), "magnitude 1e-6", amb=1.5)
    oh_a = np.zeros((300, 128), np.float32); oh_a[np.arange(300), rng.integers(0, 128, 300)] = 1
    oh_b = np.zeros((400, 128), np.float32); oh_b[np.arange(400), rng.integers(0, 128, 400)] = 1
    run(oh_a, oh_b, "one-hot rows (ties everywhere)", amb=1.5)
    base = H.synth.descriptors(5, 1)[0]
    one_a = np.tile(base, (260, 1)); one_a[np.arange(260), np.arange(260) % 128] += rng.random(260).astype(np.float32)
    one_b = np.tile(base, (300, 1)); one_b[np.arange(300), (np.arange(300) * 7) % 128] -= rng.random(300).astype(np.float32)
    run(one_a, one_b, "rows that differ in one element", amb=1.5)
    # tile edges
    for na, nb in ((128, 256), (129, 257), (127, 255), (385, 31), (64, 33), (61, 1), (1000, 1), (513, 1025)):
        run(H.synth.descriptors(10 + na, na), H.synth.descriptors(20 + nb, nb), "shape %d x %d" % (na, nb))
    # fewer query rows than the pass needs scratch for: the exact kernel, bit for bit
    A, B = H.synth.descriptors(6, 20), H.synth.descriptors(7, 500)
    listed, cap, D, Dref = run(A, B, "tiny query set", expect_listed=lambda n, cap: n == -1)
    assert np.array_equal(D.view(np.uint32), Dref.view(np.uint32))
    # non-finite rows: everything is listed and comes out of the chain
    A, B = H.synth.descriptors(8, 300), H.synth.descriptors(9, 300)
    B[17, 5] = np.nan; A[3, 100] = np.inf
    run(A, B, "NaN / inf rows")


def test_distance_with_one_nan_takes_the_exact_kernel(nm, cuda, screen):
    """ONE NaN in B makes the mean row NaN, the acceptance test fails everywhere and every 32 x 32 block is listed.
    Beyond a quarter of the blocks the list counts as overflowed and exact_distance_kernel fills the matrix (the per-block fix-up
    -- one wave per block -- would take many times longer): the result is bit-equal to NM_MATCH_DISTANCE=exact, and the call
    costs about what the exact mode costs, at the bench's 12k x 12k."""
    import time
    import torch
    if screen != "f16":
        pytest.skip("one run is enough")
    nA, nB = 12223, 12080
    A = _t(H.synth.descriptors(41, nA) * 100, cuda)
    Bh = H.synth.descriptors(42, nB) * 100
    Bh[4321, 77] = np.nan
    B = _t(Bh, cuda)
    ws = nm.MatchWorkspace(nA, nB, cuda)
    before = nm.get_distance_mode()
    out = {}
    try:
        for mode in ("exact", "mfma"):
            nm.set_distance_mode(mode)
            for _ in range(2):
                res, D = nm.sift_match(A, B, 0.8, want_distance=True, workspace=ws)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                res, D = nm.sift_match(A, B, 0.8, want_distance=True, workspace=ws)
            torch.cuda.synchronize()
            out[mode] = ((time.perf_counter() - t0) / 5, res.clone(), D.clone())
            if mode == "mfma":
                listed, cap = nm.match_distance_listed(ws, nA, nB)
                blocks = ((nA + 31) // 32) * ((nB + 31) // 32)
                assert cap == blocks // 4 and listed > cap, (listed, cap, blocks)       # every block listed: counted as overflow
    finally:
        nm.set_distance_mode(before)
    assert torch.equal(out["mfma"][1], out["exact"][1])
    assert torch.equal(out["mfma"][2].view(torch.int32), out["exact"][2].view(torch.int32))       # NaN column included, bit for bit
    assert bool(torch.isnan(out["mfma"][2][:, 4321]).all())
    assert out["mfma"][0] < 2.0 * out["exact"][0], (out["mfma"][0], out["exact"][0])


def test_fp32_mfma_rounding_is_inside_what_the_fp32_bounds_assume(nm, cuda, screen):
    """The hardware premise under the fp32 screen's bound (since round 1) and under the materialised distance pass's acceptance
    test (round 5): v_mfma_f32_32x32x2_f32 forms C + a0 b0 + a1 b1 with exact products and at most two roundings, so a chain of
    n instructions errs like a 2n-step fma chain. Measured on the device against binary64: single instructions, and both
    accumulation forms exactly as the kernels issue them on row families incl. near-duplicates and mixed binades."""
    if screen != "f32":
        pytest.skip("one run is enough")
    m = nm.selftest_mfma_f32(1 << 22, 8192)
    u = 2.0 ** -24
    assert m["results"] >= 1 << 22
    assert m["rel_u"] <= 2.0 + 1e-3, m                      # two roundings at most
    # On MI355X EVERY result equals fma(a1, b1, fma(a0, b0, C)) (profiles/r05_n_mfma_f32_model.txt): the instruction IS the
    # two-step chain, so the bounds below are arithmetic. A device where that stops being true must still stay inside them.
    assert m["frac_fma_chain"] > 0.999 or m["rel_u"] <= 1.0 + 1e-3, m
    # the chains as issued, on adversarial rows (constant rows round every product alike and come close to the worst case):
    # inside the fma-chain worst cases gamma_67 / gamma_130, hence inside the budgets, which add the norms' share and slack
    assert 0 < m["two_chain_coeff"] <= 67 * u * 1.001 < nm.match_distance_budget(), m
    assert 0 < m["one_chain_coeff"] <= 130 * u * 1.001 <= 0.5 * nm.match_accum_budget(0) * 1.001, m
