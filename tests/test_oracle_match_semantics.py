"""The matcher's semantics outside the comfortable domain, on the CPU (oracle only): the restated scan
(nmo_sift_matches, match.cu:83-117) against (i) a literal per-row Python transcription of the scan on the oracle's own
distance matrix, and (ii) the multi-GPU decomposition nmo_sift_match_shard + nmo_sift_match_merge for several shard
splits -- on distances above 2139095040 (the scan OVERWRITES min_2 at every replacement, match.cu:97, so the initial
2139095040.0f only survives while the minimum sits at candidate 0), M = 1, non-finite inputs (a NaN distance is skipped
by the scan's comparisons except at candidate 0, where it stays for good) and exact duplicates (min2 == 0: untouched)."""
import numpy as np

C0 = np.float32(2139095040.0)


def _scan(D, amb, prior):
    """match.cu:88-116, line by line, on one distance matrix."""
    res = prior.copy()
    for i in range(D.shape[0]):
        m1 = D[i, 0]; m2 = C0; idx = 0
        for j in range(1, D.shape[1]):
            cur = D[i, j]
            if cur < m1:
                m2 = m1; idx = j; m1 = cur
            elif cur < m2:
                m2 = cur
        if m2 > 0:
            with np.errstate(all="ignore"):
                res[i] = idx if np.float32(m1) / np.float32(m2) < np.float32(amb) else -1
    return res


def _cases():
    rng = np.random.default_rng(1)
    A = rng.uniform(0, 1, (40, 128)).astype(np.float32)
    B = rng.uniform(0, 1, (30, 128)).astype(np.float32)
    out = {"plain": (A, B)}
    As, Bs = (A * 1e5).astype(np.float32), (B * 1e5).astype(np.float32)
    out["distances above 2139095040"] = (As, Bs)
    out["above 2139095040, minimum at candidate 0"] = (np.repeat(Bs[:1], 40, 0) + rng.uniform(0, 10, (40, 128)).astype(np.float32), Bs)
    out["M = 1, huge"] = (As, Bs[:1])
    Bn = B.copy(); Bn[0, 5] = np.nan
    out["NaN in candidate 0"] = (A, Bn)
    Bn = B.copy(); Bn[7, 5] = np.nan; Bn[20, 3] = np.inf
    out["NaN / inf in later candidates"] = (A, Bn)
    An = A.copy(); An[3, 9] = np.nan; An[5, 1] = np.inf
    out["NaN / inf in queries"] = (An, B)
    Bn = B.copy(); Bn[:, 0] = np.inf
    out["every distance inf"] = (A, Bn)
    Bn = B.copy(); Bn[0] = A[2]; Bn[5] = A[2]; Bn[1, 0] = np.nan
    out["min2 == 0 beside a NaN"] = (A, Bn)
    return out


def test_scan_restatement_and_shard_merge_agree_with_the_literal_scan(oracle):
    splits = [[0, 1.0], [0, 0.34, 1.0], [0, 0.04, 0.07, 1.0], [0, 0, 0.5, 0.5, 1.0], [0, 0.97, 1.0]]
    for name, (A, B) in _cases().items():
        for amb in (0.8, 1.5):
            prior = np.full(len(A), -7, np.int32)
            ref, D, _ = oracle.sift_matches(A, B, amb, want_distance=True, prior=prior)
            assert np.array_equal(ref, _scan(D, amb, prior)), (name, amb)
            assert np.array_equal(oracle.get_sift_matches(D, amb, prior=prior), ref), (name, amb)
            for fr in splits:
                bounds = [int(round(f * len(B))) for f in fr]
                tr = []
                for b, e in zip(bounds[:-1], bounds[1:]):
                    if e > b:
                        tr.append(oracle.sift_match_shard(A, B[b:e], b))
                    else:                                  # empty shard: the neutral triple
                        tr.append((np.full(len(A), np.inf, np.float32), np.full(len(A), -1, np.int32), np.full(len(A), np.inf, np.float32)))
                got = oracle.sift_match_merge(np.stack([t[0] for t in tr]), np.stack([t[1] for t in tr]),
                                              np.stack([t[2] for t in tr]), amb, prior=prior)
                assert np.array_equal(got, ref), (name, amb, bounds)


def test_the_cases_bite(oracle):
    c = _cases()
    A, B = c["distances above 2139095040"]
    _, D, _ = oracle.sift_matches(A, B, 0.8)
    assert D.min() > 2.2e9
    r08, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False)
    r15, _, _ = oracle.sift_matches(A, B, 1.5, want_distance=False)
    assert (r15 >= 0).sum() > (r08 >= 0).sum()              # ratios of true second minima, not of the clamp
    A, B = c["above 2139095040, minimum at candidate 0"]
    r, _, (m1, ix, m2) = oracle.sift_matches(A, B, 1.5, want_distance=False)
    assert (ix == 0).all() and (m2 == C0).all() and (r == 0).all()      # the initial min_2 survives: ratio tiny -> matched
    A, B = c["NaN in candidate 0"]
    assert (oracle.sift_matches(A, B, 0.8, want_distance=False)[0] == -1).all()
