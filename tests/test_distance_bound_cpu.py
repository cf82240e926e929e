"""CPU check of the arithmetic the MFMA distance pass rests on ("The materialised distance matrix";
reference: kernels/match.cu:14-80): a numpy model of distance_mfma_kernel -- rows centred in binary32, norms summed in binary32,
TWO fma chains (k = 0..63 behind the norm pair, k = 64..127 from zero) and one add, every step rounded to binary32 -- against
binary64.
  * |v - d'| <= DIST_C (sqrt nx + sqrt ny)^2 for the exact distance d' of the centred rows, on ordinary and adversarial families
    (constant rows round every product alike; near-duplicates cancel massively; mixed binades);
  * every entry the acceptance test keeps, v >= K (sqrt nx + sqrt ny)^2 with K = DIST_C / 7.8e-5, is within 1e-4 relative of the
    reference's own chain acc = fma(t, t, acc), t = a_k - b_k on the ORIGINAL rows -- the contract of nm_sift_match_f32's
    `distance`; everything else is recomputed by that chain on the device and need not be modelled;
  * the centring is what keeps rows with a large common component (an offset, a dominant mean descriptor) off the list.
The instruction-level premise (v_mfma_f32_32x32x2_f32 == two fused steps) is measured on the GPU (nm_selftest_mfma_f32); the
kernel itself is tested entry by entry against the oracle there too (tests/test_gpu_match.py)."""
import numpy as np
import pytest

F = np.float32
TOL = 1e-4


def _fma(a, b, c):
    """fma in binary32: the product of two floats is exact in binary64, the sum is rounded once to 53 bits and once more to 24 --
    a double rounding that differs from a true fma in ~2^-29 of the cases, by one ulp: irrelevant for a bound that has 15 % slack."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def _model(A, B, center=True):
    A, B = A.astype(F), B.astype(F)
    mu = (B.sum(0, dtype=F) / F(len(B))).astype(F) if center else np.zeros(128, F)
    X, Y = (A - mu).astype(F), (B - mu).astype(F)
    nx = (X * X).sum(1, dtype=F)
    ny = (Y * Y).sum(1, dtype=F)
    m2y = (F(-2) * Y).astype(F)
    a0 = _fma(np.ones_like(nx)[:, None] * nx[:, None], np.ones((1, len(Y)), F), np.zeros((len(X), len(Y)), F))   # nx * 1
    a0 = _fma(np.ones((len(X), 1), F), ny[None, :] * np.ones((len(X), 1), F), a0)                                 # + 1 * ny
    for k in range(64):
        a0 = _fma(X[:, k:k + 1] * np.ones((1, len(Y)), F), m2y[None, :, k] * np.ones((len(X), 1), F), a0)
    b0 = np.zeros_like(a0)
    for k in range(64, 128):
        b0 = _fma(X[:, k:k + 1] * np.ones((1, len(Y)), F), m2y[None, :, k] * np.ones((len(X), 1), F), b0)
    v = (a0 + b0).astype(F)
    X64, Y64 = X.astype(np.float64), Y.astype(np.float64)
    dc = (X64 * X64).sum(1)[:, None] + (Y64 * Y64).sum(1)[None, :] - 2.0 * (X64 @ Y64.T)        # exact d' (to 1e-16)
    return v.astype(np.float64), dc, nx.astype(np.float64), ny.astype(np.float64)


def _ref_chain(A, B):
    A, B = A.astype(F), B.astype(F)
    acc = np.zeros((len(A), len(B)), F)
    for k in range(128):
        t = (A[:, k:k + 1] - B[None, :, k]).astype(F)
        acc = _fma(t, t, acc)
    return acc.astype(np.float64)


CASES = {
    "uniform all-positive": lambda r: (r.uniform(0, 1, (96, 128)), r.uniform(0, 1, (160, 128))),
    "sift-like norms 70..950": lambda r: (r.uniform(0, 1, (96, 128)) * r.uniform(6, 85, (96, 1)), r.uniform(0, 1, (160, 128)) * r.uniform(6, 85, (160, 1))),
    "constant rows": lambda r: (np.full((64, 128), 0.75) + 2.0 ** -20 * r.integers(0, 4, (64, 1)), np.full((96, 128), 0.75) + 2.0 ** -20 * r.integers(0, 4, (96, 1))),
    "near duplicates": lambda r: ((lambda a: (a, np.concatenate([a[:64] * (1 + 2.0 ** -12 * r.uniform(-1, 1, (64, 128))), r.uniform(-0.5, 0.5, (64, 128))])))(r.uniform(-0.5, 0.5, (96, 128)))),
    "mixed binades": lambda r: (r.uniform(-0.5, 0.5, (96, 128)) * 2.0 ** r.integers(-12, 12, (1, 128)), r.uniform(-0.5, 0.5, (128, 128)) * 2.0 ** r.integers(-12, 12, (1, 128))),
    "magnitude 1e6": lambda r: (r.uniform(0, 1, (64, 128)) * 1e6, r.uniform(0, 1, (96, 128)) * 1e6),
}


def _constants(nm_lib):
    c = float(nm_lib.nm_sift_match_distance_budget())
    return c, c / 7.8e-5


@pytest.fixture(scope="module")
def consts():
    import niftymatch_amd
    return _constants(niftymatch_amd.lib())


@pytest.mark.parametrize("name", list(CASES))
def test_two_chain_value_is_inside_the_bound_and_accepted_entries_meet_1e_4(consts, name):
    dist_c, K = consts
    assert 4.4e-6 < dist_c < 6e-6 and abs(K - dist_c / 7.8e-5) < 1e-12
    A, B = CASES[name](np.random.default_rng(31))
    v, dc, nx, ny = _model(A, B)
    s2 = (np.sqrt(nx)[:, None] + np.sqrt(ny)[None, :]) ** 2
    assert (np.abs(v - dc) <= dist_c * s2 + 1e-300).all(), float((np.abs(v - dc) / np.maximum(s2, 1e-300)).max())
    kept = v >= K * s2 * (1 + 4e-6)                      # the kernel's tabulated terms are rounded UP: it keeps no more than this
    ref = _ref_chain(A, B)
    rel = np.abs(v - ref) / np.maximum(ref, 1e-300)
    assert (rel[kept] <= TOL).all(), (name, float(rel[kept].max()))
    if name in ("uniform all-positive", "magnitude 1e6"):
        assert kept.mean() > 0.99, kept.mean()           # the pass is useful, not just safe
    if name == "sift-like norms 70..950":                # rows of very different scale keep part of their own mean
        assert kept.mean() > 0.95, kept.mean()
    if name == "near duplicates":
        assert not kept[np.arange(64), np.arange(64)].any()      # every near-duplicate pair goes to the exact recomputation
        assert kept.mean() > 0.5


def test_the_centring_is_what_keeps_rows_with_a_common_offset_off_the_list(consts):
    """Rows that share a large common component (here uniform[0, 1) + 8 per element: squared norms ~9 000, distances ~21) have
    d / (sqrt nx + sqrt ny)^2 ~ 6e-4, far below K: uncentred, every entry would be listed; centred on the mean row they are the
    uniform rows again. (Plain uniform all-positive rows sit at 0.125 and pass either way since the two-chain form.)"""
    dist_c, K = consts
    r = np.random.default_rng(32)
    A, B = r.uniform(0, 1, (96, 128)) + 8.0, r.uniform(0, 1, (160, 128)) + 8.0
    v, dc, nx, ny = _model(A, B, center=False)
    s2 = (np.sqrt(nx)[:, None] + np.sqrt(ny)[None, :]) ** 2
    assert (np.abs(v - dc) <= dist_c * s2).all()
    assert (v >= K * s2).mean() < 0.01
    v, dc, nx, ny = _model(A, B, center=True)
    s2 = (np.sqrt(nx)[:, None] + np.sqrt(ny)[None, :]) ** 2
    assert (np.abs(v - dc) <= dist_c * s2).all()
    kept = v >= K * s2 * (1 + 4e-6)
    assert kept.mean() > 0.99
    ref = _ref_chain(A, B)
    assert (np.abs(v - ref)[kept] <= TOL * ref[kept]).all()
