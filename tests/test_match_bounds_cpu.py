"""CPU check of the arithmetic the two-stage matcher screen rests on: a numpy model of the coarse pass
(fp16 images, fp32 accumulation) against exact binary64 distances.
  * |value(j) - d(j)| <= E_j with E_j built from the MEASURED residual norms of the fp16 images (Cauchy-Schwarz), for
    ordinary, adversarial (every element just below the midpoint of two fp16 numbers) and tiny / large magnitudes;
  * the ratio-test shortcut of match_finalize_kernel (decide -1 / j1 from the two smallest screen values alone) never
    contradicts the exact scan on the rows it decides.
The kernels themselves are tested bit for bit on the GPU (tests/test_gpu_match.py); this pins the formulas they evaluate."""
import numpy as np
import pytest

ERR_COEFF = 3.2e-5          # screen_err_coeff(2): fp32 accumulation of the 134-term chain, with slack
KEY_SLOP = 1.6e-5           # key truncation (2^-17) + gamma_130


def _images(X, scale):
    """What prep_kernel<2> stores and measures: fp16(scale * x) (round to nearest even), the unscaled value it stands for, and
    an upper bound of the residual's 2-norm."""
    h = (X.astype(np.float32) * np.float32(scale)).astype(np.float16)
    xh = h.astype(np.float64) / scale
    res = np.sqrt(((X.astype(np.float64) - xh) ** 2).sum(1)) * 1.00002 + 1e-18
    return h, xh, res


def _coarse(A, B):
    ha, ah, ra = _images(A, -2.0)
    hb, bh, rb = _images(B, 1.0)
    na = (A.astype(np.float64) ** 2).sum(1).astype(np.float32)
    nb = (B.astype(np.float64) ** 2).sum(1).astype(np.float32)
    # exact products of the fp16 pieces, accumulated in binary32 (any order: the bound does not depend on it)
    dot = (ha.astype(np.float32) @ hb.astype(np.float32).T).astype(np.float32)
    value = (na[:, None] + nb[None, :] + dot).astype(np.float32)
    return value.astype(np.float64), na.astype(np.float64), nb.astype(np.float64), ra, rb


def _exact(A, B):
    A64, B64 = A.astype(np.float64), B.astype(np.float64)
    return (A64 * A64).sum(1)[:, None] + (B64 * B64).sum(1)[None, :] - 2.0 * (A64 @ B64.T)


def _bound(na, nb, ra, rb):
    sna, snb = np.sqrt(na)[:, None], np.sqrt(nb)[None, :]
    ra, rb = ra[:, None], rb[None, :]
    return (ERR_COEFF + KEY_SLOP) * (sna + snb) ** 2 * 1.0001 + 2.0 * (ra * (snb + rb) + (sna + ra) * rb + ra * rb) * 1.0001 + 1e-30


CASES = {
    "uniform": lambda r: (r.uniform(0, 1, (300, 128)), r.uniform(0, 1, (500, 128))),
    "sift-like norms 70..950": lambda r: (r.uniform(0, 1, (300, 128)) * r.uniform(6, 85, (300, 1)),
                                            r.uniform(0, 1, (500, 128)) * r.uniform(6, 85, (500, 1))),
    "aligned fp16 residuals": lambda r: ((1 + 2.0 ** -11 - 2.0 ** -23) * 2.0 ** r.integers(-1, 2, (200, 128)),
                                         (1 + 2.0 ** -11 - 2.0 ** -23) * 2.0 ** r.integers(-1, 2, (400, 128))),
    "fp16 subnormal range": lambda r: (r.uniform(0, 1, (200, 128)) * 3e-6, r.uniform(0, 1, (300, 128)) * 3e-6),
    "below fp16 subnormals": lambda r: (r.uniform(0, 1, (100, 128)) * 1e-9, r.uniform(0, 1, (100, 128)) * 1e-9),
    "large, inside the domain": lambda r: (r.uniform(0, 1, (200, 128)) * 2.5e3, r.uniform(0, 1, (300, 128)) * 2.5e3),
    "mixed magnitudes": lambda r: (r.uniform(0, 1, (300, 128)) * 10.0 ** r.uniform(-7, 3.3, (300, 1)),
                                   r.uniform(0, 1, (400, 128)) * 10.0 ** r.uniform(-7, 3.3, (400, 1))),
}


@pytest.mark.parametrize("name", list(CASES))
def test_coarse_value_is_within_its_bound(name):
    A, B = (np.ascontiguousarray(x, np.float32) for x in CASES[name](np.random.default_rng(5)))
    value, na, nb, ra, rb = _coarse(A, B)
    assert na.max() < 1e9 and nb.max() < 1e9                      # the coarse pass's domain (F16_NORM_LIMIT)
    err = np.abs(value - _exact(A, B))
    E = _bound(na, nb, ra, rb)
    assert (err <= E).all(), (name, float((err / E).max()))
    if name == "aligned fp16 residuals":                          # the bound is tight: this case comes close to it
        assert (err / E).max() > 0.5


@pytest.mark.parametrize("ambiguity", [0.8, 0.6, 1.0, 1.5])
def test_ratio_test_shortcut_never_contradicts_the_exact_scan(ambiguity):
    rng = np.random.default_rng(11)
    A = (rng.uniform(0, 1, (600, 128)) * rng.uniform(6, 85, (600, 1))).astype(np.float32)
    B = (rng.uniform(0, 1, (900, 128)) * rng.uniform(6, 85, (900, 1))).astype(np.float32)
    for k in range(0, 600, 7):                                    # planted near-duplicates: genuine matches
        B[rng.integers(1, 900)] = A[k] + rng.normal(0, 0.5, 128).astype(np.float32)
    value, na, nb, ra, rb = _coarse(A, B)
    D = _exact(A, B)
    order = np.argsort(value, axis=1, kind="stable")[:, :2]
    j1, j2 = order[:, 0], order[:, 1]
    rows = np.arange(len(A))
    v1, v2 = value[rows, j1], value[rows, j2]
    sna, snbm, rbm = np.sqrt(na), np.sqrt(nb.max()), rb.max()

    def E_of(bn, rbj):
        return (ERR_COEFF + KEY_SLOP) * (sna + bn) ** 2 * 1.0001 + 2.0 * (ra * (bn + rbj) + (sna * 1.000001 + ra) * rbj + ra * rbj) * 1.0001 + 1e-30

    def Et(X):
        bn = np.minimum(sna + np.sqrt(np.maximum(X, 0.0)) * 1.00001, snbm)
        return E_of(bn, np.minimum(rbm, 4.8829e-4 * bn + 7e-4))

    e1, e2 = E_of(np.sqrt(nb[j1]) * 1.000001, rb[j1]), E_of(np.sqrt(nb[j2]) * 1.000001, rb[j2])
    lo1, lo2, hi1 = v1 - Et(v1), v2 - Et(v2), v1 + e1
    hi2 = np.maximum(hi1, v2 + e2)
    no_match = (lo2 > 0) & (lo1 >= ambiguity * hi2 * 1.00001)
    match = ~no_match & (lo2 > 0) & (v2 > hi1 + Et(hi1)) & (ambiguity > 0) & (hi1 < ambiguity * lo2 * 0.99999) & (j1 > 0)
    # the exact scan (match.cu:88-116 on binary64 distances: the margins above dwarf the binary32 rounding of the real one)
    srt = np.sort(D, axis=1)
    m1, m2, idx = srt[:, 0], srt[:, 1], np.argmin(D, axis=1)
    exact = np.where(m1 / m2 < ambiguity, idx, -1)
    assert (exact[no_match] == -1).all()
    assert (exact[match] == j1[match]).all()
    assert (no_match | match).mean() > 0.8                        # and it decides most rows, which is its point
