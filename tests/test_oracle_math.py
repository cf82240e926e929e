"""Pins oracle/nmo_math.h (the fixed restatement of the CUDA device libm calls) against glibc/numpy in ulp."""
import numpy as np

from helpers import ulp_err


def test_atan2f_within_2p5_ulp(oracle):
    rng = np.random.default_rng(1)
    for scale in (300.0, 1e-3):
        x = rng.uniform(-scale, scale, 1_000_000).astype(np.float32)
        y = rng.uniform(-scale, scale, 1_000_000).astype(np.float32)
        ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
        assert ulp_err(oracle.vec("atan2f", y, x), ref) <= 2.5


def test_atan2f_axes(oracle):
    y = np.array([0, 0, 1, -1, 2, 2, -2, -2], np.float32)
    x = np.array([1, -1, 0, 0, 2, -2, 2, -2], np.float32)
    got = oracle.vec("atan2f", y, x)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64)).astype(np.float32)
    assert np.all(np.abs(got - ref) <= np.spacing(np.abs(ref)))
    assert got[0] == 0.0 and got[2] == np.float32(np.pi / 2)


def test_expf_within_1p5_ulp(oracle):
    rng = np.random.default_rng(2)
    e = rng.uniform(-20, 20, 1_000_000).astype(np.float32)
    assert ulp_err(oracle.vec("expf", e), np.exp(e.astype(np.float64))) <= 1.5
    assert oracle.vec("expf", np.zeros(1, np.float32))[0] == 1.0


def test_sinf_cosf_within_2_ulp(oracle):
    rng = np.random.default_rng(3)
    a = rng.uniform(-7, 7, 1_000_000).astype(np.float32)
    # ulp measured against the magnitude of the result is meaningless at the zeros; measure in absolute ulp of 1.0
    s = oracle.vec("sinf", a).astype(np.float64)
    c = oracle.vec("cosf", a).astype(np.float64)
    assert np.max(np.abs(s - np.sin(a.astype(np.float64)))) <= 2 * 2.0 ** -24
    assert np.max(np.abs(c - np.cos(a.astype(np.float64)))) <= 2 * 2.0 ** -24
    assert oracle.vec("sinf", np.zeros(1, np.float32))[0] == 0.0
    assert oracle.vec("cosf", np.zeros(1, np.float32))[0] == 1.0


def test_exp_exp2_double_within_2_ulp(oracle):
    rng = np.random.default_rng(4)
    d = rng.uniform(-20, 20, 500_000)
    r = oracle.vec("exp", d, dtype=np.float64)
    assert np.max(np.abs(r - np.exp(d)) / np.exp(d)) <= 2 * 2.2204e-16
    d = rng.uniform(-10, 10, 500_000)
    r = oracle.vec("exp2", d, dtype=np.float64)
    assert np.max(np.abs(r - np.exp2(d)) / np.exp2(d)) <= 2 * 2.2204e-16
    ints = np.arange(-8, 9).astype(np.float64)
    assert np.array_equal(oracle.vec("exp2", ints, dtype=np.float64), np.exp2(ints))


def test_set_threads_zero_restores_the_default(oracle):
    """bench.py's cpu_baseline lowers the oracle to one thread for its second leg and must get the default back
    (nmo_set_threads(0) was a no-op once: the all-thread figure of every later call would have been a 1-thread one)."""
    default = oracle.set_threads(0)
    assert default >= 1
    assert oracle.set_threads(1) == 1
    assert oracle.set_threads(0) == default
    if default >= 2:
        assert oracle.set_threads(2) == 2
        assert oracle.set_threads(-5) == default
