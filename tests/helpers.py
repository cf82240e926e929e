"""Shared test helpers: synthetic frames built with the oracle's Gaussian."""
import numpy as np

import oracle_lib as O
from niftymatch_amd import synth


def blurred_frame(seed, width, height, sigma=None):
    f = synth.noise_frame(seed, width, height)
    taps, r = O.create_kernel_for_sigma(synth.preblur_sigma(width, height) if sigma is None else sigma)
    return O.convolve(f, taps, r)[0]


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    u = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.max(np.abs(got.astype(np.float64) - ref64) / u)


def assert_distance(nm, D, Dref, what="distance matrix"):
    """The materialised `distance` of compute_sift_matches against the oracle's chain (match.cu:36-42). Mode "exact": bit for
    bit. Mode "mfma" (default; fp32 matrix cores on centred rows): EVERY entry within 1e-4 relative -- the tolerance the north
    star states for distance values -- which makes an exact zero exactly zero and keeps NaN / inf where the chain has them."""
    D = D.detach().cpu().numpy() if hasattr(D, "detach") else np.asarray(D)
    Dref = np.asarray(Dref)
    assert D.shape == Dref.shape, (what, D.shape, Dref.shape)
    if nm.get_distance_mode() == "exact":
        same = D.view(np.uint32) == Dref.view(np.uint32)
        assert same.all(), "%s: %d of %d elements differ" % (what, (~same).sum(), same.size)
        return
    fin = np.isfinite(Dref)
    assert np.array_equal(np.isnan(D), np.isnan(Dref)), what + ": NaN pattern"
    assert np.array_equal(D[~fin & ~np.isnan(Dref)], Dref[~fin & ~np.isnan(Dref)]), what + ": infinities"
    a, b = D[fin].astype(np.float64), Dref[fin].astype(np.float64)
    bad = np.abs(a - b) > 1e-4 * np.abs(b)
    assert not bad.any(), "%s: %d of %d entries off by more than 1e-4 relative (worst %g)" % (
        what, bad.sum(), bad.size, float(np.max(np.abs(a - b)[bad] / np.maximum(np.abs(b[bad]), 1e-300))))
