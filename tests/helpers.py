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
