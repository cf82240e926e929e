"""Synthetic inputs for tests and bench: counter-based uniform noise frames and descriptor sets.

Pure numpy; bit-reproducible on any host. The Gaussian pre-blur that turns a noise frame into a keypoint-rich
frame is NOT done here (bench.py blurs with the HIP convolve, tests may blur with the oracle; both are the same
zero-padded separable Gaussian and agree bit for bit).
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    """SplitMix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    """n floats in [0,1): top 24 bits of splitmix64(seed * 2^40 + index)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(1 << 40)
    bits = _splitmix64(base + idx) >> np.uint64(40)
    return (bits.astype(np.float32) * np.float32(1.0 / (1 << 24))).astype(np.float32)


def noise_frame(seed, width, height):
    """Uniform [0,255) fp32 frame, row-major (height, width)."""
    return (uniform01(seed, width * height) * np.float32(255.0)).reshape(height, width)


def _s64(v):
    """Python int (mod 2^64) as the signed value torch.int64 holds."""
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def uniform01_torch(seed, n, device):
    """uniform01 computed with torch int64 arithmetic on `device` (wrap-around multiply, logical shifts emulated):
    bit-identical to the numpy version (tests/test_abi.py::test_synth_torch_equals_numpy); bench.py uses it to make
    its frames on the GPU instead of shipping them over PCIe."""
    import torch

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)
    z = torch.arange(n, dtype=torch.int64, device=device) + _s64(int(seed) * (1 << 40))
    z = z + _s64(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * _s64(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    return lsr(z, 40).to(torch.float32) * (1.0 / (1 << 24))


def noise_frame_torch(seed, width, height, device):
    return (uniform01_torch(seed, width * height, device) * 255.0).reshape(height, width)


def descriptors(seed, n, dim=128):
    """n x dim fp32 Uniform[0,1) descriptor set (BASELINE configs 1 and 5)."""
    return uniform01(seed, n * dim).reshape(n, dim)


PREBLUR_SIGMA = {(640, 480): 3.0}


def preblur_sigma(width, height):
    """sigma of the zero-padded Gaussian pre-blur for a synthetic frame (3 at 640x480, else 4)."""
    return PREBLUR_SIGMA.get((width, height), 4.0)
