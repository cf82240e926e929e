"""Wider one-pass sweeps for the kernels whose fast paths were rewritten late in round 1 (packed Gaussian/gradient,
batched frame driver, key-packed persistent matcher): more seeds, odd geometries, structured descriptor sets with
duplicates, near-duplicates, zeros and large magnitudes. Everything bit-exact against the oracle."""
import numpy as np
import pytest

import helpers as H
from test_gpu_stages import _eq, _t

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wh,sigma,seeds", [((320, 240), 1.2, (101, 102, 103)), ((252, 188), 2.5, (104, 105)),
                                            ((640, 360), 3.0, (106, 107, 108, 109))])
def test_soak_frames_batched(nm, oracle, cuda, wh, sigma, seeds):
    import torch
    w, h = wh
    frames = [H.blurred_frame(s, w, h, sigma=sigma) for s in seeds]
    frames[0] = frames[0] * np.float32(1e-3)                     # dim frame: small gradients everywhere
    frames[-1] = frames[-1] * np.float32(37.0) + np.float32(5)   # bright frame
    arenas = [nm.SiftArena(w, h, 8192) for _ in seeds]
    nm.detect_describe_batch(arenas, [_t(f, cuda) for f in frames])
    torch.cuda.synchronize()
    for a, f in zip(arenas, frames):
        ref = oracle.sift_detect_describe(f, 8192)
        n = int(a.num_items.item())
        assert n == ref["n"]
        _eq(a.kpts[:n], ref["kpts"], "soak keypoints")
        _eq(a.orients[:n], ref["orient"], "soak orientations")
        _eq(a.desc[:n], ref["desc"], "soak descriptors")
        a.close()


def _structured(rng, n, kind):
    d = rng.uniform(0, 1, (n, 128)).astype(np.float32)
    if kind == "sift":                       # sparse, clipped, normalised-looking rows
        d = np.where(d > 0.6, d, 0).astype(np.float32)
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6)
        d = np.minimum(d, 0.2).astype(np.float32) * np.float32(512)
    elif kind == "big":                      # large, but squared distances stay below the scan's initial 2139095040.0f
        d = d * np.float32(2e3)
    elif kind == "tiny":
        d = d * np.float32(1e-4)
    return np.ascontiguousarray(d, dtype=np.float32)


@pytest.mark.parametrize("kind", ["sift", "big", "tiny"])
def test_soak_matcher_structured_sets(nm, oracle, cuda, kind):
    rng = np.random.default_rng({"sift": 1, "big": 2, "tiny": 3}[kind])
    for na, nb in [(700, 1900), (1300, 257), (2049, 3071)]:
        A, B = _structured(rng, na, kind), _structured(rng, nb, kind)
        # exact duplicates of queries among the candidates (distance 0, ties on the lowest index), near-duplicates one
        # ulp-scale apart, duplicated candidates (equal best and second best -> ratio 1), all-zero rows
        B[5] = A[3]; B[nb - 1] = A[3]
        B[17] = A[11]; B[18] = A[11] * np.float32(1 + 2 ** -20)
        B[40] = B[41]
        A[20] = 0; B[60] = 0; B[61] = 0
        m1, ix, m2 = oracle.sift_match_shard(A, B, 0)
        import torch
        prior = torch.full((na,), -7, dtype=torch.int32, device=cuda)     # rows with min2 <= 0 must stay untouched
        got, _ = nm.sift_match(_t(A, cuda), _t(B, cuda), 0.8, prior=prior)
        ref_p, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False, prior=np.full(na, -7, np.int32))
        assert np.array_equal(got.cpu().numpy(), ref_p), (kind, na, nb)
        t = nm.sift_match_shard(_t(A, cuda), _t(B, cuda), 0)
        assert np.array_equal(t[1].cpu().numpy(), ix), (kind, na, nb)
        assert np.array_equal(t[0].cpu().numpy(), m1) and np.array_equal(t[2].cpu().numpy(), m2), (kind, na, nb)
