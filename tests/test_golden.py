"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
the oracle must keep reproducing them bit for bit (CPU), and the HIP path must hit them too (GPU)."""
import glob
import os

import numpy as np
import pytest

import helpers as H

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f*.npz")))


def _frames(g):
    w, h = int(g["width"]), int(g["height"])
    return [H.blurred_frame(int(s), w, h, sigma=float(g["sigma"])) for s in g["seeds"]]


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    descs = []
    for i, f in enumerate(_frames(g)):
        r = oracle.sift_detect_describe(f, 2048)
        assert r["n"] == int(g["n%d" % i]) and r["n"] > 20
        assert np.array_equal(r["counts"], g["counts%d" % i])
        for k in ("kpts", "orient", "desc"):
            assert np.array_equal(r[k], g["%s%d" % (k, i)]), k
        descs.append(r["desc"])
    res, D, (m1, ix, m2) = oracle.sift_matches(descs[0], descs[1], 0.8)
    assert np.array_equal(res, g["match"]) and np.array_equal(ix, g["idx"])
    assert np.array_equal(m1, g["min1"]) and np.array_equal(m2, g["min2"]) and np.array_equal(D[0], g["dist_row0"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_hip_path_hits_golden(nm, cuda, path):
    import torch
    g = np.load(path)
    w, h = int(g["width"]), int(g["height"])
    arenas = []
    for i, f in enumerate(_frames(g)):
        a = nm.SiftArena(w, h, 2048)
        a.detect_describe(torch.from_numpy(f).to(cuda))
        torch.cuda.synchronize()
        n = int(a.num_items.item())
        assert n == int(g["n%d" % i])
        assert np.array_equal(a.kpts[:n].cpu().numpy(), g["kpts%d" % i])
        assert np.array_equal(a.orients[:n].cpu().numpy(), g["orient%d" % i])
        assert np.array_equal(a.desc[:n].cpu().numpy(), g["desc%d" % i])
        arenas.append((a, n))
    (a0, n0), (a1, n1) = arenas
    res, D = nm.sift_match(a0.desc, a1.desc, 0.8, want_distance=True, nA=n0, nB=n1)
    torch.cuda.synchronize()
    assert np.array_equal(res.cpu().numpy(), g["match"])
    H.assert_distance(nm, D[0], g["dist_row0"], "distance row 0 (default mode: fp32 MFMA, 1e-4 relative)")
    before = nm.get_distance_mode()
    try:                                                  # the exact kernel: bit for bit
        nm.set_distance_mode("exact")
        res, D = nm.sift_match(a0.desc, a1.desc, 0.8, want_distance=True, nA=n0, nB=n1)
        torch.cuda.synchronize()
        assert np.array_equal(res.cpu().numpy(), g["match"]) and np.array_equal(D[0].cpu().numpy(), g["dist_row0"])
    finally:
        nm.set_distance_mode(before)
    m1, ix, m2 = nm.sift_match_shard(a0.desc[:n0].contiguous(), a1.desc[:n1].contiguous(), 0)
    assert np.array_equal(ix.cpu().numpy(), g["idx"]) and np.array_equal(m1.cpu().numpy(), g["min1"])
    assert np.array_equal(m2.cpu().numpy(), g["min2"])
