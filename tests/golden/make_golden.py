#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the CPU oracle (KAT-13 of SURVEY.md 8(c)).

The reference cannot run here (CUDA-only) and has no fixtures of its own, so these vectors pin the ORACLE's output at
the time the Q-table semantics were fixed; tests/test_golden.py checks both the oracle and the HIP path against them.
Inputs are regenerated from seeds (niftymatch_amd.synth + the oracle's zero-padded Gaussian pre-blur), outputs stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import helpers as H  # noqa: E402
import oracle_lib as O  # noqa: E402

CASES = {"f128x96": (128, 96, (100, 101), 2.5), "f160x120": (160, 120, (200, 201), 3.0)}


def main():
    for name, (w, h, seeds, sigma) in CASES.items():
        out = {}
        descs = []
        for i, s in enumerate(seeds):
            r = O.sift_detect_describe(H.blurred_frame(s, w, h, sigma=sigma), 2048)
            out["n%d" % i] = np.int32(r["n"])
            out["counts%d" % i] = r["counts"]
            out["kpts%d" % i] = r["kpts"]
            out["orient%d" % i] = r["orient"]
            out["desc%d" % i] = r["desc"]
            descs.append(r["desc"])
        res, D, (m1, ix, m2) = O.sift_matches(descs[0], descs[1], 0.8)
        out.update(match=res, min1=m1, idx=ix, min2=m2, dist_row0=D[0])
        out.update(width=np.int32(w), height=np.int32(h), seeds=np.array(seeds, np.int32), sigma=np.float32(sigma))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "keypoints", int(out["n0"]), int(out["n1"]), "matches", int((res >= 0).sum()))


if __name__ == "__main__":
    main()
