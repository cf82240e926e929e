#!/usr/bin/env python3
"""Freeze the descriptors of the golden frames as the ROUND-3 summation order produced them.

Until commit 265cd1b the oracle (and the kernel) added a descriptor sample's wrapped temporal vote straight into orientation
bin 0 of its partial histogram; since then a ninth slot collects those votes and is added to bin 0 after the partials have been
combined (DESIGN.md "fp spec" item 5). Both are orders the reference's atomicAdd may take (kernels/descriptor.cu:137). This script
rebuilds the oracle AS IT WAS in the parent of that commit (from this repository's own history, into a temporary directory) and
stores what it computes for the golden frames, so that any later co-change of oracle + kernel has a committed, reviewable bound:
tests/test_oracle_order_envelope.py compares today's oracle with this fixture at 1e-6 relative.

    python tests/golden/make_r3_order_fixture.py        (needs the git history; writes tests/golden/r3_order.npz)
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import helpers as H  # noqa: E402
from make_golden import CASES  # noqa: E402

COMMIT = "265cd1b^"           # last tree with the round-3 order
FLAGS = "-std=c++17 -O2 -fPIC -fopenmp -ffp-contract=off -fno-fast-math -mfma -mavx2 -fvisibility=hidden -shared".split()


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for f in ("nm_oracle.cpp", "nmo_math.h", "nmo_ransac.h", "nmo_warp.h"):
            src = subprocess.check_output(["git", "-C", ROOT, "show", "%s:oracle/%s" % (COMMIT, f)])
            open(os.path.join(tmp, f), "wb").write(src)
        so = os.path.join(tmp, "libnm_oracle_r3.so")
        subprocess.check_call(["g++"] + FLAGS + ["-o", so, os.path.join(tmp, "nm_oracle.cpp")])
        old = C.CDLL(so)
        fp = lambda a: a.ctypes.data_as(C.c_void_p)
        for name, (w, h, seeds, sigma) in CASES.items():
            for i, s in enumerate(seeds):
                gray = np.ascontiguousarray(H.blurred_frame(s, w, h, sigma=sigma), np.float32)
                cap = 2048
                desc = np.zeros((cap, 128), np.float32)
                xs, ys = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
                kp, ori = np.zeros((cap, 4), np.float32), np.zeros((cap, 2), np.float32)
                n = old.nmo_sift_detect_describe(fp(gray), C.c_int(w), C.c_int(h), C.c_int(cap), fp(desc), fp(xs), fp(ys),
                                                 fp(kp), fp(ori), None)
                out["%s_n%d" % (name, i)] = np.int32(n)
                out["%s_kpts%d" % (name, i)] = kp[:n].copy()
                out["%s_orient%d" % (name, i)] = ori[:n].copy()
                out["%s_desc%d" % (name, i)] = desc[:n].copy()
                print(name, i, "keypoints", n)
    out["commit"] = np.array(subprocess.check_output(["git", "-C", ROOT, "rev-parse", COMMIT]).decode().strip())
    np.savez_compressed(os.path.join(HERE, "r3_order.npz"), **out)


if __name__ == "__main__":
    main()
