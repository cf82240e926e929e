#!/usr/bin/env python3
"""Octave-tail launch (csrc/nm_tail.hip) against the per-octave launches: bit-for-bit comparison of every output of the
frame driver on a few geometries, and single-frame / batched timings of both (NM_FRAME_TAIL is read when an arena is made)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import niftymatch_amd as nm
import helpers as H

dev = torch.device("cuda:0")
CAP = 16384


def arena(w, h, tail):
    if tail is None:
        os.environ.pop("NM_FRAME_TAIL", None)
    else:
        os.environ["NM_FRAME_TAIL"] = str(tail)
    a = nm.SiftArena(w, h, CAP, device=dev)
    os.environ.pop("NM_FRAME_TAIL", None)
    return a


def outputs(a):
    n = int(a.num_items.item())
    return n, [t[:n].cpu().numpy().copy() for t in (a.kpts, a.orients, a.x, a.y, a.desc)]


def compare(w, h, seeds, tail=None):
    bad = 0
    a0, a1 = arena(w, h, 0), arena(w, h, tail)
    for s in seeds:
        f = torch.from_numpy(H.blurred_frame(s, w, h)).to(dev)
        a0.detect_describe(f); a1.detect_describe(f)
        torch.cuda.synchronize()
        n0, o0 = outputs(a0); n1, o1 = outputs(a1)
        ok = n0 == n1 and all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(o0, o1))
        print("%4dx%-4d seed %3d tail %s: %5d / %5d keypoints %s" % (w, h, s, tail, n0, n1, "identical" if ok else "DIFFERENT"), flush=True)
        bad += 0 if ok else 1
    a0.close(); a1.close()
    return bad


def timing(w, h, tail, B, reps=40):
    ars = [arena(w, h, tail) for _ in range(B)]
    fr = [torch.from_numpy(H.blurred_frame(s, w, h)).to(dev) for s in range(B)]
    for _ in range(5):
        nm.detect_describe_batch(ars, fr) if B > 1 else ars[0].detect_describe(fr[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        nm.detect_describe_batch(ars, fr) if B > 1 else ars[0].detect_describe(fr[0])
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    for a in ars:
        a.close()
    return us


if __name__ == "__main__":
    bad = 0
    for (w, h) in ((256, 192), (640, 480), (1920, 1080), (400, 300), (1916, 1076)):
        bad += compare(w, h, (0, 1))
    for T in (1, 3):
        bad += compare(1920, 1080, (2,), tail=T)
    print("mismatching frames:", bad, flush=True)
    for B in (1, 16):
        for tail in (0, 2, 1):
            print("1080p, %2d frame(s) per call, NM_FRAME_TAIL=%d: %.1f us per call = %.1f us per frame" %
                  (B, tail, timing(1920, 1080, tail, B), timing(1920, 1080, tail, B) / B), flush=True)
