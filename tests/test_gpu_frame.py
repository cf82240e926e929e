"""GPU parity of the whole detect/describe path: frame driver (C ABI) and the C++ API client loop vs the oracle."""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from test_gpu_stages import _eq, _t

pytestmark = pytest.mark.gpu


def _run_arena(nm, cuda, frame, capacity):
    import torch
    h, w = frame.shape
    arena = nm.SiftArena(w, h, capacity)
    arena.detect_describe(_t(frame, cuda))
    torch.cuda.synchronize()
    n = int(arena.num_items.item())
    out = dict(n=n, desc=arena.desc[:n].cpu().numpy(), x=arena.x[:n].cpu().numpy(), y=arena.y[:n].cpu().numpy(),
               kpts=arena.kpts[:n].cpu().numpy(), orient=arena.orients[:n].cpu().numpy())
    arena.close()
    return out


@pytest.mark.parametrize("wh,cap", [((128, 96), 2048), ((250, 187), 4096), ((37, 41), 256), ((640, 480), 16384),
                                    ((1920, 1080), 16384)])
def test_frame_driver_matches_oracle(nm, oracle, cuda, wh, cap):
    w, h = wh
    frame = H.blurred_frame(0, w, h)
    ref = oracle.sift_detect_describe(frame, cap)
    got = _run_arena(nm, cuda, frame, cap)
    assert got["n"] == ref["n"]
    _eq(got["kpts"], ref["kpts"], "keypoints (x,y,sigma,level), output order")
    _eq(got["orient"], ref["orient"], "orientations")
    _eq(got["x"], ref["x"], "x")
    _eq(got["y"], ref["y"], "y")
    _eq(got["desc"], ref["desc"], "descriptors")


def test_frame_driver_capacity_clipping(nm, oracle, cuda):
    frame = H.blurred_frame(3, 320, 240)
    full = oracle.sift_detect_describe(frame, 16384)
    cap = full["n"] // 2 + 3
    ref = oracle.sift_detect_describe(frame, cap)
    got = _run_arena(nm, cuda, frame, cap)
    assert ref["n"] == cap == got["n"]
    _eq(got["desc"], ref["desc"], "descriptors under capacity (Q13)")
    _eq(got["desc"], full["desc"][:cap], "prefix property")


def test_frame_driver_reuse_is_stateless(nm, oracle, cuda):
    import torch
    a = H.blurred_frame(11, 256, 192)
    b = H.blurred_frame(12, 256, 192)
    arena = nm.SiftArena(256, 192, 4096)
    outs = []
    for f in (a, b, a):
        arena.detect_describe(_t(f, cuda))
        torch.cuda.synchronize()
        n = int(arena.num_items.item())
        outs.append((n, arena.desc[:n].cpu().numpy().copy()))
    arena.close()
    assert outs[0][0] == outs[2][0] and np.array_equal(outs[0][1], outs[2][1])
    ref = oracle.sift_detect_describe(b, 4096)
    assert outs[1][0] == ref["n"]
    _eq(outs[1][1], ref["desc"], "second frame after reuse")


def test_cpp_api_client_loop(nm, oracle, cuda):
    """The reference-style client (SiftParams/PyramidData/SiftData + compute_*) gives the same answer."""
    frame = H.blurred_frame(0, 640, 480)
    cap = 4096
    ref = oracle.sift_detect_describe(frame, cap)
    desc = np.zeros((cap, 128), np.float32)
    x = np.zeros(cap, np.float32)
    y = np.zeros(cap, np.float32)
    n = nm.lib().nm_client_detect_describe(frame.ctypes.data, 640, 480, cap, desc.ctypes.data, x.ctypes.data,
                                           y.ctypes.data)
    assert n == ref["n"]
    _eq(desc[:n], ref["desc"], "C++ API descriptors")
    _eq(x[:n], ref["x"], "C++ API x")
    _eq(y[:n], ref["y"], "C++ API y")
