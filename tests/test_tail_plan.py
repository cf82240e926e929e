"""Host logic of the octave-tail launch (niftymatch_amd/csrc/nm_tail.hip; what it replaces: the per-octave launch sequence of
the reference's client loop, sift/siftfunctions.cu:42-181): its work items run in ONE order, and the launch is deadlock-free
for any number of resident workgroups only if every item depends on items that come EARLIER in that order. nm_sift_tail_plan
is a host function, so this needs no GPU."""
import ctypes as C

import pytest

A, B, DET, SCAN, GRAD = 0, 1, 2, 3, 4


def plan(nm, w, h, T=2):
    seg = (C.c_int * (8 * 40))()
    info = (C.c_int * 4)()
    n = nm.lib().nm_sift_tail_plan(w, h, T, seg, 40, info)
    rows = [tuple(seg[8 * i: 8 * i + 8]) for i in range(n)]
    return rows, list(info)


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160), (640, 480), (1916, 1076), (400, 300), (256, 192), (1280, 720),
                                 (4096, 130), (129, 2000)])
@pytest.mark.parametrize("T", [1, 2, 3])
def test_segments_are_in_topological_order(nm, w, h, T):
    rows, info = plan(nm, w, h, T)
    if not rows:
        return                                      # not covered: too few octaves for this T
    items, lds, scan_lds, n_oct = info
    assert 1 <= n_oct <= 8 and lds <= 144 * 1024 and scan_lds <= 60 * 1024
    pos = {}
    first = 0
    for i, (kind, slot, per, fst, o, whole, ow, oh) in enumerate(rows):
        assert fst == first and per >= 1 and o == T + slot and (ow, oh) == (w >> o, h >> o)
        first += per
        pos[(kind, slot)] = i
    assert first == items
    for (kind, slot), i in pos.items():
        whole = rows[i][5]
        deps = []
        if kind == A and slot > 0:
            deps.append((A, slot - 1))              # level 0 = the decimated level 3 of the octave before
        if kind in (B, GRAD):
            deps.append((A, slot))                  # level 3 (tiles) / levels 1..3 (whole plane)
        if kind == DET:
            deps.append((A, slot) if whole else (B, slot))
        for d in deps:
            assert d in pos and pos[d] < i, (w, h, T, (kind, slot), d)
    for slot in range(n_oct):                       # every tail octave has its levels and its detection; tiles have a B segment
        assert (A, slot) in pos and (DET, slot) in pos
        whole = rows[pos[(A, slot)]][5]
        assert ((B, slot) in pos) == (not whole) and ((GRAD, slot) in pos) == bool(whole)
        kind, _, per, _, o, _, ow, oh = rows[pos[(A, slot)]]
        assert per == (1 if whole else -(-ow // 64) * -(-oh // 32))
    assert SCAN not in [r[0] for r in rows]         # the scans are a launch of their own (nm_launch_tail_scan)


def test_geometries_outside_the_plan_fall_back(nm):
    assert plan(nm, 128, 96)[0] == []               # two octaves: nothing behind octave 1
    assert plan(nm, 64, 64)[0] == []
    assert plan(nm, 1920, 1080, T=0)[0] == []
    rows, info = plan(nm, 1920, 1080)
    assert [r[4] for r in rows if r[0] == A] == [2, 3, 4, 5] and info[3] == 4
    assert [r[5] for r in rows if r[0] == A] == [0, 0, 0, 1]      # 60 x 33 is the one plane that is a single item
