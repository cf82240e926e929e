"""Row (a1) of SURVEY.md section 8 PINNED by the reference itself: tests/golden/siftparams_ref.json was produced by compiling
/root/reference/src/gpu/sift/siftparams.h (:30-51), unmodified, with g++ in the build container (oracle/Makefile targets
`ref` / `golden`, dumper oracle/siftparams_dump.cpp) -- the one part of the reference's hot path that builds without
nvcc. The oracle's restatement (nmo_sift_params) and the product's drop-in header (niftymatch_amd/nm/siftparams.h) must
reproduce every field bit for bit on every geometry of the fixture. The fixture travels to the GPU box; the reference
does not."""
import json
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "siftparams_ref.json")
DUMPER = os.path.join(ROOT, "oracle", "siftparams_dump.cpp")
REF_HEADER = "/root/reference/src/gpu/sift/siftparams.h"


def _bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def _golden():
    g = json.load(open(FIXTURE))
    assert len(g) >= 10 and g[0]["width"] == 1920 and g[0]["num_octaves"] == 6
    return g


def test_oracle_params_equal_the_reference_fixture(oracle):
    for ref in _golden():
        p = oracle.sift_params(ref["width"], ref["height"])
        got = {"width": p.width, "height": p.height, "num_octaves": p.num_octaves, "num_dog_levels": p.num_dog_levels,
               "level_max": p.level_max, "level_min": p.level_min, "sigma_d_0": _bits(p.sigma_d_0),
               "sigma_k": _bits(p.sigma_k), "sigma_0": _bits(p.sigma_0), "sigma_n": _bits(p.sigma_n),
               "base_smooth": _bits(p.base_smooth), "peak_threshold": _bits(p.peak_threshold),
               "edge_threshold": _bits(p.edge_threshold), "sigmas": [_bits(p.sigmas[i]) for i in range(p.num_sigmas)]}
        assert got == ref, (ref["width"], ref["height"])


def test_product_header_equals_the_reference_fixture(tmp_path):
    """The same dumper compiled against the PRODUCT's nm/siftparams.h (what libnm_hip.so's arena and the C++ API use)."""
    exe = str(tmp_path / "siftparams_nm_dump")
    hdr = os.path.join(ROOT, "niftymatch_amd", "nm", "siftparams.h")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-DSIFTPARAMS_HEADER=\"%s\"" % hdr, "-o", exe, DUMPER])
    got = json.loads(subprocess.check_output([exe]).decode())
    assert got == _golden()


def test_taps_follow_from_the_pinned_sigmas(oracle):
    """The Gaussian taps are a pure function of the pinned sigmas (pyramidata.cu:105-123): lengths {15,11,15,17,21,27} for
    (base, sigmas[0..4]) at every geometry (KAT-1), so everything upstream of the first convolution is reference-pinned."""
    ref = _golden()[0]
    sig = [struct.unpack("<f", struct.pack("<I", b))[0] for b in [ref["base_smooth"]] + ref["sigmas"]]
    assert [len(oracle.create_kernel_for_sigma(s)[0]) for s in sig] == [15, 11, 15, 17, 21, 27]


@pytest.mark.skipif(not os.path.exists(REF_HEADER), reason="the reference tree exists only in the build container")
def test_fixture_is_what_the_reference_header_gives_today(tmp_path):
    """Regenerates the dump from the reference header where it lies (no copy) and compares it with the committed fixture."""
    exe = str(tmp_path / "siftparams_ref_dump")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-ffp-contract=off", "-DSIFTPARAMS_HEADER=\"%s\"" % REF_HEADER, "-o", exe, DUMPER])
    assert json.loads(subprocess.check_output([exe]).decode()) == _golden()
