"""The boundary's error convention ("Errors") PINNED by the reference itself:
tests/golden/exception_ref.json was produced by compiling /root/reference/src/gpu/utils/exception.h (:25-110), unmodified,
with g++ in the build container (oracle/Makefile targets `ref` / `golden`, dumper oracle/exception_dump.cpp). The product's
drop-in header niftymatch_amd/nm/exception.h must give a client's catch block the same what() text, character for
character, through the same std exception types. The fixture travels to the GPU box; the reference does not."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, "tests", "golden", "exception_ref.json")
DUMPER = os.path.join(ROOT, "oracle", "exception_dump.cpp")
REF_HEADER = "/root/reference/src/gpu/utils/exception.h"
NM_HEADER = os.path.join(ROOT, "niftymatch_amd", "nm", "exception.h")


def _golden():
    g = json.load(open(FIXTURE))
    assert [r["macro"] for r in g][:3] == ["RUNTIME_EXCEPTION", "LOGIC_EXCEPTION", "RANGE_EXCEPTION"]
    assert all(r["thrown"] and r["caught_as_its_std_type"] for r in g)
    return g


def _dump(tmp_path, header, std, name):
    exe = str(tmp_path / name)
    # nm/exception.h declares nm_error_string (the C ABI's) for nm_check; the dumper never calls it
    subprocess.check_call(["g++", "-std=" + std, "-O2", "-DEXCEPTION_HEADER=\"%s\"" % header, "-o", exe, DUMPER])
    return json.loads(subprocess.check_output([exe]).decode())


@pytest.mark.parametrize("std", ["c++11", "c++17"])
def test_product_header_equals_the_reference_fixture(tmp_path, std):
    assert _dump(tmp_path, NM_HEADER, std, "exception_nm_dump") == _golden()


def test_messages_have_the_documented_shape():
    """The text the reference's helper prints (exception.h:96-109): file, line, then the detailed description; the default
    description is "-"."""
    g = {r["macro"]: r["what"] for r in _golden()}
    assert g["RUNTIME_EXCEPTION"] == ("Exception in file 'client.cpp' in line 100\n"
                                      "Detailed description: Pyramid depth must be positive\n")
    assert g["throw_it default"].endswith("Detailed description: -\n")


@pytest.mark.skipif(not os.path.exists(REF_HEADER), reason="the reference tree exists only in the build container")
def test_fixture_is_what_the_reference_header_gives_today(tmp_path):
    """Regenerates the dump from the reference header where it lies (no copy) and compares it with the committed fixture."""
    assert _dump(tmp_path, REF_HEADER, "c++11", "exception_ref_dump") == _golden()
