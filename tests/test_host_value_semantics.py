"""Host-only part of the containers' value semantics (no GPU): default-constructed PyramidData / SiftData own no device or
pinned memory, so copying, assigning, moving and growing a std::vector of them must work anywhere and free nothing twice.
(The GPU half -- objects with live buffers -- is tests/test_gpu_api_fused.py::test_pyramiddata_and_siftdata_value_semantics.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "pyramidata.h"
#include "siftdata.h"
#include <cstdio>
#include <type_traits>
#include <utility>
#include <vector>
int main() {
    static_assert(std::is_copy_constructible<PyramidData>::value && std::is_copy_assignable<PyramidData>::value, "copyable");
    static_assert(std::is_nothrow_move_constructible<nm::pinned_counts>::value, "pinned buffer moves");
    PyramidData a, b(a), c;
    c = a; c = PyramidData(); a = std::move(b);
    std::vector<PyramidData> v(3);
    v.push_back(a); v.push_back(PyramidData()); v.resize(17);
    nm::pinned_counts p, q(p), r(std::move(q));
    p = r; r = std::move(p);
    SiftData s, t(s), u;
    u = t; u = u;
    std::vector<SiftData> w(2);
    w.push_back(s); w.resize(9);
    std::printf("%zu %zu %d %d\n", v.size(), w.size(), (int)r.allocated(), u._capacity);
    return 0;
}
'''


def test_default_constructed_containers_copy_move_and_grow(tmp_path, nm):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = tmp_path / "vs.cpp"
    src.write_text(SRC)
    exe = tmp_path / "vs"
    libdir = os.path.dirname(nm.LIB_PATH)
    r = subprocess.run([hipcc, "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "niftymatch_amd", "nm"),
                        str(src), "-L", os.path.join(libdir, "nm"), "-lsift", "-lkernels", "-lgpuutils", "-o", str(exe)],
                       capture_output=True, text=True)      # the drop-in static libraries, as NiftyMatch_LIBS names them
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["17", "9", "0", "0"]
