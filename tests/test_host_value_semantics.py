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


# Source compatibility of `nm::lazy_int SiftData::_num_items` with client code written for the reference's `int _num_items`
# (sift/siftdata.h:66). COMPILES: every way the reference itself uses the member (siftdata.cu:23,45,56;
# siftfunctions.cu:18-19,167-178) and the common client idioms. MUST NOT COMPILE (documented in INTEGRATION.md with their
# one-line fixes): the four things a class type cannot do.
COMPAT_OK = r'''
#include "siftdata.h"
#include <algorithm>
#include <cstdio>
#include <vector>
static int takes_int(int v) { return v; }
int main() {
    SiftData data, other;
    const SiftData *A = &data, *B = &other;
    const int A_size = A->_num_items;                       // siftfunctions.cu:18
    const int B_size = B->_num_items;
    int capacity = 16, num_pts = 5;
    if (num_pts + data._num_items > capacity) num_pts = capacity - data._num_items;      // siftfunctions.cu:167-168
    std::vector<float> desc(128 * 64);
    float *d = &desc[data._num_items * 128];                // :172
    data._num_items += num_pts;                             // :178
    other._num_items = data._num_items;                     // siftdata.cu:23
    data._num_items = data._capacity = 0;                   // siftdata.cu:56
    data._num_items = 0;                                    // siftdata.cu:45
    // client idioms
    const bool cond = A_size > B_size;
    int n1 = cond ? data._num_items : 0;                    // was ambiguous with an implicit int constructor
    int n2 = cond ? 7 : other._num_items;
    bool e1 = data._num_items == 0, e2 = 0 == data._num_items, e3 = data._num_items < other._num_items, e4 = data._num_items != other._num_items;
    int n3 = takes_int(data._num_items) + data._num_items * 2 - other._num_items / 1;
    long n4 = data._num_items;  size_t n5 = (size_t)data._num_items;  double n6 = data._num_items;
    for (int i = 0; i < data._num_items; ++i) n3 += i;
    ++data._num_items; data._num_items++; --data._num_items; data._num_items--; data._num_items -= 1;
    int n7 = std::min<int>(data._num_items, capacity), n8 = std::max(int(data._num_items), capacity);
    std::vector<int> v(data._num_items + 1);
    switch (data._num_items) { case 0: break; default: break; }
    if (!data._num_items) n3 += 1;
    auto lam = [&]() -> int { return data._num_items; };
    std::printf("%d %d %d\n", int(data._num_items), (int)other._num_items, n1 + n2 + n3 + (int)n4 + (int)n5 + (int)n6 + n7 + n8 + (e1 + e2 + e3 + e4) + lam() + (int)v.size() + (int)(d - &desc[0]));
    return 0;
}
'''
COMPAT_BAD = {
    "int_reference": "int &r = data._num_items; (void)r;",
    "int_pointer": "int *p = &data._num_items; (void)p;",
    "std_min_deduction": "int m = std::min(data._num_items, 4); (void)m;",
    "implicit_from_int": "nm::lazy_int z = 3; (void)z;",
}


def test_lazy_int_source_compatibility_compile_only(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    inc = os.path.join(ROOT, "niftymatch_amd", "nm")
    base = [hipcc, "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror=non-pod-varargs", "-I", inc]
    ok = tmp_path / "ok.cpp"
    ok.write_text(COMPAT_OK)
    r = subprocess.run(base + [str(ok)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    for name, line in COMPAT_BAD.items():
        src = tmp_path / (name + ".cpp")
        src.write_text('#include "siftdata.h"\n#include <algorithm>\nint main() { SiftData data; %s return 0; }\n' % line)
        r = subprocess.run(base + [str(src)], capture_output=True, text=True)
        assert r.returncode != 0, "%s compiles now: update INTEGRATION.md's list" % name
    va = tmp_path / "varargs.cpp"                   # a class through C varargs: an error under clang (-Wnon-pod-varargs)
    va.write_text('#include "siftdata.h"\n#include <cstdio>\nint main() { SiftData data; std::printf("%d", data._num_items); return 0; }\n')
    assert subprocess.run(base + [str(va)], capture_output=True, text=True).returncode != 0
