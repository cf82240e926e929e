"""CPU-side checks of the drop-in boundary: libnm_hip.so loads without a GPU and exports every symbol that
include/nm_abi.h declares; host-only entry points behave like the reference's."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for hdr in ("nm_abi.h", "nm_client.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"(?:NM_API|visibility\(\"default\"\)\)\))\s+[\w\s\*]+?\b(\w+)\s*\(", text):
            names.append(m.group(1))
    return names


def test_library_exports_every_declared_symbol(nm):
    lib = nm.lib()
    declared = _declared()
    assert len(declared) >= 35 and "nm_sift_match_f32" in declared and "DivUp" in declared
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert set(nm.ABI_SYMBOLS) <= set(declared)


def test_batch_limit_is_the_same_in_the_header_and_the_binding(nm):
    """NM_SIFT_MAX_BATCH (include/nm_abi.h) = niftymatch_amd.SIFT_MAX_BATCH = the kernels' NM_MAX_BATCH: the per-frame pointer
    arrays of every frame-driver launch are sized by it (static_asserts keep them inside the 4 KB of kernel arguments)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = int(re.search(r"#define NM_SIFT_MAX_BATCH (\d+)", open(os.path.join(root, "include", "nm_abi.h")).read()).group(1))
    dev = int(re.search(r"#define NM_MAX_BATCH (\d+)", open(os.path.join(root, "niftymatch_amd", "csrc", "nm_common.hpp")).read()).group(1))
    assert hdr == dev == nm.SIFT_MAX_BATCH == 64


def test_integer_helpers(nm):
    lib = nm.lib()            # kernels/cudamath.cu:5-23
    assert [lib.DivUp(5, 2), lib.DivUp(4, 2), lib.DivDown(5, 2)] == [3, 2, 2]
    assert [lib.AlignUp(5, 4), lib.AlignUp(8, 4), lib.AlignDown(5, 4)] == [8, 8, 4]
    assert lib.nm_version().startswith(b"niftymatch_amd")


def test_host_taps_equal_oracle(nm, oracle):
    for s in (1.5198684, 1.2262735, 1.5450078, 1.9465879, 2.4525473, 3.0900159, 4.0, 0.3):
        t, r = nm.create_kernel_for_sigma(s)
        to, ro = oracle.create_kernel_for_sigma(s)
        assert r == ro and np.array_equal(t, to)


def test_cpp_headers_mirror_reference_names():
    """Every header a client of the hot path includes by name exists, flat, like ${prefix}/include/nm."""
    need = ["macros.h", "exception.h", "siftparams.h", "pyramidata.h", "siftdata.h", "siftfunctions.h", "convolution.h",
            "downsample.h", "cudamath.h", "keypoint.h", "orientation.h", "descriptor.h", "match.h", "transpose.h",
            "cudatimer.h", "cudautils.h", "bgra_2_gray.h", "cast.h", "ransac.h", "cudatex2D.h", "resample.h", "undistort.h"]
    have = os.listdir(os.path.join(ROOT, "niftymatch_amd", "nm"))
    assert not [h for h in need if h not in have]
    cfg = open(os.path.join(ROOT, "niftymatch_amd", "cmake", "NiftyMatchConfig.cmake")).read()
    for var in ("NiftyMatch_INCLUDE_DIR", "NiftyMatch_LIBS", "NiftyMatch_gpuutils_LIB", "NiftyMatch_kernels_LIB",
                "NiftyMatch_sift_LIB"):
        assert var in cfg


def test_no_cpu_fallback_when_library_missing(monkeypatch, nm):
    monkeypatch.setattr(nm, "_lib", None)
    monkeypatch.setattr(nm, "LIB_PATH", "/nonexistent/libnm_hip.so")
    try:
        nm.lib()
        raise AssertionError("expected NmError")
    except nm.NmError as e:
        assert "no CPU fallback" in str(e)


def test_cmake_find_package_dropin(tmp_path, nm):
    """A client CMakeLists written against the reference (FIND_PACKAGE(NiftyMatch CONFIG), NiftyMatch_LIBS,
    NiftyMatch_INCLUDE_DIR, flat includes) configures, builds and links against the installed drop-in."""
    import shutil
    import subprocess
    if shutil.which("cmake") is None or not os.path.exists(os.path.join(ROOT, "niftymatch_amd", "lib", "nm", "libsift.a")):
        import pytest
        pytest.skip("cmake or the static libraries are not available")
    prefix = tmp_path / "prefix"
    subprocess.check_call([os.path.join(ROOT, "tools", "install_prefix.sh"), str(prefix)], stdout=subprocess.DEVNULL)
    for f in ("macros.h", "siftfunctions.h", "NiftyMatchConfig.cmake"):
        assert (prefix / "include" / "nm" / f).exists()
    src = tmp_path / "client"
    src.mkdir()
    (src / "CMakeLists.txt").write_text(
        "CMAKE_MINIMUM_REQUIRED(VERSION 3.10)\nPROJECT(client CXX)\nSET(CMAKE_CXX_STANDARD 17)\n"
        "FIND_PACKAGE(NiftyMatch CONFIG REQUIRED)\nINCLUDE_DIRECTORIES(${NiftyMatch_INCLUDE_DIR} /opt/rocm/include)\n"
        "ADD_DEFINITIONS(-D__HIP_PLATFORM_AMD__)\nADD_EXECUTABLE(app main.cpp)\n"
        "TARGET_LINK_LIBRARIES(app ${NiftyMatch_LIBS})\n")
    (src / "main.cpp").write_text(
        '#include "siftfunctions.h"\n#include "convolution.h"\n#include "match.h"\n#include "cudamath.h"\n#include "macros.h"\n#include "resample.h"\n#include "undistort.h"\n#include <cstdio>\n'
        "int main() { SiftParams p(1920, 1080); std::printf(\"%d %zu %d\\n\", p._num_octaves, p._sigmas.size(), DivUp(7, 2));\n"
        "  if (p._num_octaves < 0) { PyramidData py(p); SiftData d(16); compute_dog(py, 8, 8); compute_sift_matches(&d, &d, nullptr);\n"
        "    CudaTex2D t((const uchar4 *)nullptr, 4, 4); resample_perspective_transform(nullptr, t, 4, 4, nullptr, nullptr, nullptr);\n"
        "    cuda_undistort(nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr); }\n"
        "  return 0; }\n")
    build = tmp_path / "build"
    build.mkdir()
    subprocess.check_call(["cmake", "-DNiftyMatch_DIR=%s" % (prefix / "include" / "nm"), str(src)], cwd=build,
                          stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", "."], cwd=build, stdout=subprocess.DEVNULL)
    out = subprocess.check_output([str(build / "app")]).decode().split()
    assert out == ["6", "5", "4"]


def test_match_plan_rejects_sets_beyond_the_32_bit_domain(nm):
    """2^22 rows or more would overflow the plan's 32-bit unit arithmetic (qblocks x T wraps at 2^24 x 2^24 and the host's
    group_owner then divides by zero): rejected with a status before any plan is made."""
    import ctypes as C
    lib = nm.lib()
    out = (C.c_int * 10)()
    buf = (C.c_int * 5)()
    for nA, nB in [(1 << 22, 10), (10, 1 << 22), (1 << 24, 1 << 24), (2 ** 31 - 1, 2 ** 31 - 1)]:
        assert lib.nm_sift_match_plan(nA, nB, out) != 0
        assert lib.nm_sift_match_plan_segments(nA, nB, 0, buf, 1) == -1
    assert lib.nm_sift_match_plan((1 << 22) - 1, (1 << 22) - 1, out) == 0 and out[0] == 1 << 14 and out[1] == 1 << 15
    assert lib.nm_sift_match_plan_segments((1 << 22) - 1, (1 << 22) - 1, 255, buf, 1) >= 1


def test_match_plan_invariants(nm):
    """Host logic of the matcher's work distribution (nm_sift_match_plan / _plan_segments, no GPU needed): every unit
    (query block, candidate tile) is processed by exactly one workgroup, workgroups of a group differ by at most one unit,
    the segments of a query block use the partial-list slots 0..n-1 exactly once with n <= S <= 64, exactly one segment
    per block is marked as finishing it and it is the block's last tile, the XCD-grouped order really puts the same
    candidate tiles on the workgroups of one group at the same step, and the workspace bound holds for every smaller
    shape served by the same buffer."""
    import ctypes as C
    lib = nm.lib()

    def plan(nA, nB):
        out = (C.c_int * 10)()
        assert lib.nm_sift_match_plan(nA, nB, out) == 0
        return list(out)

    def segments(nA, nB, wg):
        buf = (C.c_int * (5 * 256))()
        n = lib.nm_sift_match_plan_segments(nA, nB, wg, buf, 256)
        assert 0 <= n <= 256
        return [tuple(buf[5 * k: 5 * k + 5]) for k in range(n)]

    shapes = [(1, 1), (255, 127), (256, 128), (257, 129), (1000, 50), (50, 1000), (12223, 12080), (16384, 16384),
              (300, 100000), (100000, 300), (100000, 12500), (4097, 8193), (4096, 3000), (5000, 20000)]
    grouped = 0
    for nA, nB in shapes:
        qb, T, G, S, X, Gx, Tc, Cn, q_base, q_rem = plan(nA, nB)
        assert qb == -(-nA // 256) and T == -(-nB // 128)
        assert 1 <= G <= 256 and G == X * Gx and 1 <= S <= 64 and Cn == -(-T // Tc) and X * q_base + q_rem == qb
        grouped += X > 1
        seen = {}
        slots = {}
        ends = {}
        per_wg = []
        for wg in range(G):
            units = 0
            for (b, t0, n, slot, last) in segments(nA, nB, wg):
                assert 0 <= b < qb and 0 <= t0 and n >= 1 and t0 + n <= T and 0 <= slot < S
                for t in range(t0, t0 + n):
                    assert (b, t) not in seen
                    seen[(b, t)] = wg
                assert slot not in slots.setdefault(b, set())
                slots[b].add(slot)
                if last:
                    assert b not in ends and t0 + n == T
                    ends[b] = slot
                units += n
            per_wg.append(units)
        assert len(seen) == qb * T, (nA, nB)
        for b in range(qb):
            assert slots[b] == set(range(len(slots[b]))) and ends[b] == max(slots[b]), (nA, nB, b)
        for x in range(X):                               # balance inside every group
            grp = per_wg[x::X]
            assert max(grp) - min(grp) <= 1
        # partial lists (nA * S * 20 B) stay inside what the workspace bound reserves for them (nA * 64 * 20 B)
        assert lib.nm_sift_match_workspace_bytes(nA, nB) >= nA * S * 20 + 4 * (nA + nB) + 4 * nA
    assert grouped >= 4, "the XCD-grouped order was never chosen"
    # the point of the grouped order: at the bench's shape the workgroups of one XCD group that start a chunk together
    # read the same candidate tiles (6 query blocks share each tile range), not 32 different ranges
    qb, T, G, S, X, Gx, Tc, Cn, q_base, q_rem = plan(12223, 12080)
    assert (X, Gx, q_base) == (8, 32, 6)
    first_tiles = [segments(12223, 12080, 0 + X * v)[0][1] for v in range(Gx)]        # group 0
    assert len(set(t // Tc for t in first_tiles)) <= Cn and max(first_tiles.count(t) for t in set(first_tiles)) >= 1
    chunk_of = [t // Tc for t in first_tiles]
    assert max(chunk_of.count(c) for c in set(chunk_of)) >= 5        # >= 5 workgroups on the same chunk of tiles
    # a workspace sized for the largest shape serves every smaller one
    big = lib.nm_sift_match_workspace_bytes(16384, 16384)
    for nA, nB in [(12223, 12080), (16384, 1), (1, 16384), (5000, 16000)]:
        assert lib.nm_sift_match_workspace_bytes(nA, nB) <= big


def test_abi_headers_are_plain_c(tmp_path, nm):
    """include/*.h must be consumable by a C compiler (the boundary is a C ABI): compile and link a C99 client with
    -pedantic against libnm_hip.so and call a host-only entry point."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        import pytest
        pytest.skip("gcc not available")
    src = tmp_path / "client.c"
    src.write_text('#include "nm_abi.h"\n#include "nm_client.h"\n#include <stdio.h>\n'
                   "int main(void) { int p[10]; if (nm_sift_match_plan(12223, 12080, p)) return 1;\n"
                   '  printf("%d %d %d %d\\n", p[0], p[1], p[2], DivUp(7, 2)); return 0; }\n')
    exe = tmp_path / "client"
    libdir = os.path.join(ROOT, "niftymatch_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", libdir, "-lnm_hip", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert out == ["48", "95", "256", "4"]


def test_native_multi_gpu_entry_is_exported_and_sized(nm):
    """The native sharded-match entry (shard -> ncclAllGather -> merge) is part of the C ABI, resolves RCCL lazily (the
    library loads on a box without a GPU and without librccl in the process) and sizes its workspace monotonically."""
    lib = nm.lib()
    for name in ("nm_sift_match_allgather_f32", "nm_sift_match_allgather_workspace_bytes", "nm_sift_match_merge_packed_f32"):
        assert hasattr(lib, name)
    w1 = lib.nm_sift_match_allgather_workspace_bytes(100000, 12500, 1)
    w8 = lib.nm_sift_match_allgather_workspace_bytes(100000, 12500, 8)
    assert w8 - w1 >= 7 * 3 * 100000 * 4 - 1024 and w1 >= lib.nm_sift_match_workspace_bytes(100000, 12500) + 3 * 100000 * 4
    import subprocess
    out = subprocess.run(["readelf", "-d", nm.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in out, "libnm_hip.so must not depend on librccl at link time"


def test_native_multi_gpu_entry_builds_and_refuses_ranks_without_a_communicator(nm, tmp_path):
    """the first real N > 1 run must be self-checking. (i) examples/allpairs_rccl.cpp -- the native RCCL
    client of nm_sift_match_allgather_f32, one host thread per GPU -- compiles and links against libnm_hip.so + librccl here
    (no GPU needed); (ii) the entry refuses n_ranks > 1 without a communicator instead of merging one shard silently (a pure
    host-side check: no device is touched), and accepts the degenerate calls that must be no-ops."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "allpairs_rccl")
    libdir = os.path.dirname(nm.LIB_PATH)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "allpairs_rccl.cpp"), "-L", libdir, "-lnm_hip", "-lrccl", "-lpthread",
                        "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.getsize(exe) > 10000
    lib = nm.lib()
    dummy = C.c_void_p(4096)          # never dereferenced: the refusal happens before any device work
    assert lib.nm_sift_match_allgather_f32(dummy, 100, dummy, 50, 0, 2, dummy, 0.8, dummy, None, None) != 0
    assert lib.nm_sift_match_allgather_f32(dummy, 100, dummy, 50, 0, 0, dummy, 0.8, dummy, None, None) != 0
    assert lib.nm_sift_match_allgather_f32(None, 100, dummy, 50, 0, 1, dummy, 0.8, dummy, None, None) != 0
    assert lib.nm_sift_match_allgather_f32(dummy, 0, dummy, 50, 0, 8, dummy, 0.8, dummy, None, None) == 0     # no queries: no-op
    assert lib.nm_sift_match_allgather_workspace_bytes(100000, 12500, 8) > lib.nm_sift_match_allgather_workspace_bytes(100000, 12500, 1)
