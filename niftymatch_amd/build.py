"""Build libnm_hip.so (HIP kernels + C ABI + the C++ API layer) for gfx950 with hipcc, in-tree.

    python -m niftymatch_amd.build [--force]

Objects go to niftymatch_amd/_build/, the library to niftymatch_amd/lib/libnm_hip.so (git-ignored; travels to the GPU
box with the snapshot). Also archives the drop-in static libraries lib/nm/lib{gpuutils,kernels,sift}.a.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
NMSRC = os.path.join(ROOT, "nm", "src")
BUILD = os.path.join(ROOT, "_build")
LIBDIR = os.path.join(ROOT, "lib")
LIB = os.path.join(LIBDIR, "libnm_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]

KERNEL_SOURCES = ["nm_pyramid.hip", "nm_keypoint.hip", "nm_describe.hip", "nm_match.hip", "nm_image.hip", "nm_warp.hip", "nm_ransac.hip", "nm_selftest.hip"]
SIFT_SOURCES = ["nm_frame.hip", "nm_tail.hip"]
KERNEL_CPP = ["kernels_api.cpp", "ransac.cpp"]
SIFT_CPP = ["pyramidata.cpp", "siftdata.cpp", "siftfunctions.cpp", "nm_client.cpp"]
UTIL_CPP = ["gpuutils.cpp"]


def _deps(src):
    d = [src]
    for folder in (CSRC, os.path.join(ROOT, "nm"), os.path.join(os.path.dirname(ROOT), "include")):
        for f in os.listdir(folder):
            if f.endswith((".h", ".hpp")):
                d.append(os.path.join(folder, f))
    d.append(os.path.abspath(__file__))
    return d


def _compile(src, force):
    obj = os.path.join(BUILD, os.path.basename(src) + ".o")
    if not force and os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(p) for p in _deps(src)):
        return obj
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=False):
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(os.path.join(LIBDIR, "nm"), exist_ok=True)
    groups = {
        "kernels": [os.path.join(CSRC, f) for f in KERNEL_SOURCES] + [os.path.join(NMSRC, f) for f in KERNEL_CPP],
        "sift": [os.path.join(CSRC, f) for f in SIFT_SOURCES] + [os.path.join(NMSRC, f) for f in SIFT_CPP],
        "gpuutils": [os.path.join(NMSRC, f) for f in UTIL_CPP],
    }
    all_src = [s for g in groups.values() for s in g]
    with ThreadPoolExecutor(max_workers=min(8, len(all_src))) as ex:
        objs = dict(zip(all_src, ex.map(lambda s: _compile(s, force), all_src)))
    newest = max(os.path.getmtime(o) for o in objs.values())
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + list(objs.values())
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        for name, srcs in groups.items():
            ar = os.path.join(LIBDIR, "nm", "lib%s.a" % name)
            if os.path.exists(ar):
                os.remove(ar)
            subprocess.check_call(["ar", "rcs", ar] + [objs[s] for s in srcs])
    # Extras: the native example clients and the diagnostic microbenchmarks. They are not part of the library, so a
    # failure here (e.g. a host without the RCCL development files) is reported and skipped, never raised.
    inc = os.path.join(ROOT, "..", "include")
    extras = []
    for name, libs in (("pairs_native", []), ("allpairs_rccl", ["-lrccl", "-lpthread"])):
        src = os.path.join(ROOT, "..", "examples", name + ".cpp")
        if os.path.exists(src):
            extras.append((src, os.path.join(LIBDIR, name), [HIPCC, "--offload-arch=" + ARCH, "-O2", "-std=c++17", "-I", inc, src,
                           "-L", LIBDIR, "-lnm_hip"] + libs + ["-Wl,-rpath,$ORIGIN", "-o", os.path.join(LIBDIR, name)], True))
    micro = os.path.join(ROOT, "..", "tools", "micro")
    if os.path.isdir(micro):
        for f in sorted(os.listdir(micro)):
            if f.endswith(".hip"):
                src, exe = os.path.join(micro, f), os.path.join(LIBDIR, f[:-4])
                extras.append((src, exe, [HIPCC, "--offload-arch=" + ARCH, "-O3", src, "-o", exe], False))

    def _extra(job):
        src, exe, cmd, needs_lib = job
        newest_in = max(os.path.getmtime(src), os.path.getmtime(LIB) if needs_lib else 0)
        if not force and os.path.exists(exe) and os.path.getmtime(exe) >= newest_in:
            return
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write("warning: optional build of %s failed (library unaffected):\n%s\n" % (os.path.basename(exe), r.stderr[-2000:]))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(_extra, extras))
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
