"""Scratch builds for A/B timing: recompile ONE kernel source with extra preprocessor flags and link it with the objects of
the regular build into tools/_variants/libnm_hip_<name>.so (git-ignored; selected at run time with NM_HIP_LIB=<path> NM_DIAGNOSTIC=1).
    python tools/build_variant.py <name> <source.hip>[,<source2.hip>] -DFLAG=1 ...
Results of such variants may be wrong by design (pieces compiled out); only their times are read."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from niftymatch_amd import build as B
name, srcs = sys.argv[1], sys.argv[2].split(",")
flags = sys.argv[3:]
B.build()
out_dir = os.path.join(root, "tools", "_variants")        # NOT under the product's lib/: a diagnostic build must never sit beside libnm_hip.so
os.makedirs(out_dir, exist_ok=True)
vobjs = []
for src in srcs:
    obj = os.path.join(B.BUILD, "variant_%s_%s.o" % (name, os.path.basename(src)))
    subprocess.check_call([B.HIPCC] + B.FLAGS + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj])
    vobjs.append(obj)
objs = [os.path.join(B.BUILD, f) for f in os.listdir(B.BUILD)
        if f.endswith(".o") and not f.startswith("variant_") and f not in [os.path.basename(x) + ".o" for x in srcs]]
lib = os.path.join(out_dir, "libnm_hip_%s.so" % name)
subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + vobjs + objs)
for obj in vobjs:                  # the variant's objects are only link inputs: ~0.5 MB each, a hundred of them once rode every gpurun push
    os.remove(obj)
print(lib)
