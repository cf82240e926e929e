# usage (GPU box): bash tools/gpu_desc_ab.sh <variant> [<variant> ...]  -- frame_desc_kernel of a 64-frame call alone: product against
# tools/_variants/libnm_hip_<variant>.so, alternating, five rounds (the clock state drifts by a few per cent between runs)
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
timeout -k 10 120 python tools/ksite.py describe 64 2>&1 | grep "^describe" | sed "s/^/product: /" || exit 1
for v in "$@"; do
NM_DIAGNOSTIC=1 NM_HIP_LIB=$PWD/tools/_variants/libnm_hip_$v.so timeout -k 10 120 python tools/ksite.py describe 64 2>&1 | grep "^describe" | sed "s/^/$v: /" || exit 1
done
done
