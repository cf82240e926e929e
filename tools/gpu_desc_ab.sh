# usage (GPU box): bash tools/gpu_desc_ab.sh <variant>  -- frame_desc_kernel of a 64-frame call alone: product against tools/_variants/libnm_hip_<variant>.so, alternating
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout -k 10 120 python tools/ksite.py describe 64 2>&1 | grep "^describe" | sed "s/^/product: /" || exit 1
NM_DIAGNOSTIC=1 NM_HIP_LIB=$PWD/tools/_variants/libnm_hip_$1.so timeout -k 10 120 python tools/ksite.py describe 64 2>&1 | grep "^describe" | sed "s/^/$1: /" || exit 1
done
