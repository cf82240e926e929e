# usage (GPU box): bash tools/gpu_coarse_ab.sh <variant> [<variant> ...]  -- match_coarse_kernel of a 16-pair call (tools/kcoarse16.py):
# product against tools/_variants/libnm_hip_<variant>.so, alternating, three rounds
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout -k 10 120 python tools/kcoarse16.py 2>&1 | grep "coarse launch" || exit 1
for v in "$@"; do
NM_DIAGNOSTIC=1 NM_HIP_LIB=$PWD/tools/_variants/libnm_hip_$v.so timeout -k 10 120 python tools/kcoarse16.py 2>&1 | grep "coarse launch" || exit 1
done
done
