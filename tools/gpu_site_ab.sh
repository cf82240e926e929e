# usage (GPU box): bash tools/gpu_site_ab.sh <describe|orient|detect|pyramid> <variant> [<variant> ...]  -- one frame-driver kernel of a
# 64-frame call alone (tools/ksite.py): product against tools/_variants/libnm_hip_<variant>.so, alternating, five rounds (the clock
# state drifts by a few per cent between runs)
cd $GRAFT_REPO_ROOT
site=$1; shift
for i in 1 2 3 4 5; do
timeout -k 10 120 python tools/ksite.py $site 64 2>&1 | grep "^$site" | sed "s/^/product: /" || exit 1
for v in "$@"; do
NM_DIAGNOSTIC=1 NM_HIP_LIB=$PWD/tools/_variants/libnm_hip_$v.so timeout -k 10 120 python tools/ksite.py $site 64 2>&1 | grep "^$site" | sed "s/^/$v: /" || exit 1
done
done
