# usage (GPU box): bash tools/gpu_pairxcd.sh  -- coarse pass with one XCD per pair (default) against every workgroup on every pair (NM_COARSE_PAIR_XCD=0):
# the matcher's GPU tests, the 16-pair launch alone (tools/kcoarse16.py), the headline, and the launch's HBM traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_match.py tests/test_gpu_bench_config.py tests/test_gpu_streams_graphs.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for i in 1 2 3; do
timeout -k 10 120 python tools/kcoarse16.py 2>&1 | grep "coarse launch" | sed "s/^/pair-per-xcd: /" || exit 1
NM_COARSE_PAIR_XCD=0 timeout -k 10 120 python tools/kcoarse16.py 2>&1 | grep "coarse launch" | sed "s/^/all-on-all:   /" || exit 1
done
bash tools/ab_env.sh "NM_COARSE_PAIR_XCD=0" || exit 1
timeout -k 10 300 python3 tools/pmc_collect.py r06_l match_coarse_kernel 2>&1 | tail -2
