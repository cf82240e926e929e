# usage (GPU box): bash tools/gpu_pair_latency_sweep.sh  -- batch-1 pair latency under grid-size switches, two alternations
cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout -k 10 120 python tools/kpair_latency.py || exit 1
for v in 12288 16384 24576 32768 49152; do NM_DESC_BLOCKS=$v timeout -k 10 120 python tools/kpair_latency.py || exit 1; done
for v in 768 1536 2048 3072; do NM_ORIENT_BLOCKS=$v timeout -k 10 120 python tools/kpair_latency.py || exit 1; done
done
