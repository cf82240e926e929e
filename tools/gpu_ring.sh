# usage (GPU box): bash tools/gpu_ring.sh  -- conv_pk_ring_kernel (NM_CONV_RING = workgroups per CU) against the one-tile-per-workgroup launches:
# bit-exactness through the GPU tests that drive batched calls, then the 64-frame chain alone (tools/kpyr_all.py), alternating
cd $GRAFT_REPO_ROOT
NM_CONV_RING=8 timeout -k 10 400 python -m pytest tests/test_gpu_frame.py tests/test_gpu_bench_config.py tests/test_gpu_stages.py -m gpu -x -q 2>&1 | tail -3 || exit 1
for i in 1 2; do
for r in 0 8 3 2; do
echo "ring $r:"
NM_CONV_RING=$r timeout -k 10 120 python tools/kpyr_all.py 64 10 nodog 2>&1 | grep "^B " || exit 1
NM_CONV_RING=$r timeout -k 10 120 python tools/kpyr_all.py 64 10 all 2>&1 | grep "^B " || exit 1
done
done
