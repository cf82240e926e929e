# usage (GPU box): bash tools/gpu_chain_kernels.sh  -- per-launch durations of the frame driver's scale-space chain alone (64 frames, kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_ck
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/_ck -- python3 tools/kpyr_all.py 64 12 nodog > /dev/null 2>&1 || exit 1
python3 tools/prof_summary.py gpurun_out/_ck 200 | grep -v "at::\|Functor" | cut -c1-130
rm -rf gpurun_out/_ck
