# usage (on the GPU box): bash tools/prof_headline.sh <tag>  -- headline-only bench under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/${T}_prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-allpairs --no-detect256 --no-dropin --no-latency --no-f32-loop > gpurun_out/${T}_bench_headline_only_profiled.json 2> gpurun_out/${T}_prof.err
python tools/prof_summary.py gpurun_out/${T}_prof 90 > gpurun_out/${T}_bench_headline_only_kernel_summary.txt 2>&1; head -16 gpurun_out/${T}_bench_headline_only_kernel_summary.txt | cut -c1-130
rm -rf gpurun_out/${T}_prof
