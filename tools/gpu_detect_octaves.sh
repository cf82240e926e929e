# usage (GPU box): bash tools/gpu_detect_octaves.sh [<variant>]  -- per-octave durations of detect_stage_kernel in 64-frame calls (kernel trace of
# tools/ksite.py detect 64), product or tools/_variants/libnm_hip_<variant>.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$1" ]; then export NM_DIAGNOSTIC=1 NM_HIP_LIB=$PWD/tools/_variants/libnm_hip_$1.so; fi
rm -rf gpurun_out/_dto
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/_dto -- python3 tools/ksite.py detect 64 > /dev/null 2>&1 || exit 1
python3 tools/prof_summary.py gpurun_out/_dto 200 | grep "detect_stage_kernel\|tail_kernel" | cut -c1-130
rm -rf gpurun_out/_dto
