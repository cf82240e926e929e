# usage (GPU box): bash tools/pmc_detect.sh [variant]  -- HBM bytes of the detection launches of a 64-frame call (FETCH_SIZE / WRITE_SIZE passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -n "$1" ]; then export NM_DIAGNOSTIC=1 NM_HIP_LIB=$GRAFT_REPO_ROOT/tools/_variants/libnm_hip_$1.so; fi
T=r05_an_${1:-product}
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${T}_f --output-format csv -- python3 tools/ksite.py detect 64 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${T}_w --output-format csv -- python3 tools/ksite.py detect 64 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/${T}_f gpurun_out/${T}_w 14 > gpurun_out/${T}_detect_pmc_traffic.txt 2>&1; grep -i "detect\|kernel " gpurun_out/${T}_detect_pmc_traffic.txt | cut -c1-150
rm -rf gpurun_out/${T}_f gpurun_out/${T}_w
