# usage (GPU box): bash tools/pmc_gather_traffic.sh <tag>  -- HBM-side bytes the orientation / descriptor launches of a 64-frame call fetch (FETCH_SIZE, TCC hits / misses)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE -d gpurun_out/${T}_f --output-format csv -- python3 tools/ksite.py describe 64 > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d gpurun_out/${T}_h --output-format csv -- python3 tools/ksite.py describe 64 > /dev/null 2>&1
python tools/pmc_counters.py gpurun_out/${T}_f frame_ > gpurun_out/${T}_gather_traffic.txt 2>&1
python tools/pmc_counters.py gpurun_out/${T}_h frame_ >> gpurun_out/${T}_gather_traffic.txt 2>&1
rm -rf gpurun_out/${T}_f gpurun_out/${T}_h
cat gpurun_out/${T}_gather_traffic.txt
