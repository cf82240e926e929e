# usage (GPU box): bash tools/pmc_coarse.sh  -- HBM bytes of match_coarse_kernel's 16-pair launch (FETCH_SIZE / WRITE_SIZE passes over tools/kcoarse16.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=r05_ao
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${T}_f --output-format csv -- python3 tools/kcoarse16.py > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${T}_w --output-format csv -- python3 tools/kcoarse16.py > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/${T}_f gpurun_out/${T}_w 14 > gpurun_out/${T}_pmc_hbm_traffic_match_coarse16.txt 2>&1; grep -i "match_\|prep_\|kernel " gpurun_out/${T}_pmc_hbm_traffic_match_coarse16.txt | cut -c1-150
rm -rf gpurun_out/${T}_f gpurun_out/${T}_w
