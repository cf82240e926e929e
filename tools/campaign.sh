cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_n_bench.json 2> gpurun_out/r05_n_bench.err; python tools/show_bench.py gpurun_out/r05_n_bench.json | cut -c1-300
echo "--- profiled headline-only run"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_n_prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-allpairs --no-detect256 --no-dropin --no-latency --no-f32-loop > gpurun_out/r05_n_bench_headline_only_profiled.json 2> gpurun_out/r05_n_prof.err
python tools/prof_summary.py gpurun_out/r05_n_prof 90 > gpurun_out/r05_n_bench_headline_only_kernel_summary.txt 2>&1; head -30 gpurun_out/r05_n_bench_headline_only_kernel_summary.txt | cut -c1-130
find gpurun_out/r05_n_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_n_bench_headline_only_kernel_stats.csv; rm -rf gpurun_out/r05_n_prof
echo "--- PMC: distance kernel"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r05_n_pmc_f --output-format csv -- python3 tools/kdist.py > /dev/null 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r05_n_pmc_w --output-format csv -- python3 tools/kdist.py > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/r05_n_pmc_f gpurun_out/r05_n_pmc_w 12 > gpurun_out/r05_n_distance_pmc_traffic.txt 2>&1; grep -i "distance\|kernel " gpurun_out/r05_n_distance_pmc_traffic.txt | cut -c1-140
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA -d gpurun_out/r05_n_pmc_m --output-format csv -- python3 tools/kdist.py > /dev/null 2>&1
python tools/pmc_counters.py gpurun_out/r05_n_pmc_m distance_mfma > gpurun_out/r05_n_distance_pmc_counters.txt 2>&1; cat gpurun_out/r05_n_distance_pmc_counters.txt
rm -rf gpurun_out/r05_n_pmc_f gpurun_out/r05_n_pmc_w gpurun_out/r05_n_pmc_m
echo "--- mfma f32 selftest"
python - <<'PY' > gpurun_out/r05_n_mfma_f32_model.txt 2>&1
import niftymatch_amd as nm
print(nm.selftest_mfma_f32(1 << 24, 8192), "distance budget", nm.match_distance_budget(), "f32 screen budget", nm.match_accum_budget(0))
PY
cat gpurun_out/r05_n_mfma_f32_model.txt | grep -v amdgpu
