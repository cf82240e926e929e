cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 ./niftymatch_amd/lib/lds_atomic > gpurun_out/r06_b_lds_atomic.txt 2>&1 || exit 1
for i in 1 2; do
timeout -k 10 120 python tools/ksite.py describe 64 >> gpurun_out/r06_b_desc_ab.txt 2>&1 || exit 1
NM_DIAGNOSTIC=1 NM_HIP_LIB=$PWD/tools/_variants/libnm_hip_descatomic.so timeout -k 10 120 python tools/ksite.py describe 64 >> gpurun_out/r06_b_desc_ab.txt 2>&1 || exit 1
done
cat gpurun_out/r06_b_lds_atomic.txt gpurun_out/r06_b_desc_ab.txt
