# usage (GPU box): bash tools/pmc_desc_stalls.sh <tag>  -- where frame_desc_kernel's wave cycles go: SQ counters over tools/ksite.py describe 64
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > gpurun_out/${T}_sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_IFETCH SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_RD" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d gpurun_out/${T}_p$i --output-format csv -- python3 tools/ksite.py describe 64 > /dev/null 2>&1
  python tools/pmc_counters.py gpurun_out/${T}_p$i frame_desc >> gpurun_out/${T}_desc_stall_counters.txt 2>&1
  rm -rf gpurun_out/${T}_p$i
done
cat gpurun_out/${T}_desc_stall_counters.txt
