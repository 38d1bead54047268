#!/bin/bash
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/pmc
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_LDS TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc/p$i -- python3 $ROOT/tools/pmc_gemm.py > $ROOT/gpurun_out/pmc/p$i.log 2>&1; echo "pass $i rc=$?" )
done
python tools/pmc_summary.py
