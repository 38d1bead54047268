#!/bin/bash
# rocprofv3 --pmc passes over every kernel of one training step (eager launches), one counter set per pass (gfx950 slot limits;
# FETCH_SIZE / WRITE_SIZE cannot share a pass).  Summary by tools/pmc_step_summary.py -> gpurun_out/pmc_step_summary.txt
ROOT=$(pwd); export TMPDIR=/tmp; rm -rf gpurun_out/pmcs; mkdir -p gpurun_out/pmcs
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmcs/p$i -- python3 $ROOT/tools/pmc_step.py ${1:-4096} > $ROOT/gpurun_out/pmcs/p$i.log 2>&1; echo "pass $i rc=$?" )
done
python tools/pmc_step_summary.py | tee gpurun_out/pmc_step_summary.txt
