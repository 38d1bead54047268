#!/bin/bash
# rocprofv3 --pmc passes over the fused projection + attention kernels (one counter set per pass: the SQ / TCC slot limits
# of gfx950, and FETCH_SIZE / WRITE_SIZE cannot share a pass).  Summary + profiles/pmc_traffic.json by tools/pmc_fused_summary.py.
ROOT=$(pwd); export TMPDIR=/tmp; rm -rf gpurun_out/pmcf; mkdir -p gpurun_out/pmcf
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUSY_sum TCC_CYCLE_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmcf/p$i -- python3 $ROOT/tools/pmc_fused.py > $ROOT/gpurun_out/pmcf/p$i.log 2>&1; echo "pass $i rc=$?" )
done
python tools/pmc_fused_summary.py | tee gpurun_out/pmc_fused_summary.txt
