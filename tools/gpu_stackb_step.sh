#!/bin/bash
# Stack B fused training step: timing (chains / launch by launch) + one steady-state step of the rocprofv3 kernel trace.
ROOT=$(pwd); export TMPDIR=/tmp; OUT=gpurun_out/sb; mkdir -p $OUT; rm -rf $OUT/prof
timeout -k 10 300 python tools/stackb_fused_time.py > $OUT/time_chain.txt 2> $OUT/time_chain.err; echo "time chain rc=$?"
SB_PLAN=ops timeout -k 10 300 python tools/stackb_fused_time.py > $OUT/time_ops.txt 2> $OUT/time_ops.err; echo "time ops rc=$?"
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -- python3 $ROOT/tools/stackb_train_prof.py ${1:-4096} fused > $ROOT/$OUT/prof.log 2>&1; echo "rocprof rc=$?" )
python tools/step_trace.py $OUT/prof/*/*_kernel_trace.csv 3 repack_kernel > $OUT/step_trace.txt 2>&1; echo "trace rc=$?"
cp $OUT/prof/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null; rm -rf $OUT/prof
cat $OUT/time_chain.txt $OUT/time_ops.txt; tail -4 $OUT/step_trace.txt
