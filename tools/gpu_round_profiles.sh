#!/bin/bash
# Everything profiles/ holds for one round, in one gpurun call (about 6 minutes of box time):
#   bench lines (default B=4096 bf16 with the CPU baseline; B=8192 bf16; B=1024 fp32; B=7 launch floor),
#   rocprofv3 --kernel-trace --stats of the default command, the --pmc passes over the fused kernel (separate passes),
#   and the in-kernel phase stamps of the diagnostic build.
# tools/collect_profiles.py then copies the summaries into profiles/ under round-numbered names.
set -u
ROOT=$(pwd); export TMPDIR=/tmp
OUT=gpurun_out/round; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python bench.py > $OUT/bench_b4096_bf16.json 2> $OUT/bench_b4096_bf16.err; echo "bench default rc=$?"
timeout -k 10 300 python bench.py --batch 8192 --no-cpu-baseline > $OUT/bench_b8192_bf16.json 2> $OUT/bench_b8192_bf16.err; echo "bench B=8192 rc=$?"
timeout -k 10 300 python bench.py --batch 1024 --dtype fp32 --no-cpu-baseline > $OUT/bench_b1024_fp32.json 2> $OUT/bench_b1024_fp32.err; echo "bench fp32 rc=$?"
timeout -k 10 300 python bench.py --batch 7 --no-cpu-baseline > $OUT/bench_b7_bf16.json 2> $OUT/bench_b7_bf16.err; echo "bench B=7 rc=$?"
timeout -k 10 200 python tools/fwd_only.py > $OUT/fwd_only.txt 2> $OUT/fwd_only.err; echo "fwd only rc=$?"
( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -- python3 $ROOT/bench.py --no-cpu-baseline > $ROOT/$OUT/rocprof.log 2>&1; echo "rocprof rc=$?" )
python tools/step_trace.py $OUT/prof/*/*_kernel_trace.csv 3 > $OUT/step_trace.txt 2>&1; echo "step trace rc=$?"
( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof8192 -- python3 $ROOT/bench.py --batch 8192 --no-cpu-baseline > $ROOT/$OUT/rocprof8192.log 2>&1; echo "rocprof B=8192 rc=$?" )
python tools/step_trace.py $OUT/prof8192/*/*_kernel_trace.csv 3 > $OUT/step_trace_b8192.txt 2>&1; cp $OUT/prof8192/*/*_kernel_stats.csv $OUT/kernel_stats_b8192.csv 2>/dev/null; rm -rf $OUT/prof8192
bash tools/gpu_pmc_fused.sh > $OUT/pmc.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc_fused_summary.txt gpurun_out/pmc_traffic.json $OUT/ 2>/dev/null
bash tools/gpu_pmc_step.sh > $OUT/pmc_step.log 2>&1; echo "pmc step rc=$?"
cp gpurun_out/pmc_step_summary.txt $OUT/ 2>/dev/null
timeout -k 10 200 python tools/tf_stamps.py > $OUT/tf_stamps.txt 2>&1; echo "stamps rc=$?"
timeout -k 10 200 python tools/chain_stamps.py 4096 > $OUT/chain_stamps.txt 2>&1; echo "chain stamps rc=$?"
timeout -k 10 200 python tools/chain_stamps.py 4096 bwd > $OUT/chain_stamps_bwd.txt 2>&1; echo "chain stamps (backward head chain) rc=$?"
timeout -k 10 120 tools/probes/calibrate > $OUT/calibration.txt 2>&1; echo "calibration rc=$?"
# data-parallel rehearsal on the one GPU (1-rank group, the collectives really run; bench.py starts its rank on a free port)
MMDEER_FORCE_COMM=1 MMDEER_DP_OVERLAP=0 timeout -k 10 300 python bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_dp1_single.json 2> $OUT/bench_dp1_single.err; echo "dp single rc=$?"
MMDEER_FORCE_COMM=1 MMDEER_DP_OVERLAP=1 timeout -k 10 300 python bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_dp1_overlap.json 2> $OUT/bench_dp1_overlap.err; echo "dp overlap rc=$?"
MMDEER_FORCE_COMM=1 timeout -k 10 300 python bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_dp1_auto.json 2> $OUT/bench_dp1_auto.err; echo "dp auto rc=$?"
MMDEER_CHAIN=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_b4096_nochain.json 2> $OUT/bench_b4096_nochain.err; echo "bench without chains rc=$?"
MMDEER_CHAIN_DEPTH=8 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_b4096_depth8.json 2> $OUT/bench_b4096_depth8.err; echo "bench with the 8-deep ring rc=$?"
MMDEER_ADAM_FUSED=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_b4096_adam_unfused.json 2> $OUT/bench_b4096_adam_unfused.err; echo "bench with update + repack rc=$?"
timeout -k 10 300 python bench.py --workload stackb_train > $OUT/bench_stackb_train.json 2> $OUT/bench_stackb_train.err; echo "bench Stack B training rc=$?"
# Stack B training step: chains against launch by launch, one steady-state step of the kernel trace, stamps of one encoder chain
bash tools/gpu_stackb_step.sh > $OUT/stackb_step.log 2>&1; echo "stack B step rc=$?"
cp gpurun_out/sb/step_trace.txt $OUT/stackb_step_trace.txt; cp gpurun_out/sb/time_chain.txt $OUT/stackb_train.txt; cp gpurun_out/sb/time_ops.txt $OUT/stackb_train_ops.txt; cp gpurun_out/sb/kernel_stats.csv $OUT/stackb_kernel_stats.csv
timeout -k 10 200 python tools/sb_chain_stamps.py 4096 1 > $OUT/stackb_chain_stamps.txt 2>&1; echo "stack B chain stamps rc=$?"
ls -la $OUT
