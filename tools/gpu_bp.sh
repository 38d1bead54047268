#!/bin/bash
# bench line + rocprofv3 kernel stats of the same command (no tests).  BENCH_ARGS / ENVS pass through.
set -u
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step', d['ms_per_step'], 'with opt', d.get('train_step_with_optimizer_ms'), 'roofline', r.get('kernel'), r['avg_launch_us'], 'us frac', r['frac'])"
ROOT=$(pwd); export TMPDIR=/tmp
rm -rf gpurun_out/prof
( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > $ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?" )
python tools/prof_summary.py
