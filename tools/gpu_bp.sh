#!/bin/bash
# bench line + rocprofv3 kernel stats of the same command (no tests).  BENCH_ARGS / ENVS pass through.
# Every call keeps its own logs: TAG (default: a timestamp) names gpurun_out/bench_$TAG.log, rocprof_$TAG.log and
# prof_$TAG/, so a failing run's stderr is never overwritten by the next one (an abort whose log was lost).
set -u
TAG=${TAG:-$(date +%H%M%S)}
mkdir -p gpurun_out
echo "tag=$TAG args='${BENCH_ARGS:-}'"
timeout -k 10 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/bench_$TAG.log 2>&1; rc=$?; echo "bench rc=$rc"
if [ $rc -ne 0 ]; then grep -v amdgpu.ids gpurun_out/bench_$TAG.log | tail -n 30; exit $rc; fi
grep "^{" gpurun_out/bench_$TAG.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step', d['ms_per_step'], 'with opt', d.get('train_step_with_optimizer_ms'), 'roofline', r.get('kernel', '')[:24], r.get('avg_launch_us'), 'us frac', r.get('frac'))"
ROOT=$(pwd); export TMPDIR=/tmp
rm -rf gpurun_out/prof_$TAG
( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} > $ROOT/gpurun_out/rocprof_$TAG.log 2>&1; rc=$?; echo "rocprof rc=$rc"; if [ $rc -ne 0 ]; then grep -v amdgpu.ids $ROOT/gpurun_out/rocprof_$TAG.log | tail -n 30; fi )
python tools/prof_summary.py gpurun_out/prof_$TAG
