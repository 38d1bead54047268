#!/bin/bash
# bench + rocprofv3 kernel stats only (no tests)
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_only.json 2> gpurun_out/bench_only.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/bench_only.json')); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
rm -rf gpurun_out/prof
( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?" )
python tools/prof_summary.py
