#!/bin/bash
mkdir -p gpurun_out
echo "=== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 3
for v in 0 1; do
  echo "MMDEER_NT8=$v"
  MMDEER_NT8=$v timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  MMDEER_NT8=$v timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
ROOT=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/rocprof.log 2>&1 ) || exit 1
python tools/prof_summary.py > gpurun_out/prof_summary.txt; grep "glds\|per step" gpurun_out/prof_summary.txt
