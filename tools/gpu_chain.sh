#!/bin/bash
mkdir -p gpurun_out
echo "=== tests (chain)"; timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_trainer.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 5
ROOT=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/rocprof.log 2>&1 ) || exit 1
python tools/prof_summary.py > gpurun_out/prof_summary.txt; sed -n 1,12p gpurun_out/prof_summary.txt; grep chain gpurun_out/prof_summary.txt
for c in 1 0; do
  echo "MMDEER_CHAIN=$c"; MMDEER_CHAIN=$c timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*'
done
