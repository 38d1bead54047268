#!/bin/bash
mkdir -p gpurun_out
echo "=== tests"; timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 8
for r in 2 3; do
  echo "=== MMDEER_TTRING=$r"
  MMDEER_TTRING=$r timeout -k 10 300 python tools/gemm_bench.py 2>&1 | grep "bfloat16" | grep "dW" | grep "tile=2"
  MMDEER_TTRING=$r timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
