#!/bin/bash
mkdir -p gpurun_out
for sd in 0 1; do
  echo "MMDEER_SIDE=$sd"
  MMDEER_SIDE=$sd timeout -k 10 600 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 2
  MMDEER_SIDE=$sd timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*'
  MMDEER_SIDE=$sd timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --eager 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*'
done
