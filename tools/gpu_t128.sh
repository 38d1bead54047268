#!/bin/bash
for t in 512 1024 100000; do
  echo "MMDEER_T128=$t"
  MMDEER_T128=$t timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  MMDEER_T128=$t timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
