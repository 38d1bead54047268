#!/bin/bash
for b in 8192 1024 100 7; do
  echo "B=$b"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch $b 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"final_loss": [0-9.a-zN]*' | tr '\n' ' '; echo
done
echo "fp32 B=1024"; timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 1024 --dtype fp32 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"final_loss": [0-9.a-zN]*' | tr '\n' ' '; echo
