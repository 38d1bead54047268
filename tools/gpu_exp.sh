#!/bin/bash
mkdir -p gpurun_out
for x in 0 1; do
  echo "=== MMDEER_XCD=$x"
  MMDEER_XCD=$x timeout -k 10 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | grep bfloat16 | tee gpurun_out/gemm_xcd$x.log
done
