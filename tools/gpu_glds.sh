#!/bin/bash
mkdir -p gpurun_out
echo "=== ops+model tests (glds on)"; timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 15
for x in 0 1; do
  echo "=== MMDEER_GLDS=$x"
  MMDEER_GLDS=$x timeout -k 10 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids | grep "bfloat16" | grep "fwd" | tee gpurun_out/gemm_glds$x.log
done
echo "=== bench"; timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -n 2
