#!/bin/bash
# A/B: device-side kernargs, after the epilogue / RNG change
mkdir -p gpurun_out
python -c "from mmdeer import build; build.build_stamps()" > gpurun_out/build_stamps.log 2>&1 || tail -n 5 gpurun_out/build_stamps.log
echo "=== tests"; timeout -k 10 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 8
for k in 0 1; do
  echo "=== HIP_FORCE_DEV_KERNARG=$k"
  HIP_FORCE_DEV_KERNARG=$k timeout -k 10 200 python tools/gemm_stamps2.py 2>&1 | grep "^---"
  HIP_FORCE_DEV_KERNARG=$k timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
  HIP_FORCE_DEV_KERNARG=$k timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
done
