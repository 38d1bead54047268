#!/bin/bash
mkdir -p gpurun_out
echo "=== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 8
ROOT=$(pwd); export TMPDIR=/tmp
for ks in 0 8 32; do
  echo "=== MMDEER_KSTEPS=$ks"
  export MMDEER_KSTEPS=$ks
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/dw_${ks} -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/dw_${ks}.log 2>&1 ) || exit 1
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/dw_${ks}.log
  python tools/prof_summary.py gpurun_out/dw_${ks} 2>/dev/null | grep "tt\|reduce_partials\|true, true\|per step\|nig_"
done
