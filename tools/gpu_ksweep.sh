#!/bin/bash
# split-K depth / ring depth sweep of the weight-gradient launches (per-kernel times from rocprofv3)
mkdir -p gpurun_out
ROOT=$(pwd); export TMPDIR=/tmp
for ks in 8 16 32; do
  for ring in 2 3; do
    echo "=== MMDEER_KSTEPS=$ks MMDEER_TTRING=$ring"
    export MMDEER_KSTEPS=$ks MMDEER_TTRING=$ring
    ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/ks_${ks}_${ring} -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/ks_${ks}_${ring}.log 2>&1 ) || exit 1
    grep -o '"ms_per_step": [0-9.]*' gpurun_out/ks_${ks}_${ring}.log
    python tools/prof_summary.py gpurun_out/ks_${ks}_${ring} 2>/dev/null | grep "tt_glds\|reduce_partials\|true, true\|per step"
  done
done
