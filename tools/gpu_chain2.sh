#!/bin/bash
mkdir -p gpurun_out
ROOT=$(pwd); export TMPDIR=/tmp
for b in 512 4096; do
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_b$b -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch $b > $ROOT/gpurun_out/rocprof_b$b.log 2>&1 ) || exit 1
echo "B=$b"; python tools/prof_summary.py gpurun_out/prof_b$b | grep "chain\|64, 64"
done
