#!/bin/bash
# One GPU-box round: parity tests, smoke, short bench.  Logs go to gpurun_out/.
# A step that times out (124/137) stops the round: no further GPU step after a hang.
set -u
mkdir -p gpurun_out
step() {  # name, timeout, command...
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/round.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/round.log
  tail -n 25 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a gpurun_out/round.log; exit $rc; fi
  return 0
}
: > gpurun_out/round.log
step build 300 python __graft_entry__.py
step pytest_ops 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider
step pytest_model 600 python -m pytest tests/test_gpu_model.py -m gpu -q -p no:cacheprovider
step smoke 300 python __graft_entry__.py smoke
step bench 600 python bench.py --steps 20 --warmup 5
step gemm_bench 600 python tools/gemm_bench.py
head -70 gpurun_out/gemm_bench.log
ROOT=$(pwd)
export TMPDIR=/tmp
( cd /tmp && step_dir=$ROOT/gpurun_out && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/rocprof.log 2>&1 ; echo "rocprof rc=$?" | tee -a $ROOT/gpurun_out/round.log )
find gpurun_out/prof -name "*stats*" | head; tail -n 5 gpurun_out/rocprof.log
step stamps 300 python tools/gemm_stamps.py
cat gpurun_out/stamps.log
