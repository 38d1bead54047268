#!/bin/bash
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_stackb -- python3 $ROOT/tools/stackb_prof.py > $ROOT/gpurun_out/rocprof_stackb.log 2>&1; echo "rocprof rc=$?" )
python tools/prof_summary.py gpurun_out/prof_stackb 25
