#!/bin/bash
mkdir -p gpurun_out
python -c "from mmdeer import build; build.build_stamps()" > gpurun_out/build_stamps.log 2>&1 || tail -n 5 gpurun_out/build_stamps.log
timeout -k 10 200 python tools/gemm_stamps3.py 2>&1 | grep -v "amdgpu.ids"
