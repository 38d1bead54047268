#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python tools/gemm_stamps4.py 2>&1 | grep -v "amdgpu.ids"
