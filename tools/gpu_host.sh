#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/host_time.py 2>&1 | grep -v "amdgpu.ids" | cut -c1-150
