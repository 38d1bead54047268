#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_side.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 30
