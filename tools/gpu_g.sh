#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 15
