#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_model.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 12
timeout -k 10 300 python tools/host_time.py 2>&1 | grep "host enqueue"
timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"optimizer_ms": [0-9.]*'
