#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py -m gpu -q -p no:cacheprovider -k fused_adamw 2>&1 | grep -B2 -A12 "AssertionError\|assert worst" | head -40
