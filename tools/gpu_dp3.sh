#!/bin/bash
mkdir -p gpurun_out
for ov in 0 1; do
  echo "=== 1-rank RCCL rehearsal, MMDEER_DP_OVERLAP=$ov"
  MMDEER_DP_OVERLAP=$ov MMDEER_FORCE_COMM=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/dpo_$ov.log 2>&1; echo "rc=$?"
  grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*\|"grad_exchange": "[^"]*"' gpurun_out/dpo_$ov.log | tr '\n' ' '; echo; grep "\[bench\]\|Error\|error" gpurun_out/dpo_$ov.log | cut -c1-300 | head -5
done
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_trainer.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 3
