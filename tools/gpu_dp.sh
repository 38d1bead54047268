#!/bin/bash
mkdir -p gpurun_out
for pl in fp32 bf16; do
  echo "=== 1-rank RCCL rehearsal, payload $pl"
  MMDEER_FORCE_COMM=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --grad-comm $pl > gpurun_out/dp_$pl.log 2>&1; echo "rc=$?"
  grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*\|"parallelism": "[^"]*"' gpurun_out/dp_$pl.log | tr '\n' ' '; echo
done
