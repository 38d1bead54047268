#!/bin/bash
mkdir -p gpurun_out
for gc in 0 1; do
  echo "=== 1-rank RCCL rehearsal, MMDEER_GRAPH_COMM=$gc"
  MMDEER_GRAPH_COMM=$gc MMDEER_FORCE_COMM=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/dpg_$gc.log 2>&1; echo "rc=$?"
  grep -o '"ms_per_step": [0-9.]*\|"final_loss": [0-9.]*' gpurun_out/dpg_$gc.log | tr '\n' ' '; echo; grep "\[bench\]" gpurun_out/dpg_$gc.log | cut -c1-300
done
