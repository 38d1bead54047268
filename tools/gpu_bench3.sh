#!/bin/bash
# round-3 bench checks: default line, 1-rank data-parallel rehearsals (plan selection), B = 7 launch floor (plain and under rocprofv3)
set -u
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/b3_default.json 2> gpurun_out/b3_default.err; echo "default rc=$?"
MMDEER_FORCE_COMM=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b3_dp1.log 2>&1; echo "dp1 rc=$?"
MMDEER_DP_OVERLAP=1 MMDEER_FORCE_COMM=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/b3_dp1_overlap.log 2>&1; echo "dp1 overlap rc=$?"
TAG=b7 BENCH_ARGS="--batch 7" bash tools/gpu_bp.sh > gpurun_out/b3_b7.txt 2>&1; echo "b7 rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/b3_default.json", "gpurun_out/b3_dp1.log", "gpurun_out/b3_dp1_overlap.log"):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d["roofline"]
        print(f, d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_median"], d["ms_per_step_max"], "roof", r["avg_launch_us"], r["min_launch_us"], r["frac"], d.get("data_parallel"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -n 12 gpurun_out/b3_b7.txt
