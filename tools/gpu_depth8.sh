#!/bin/bash
# 8-deep weight ring of the 16-sample chain kernel (option chain_depth = 8) against the 4-deep default: parity tests, Stack C bench, Stack B step.
OUT=gpurun_out/d8; mkdir -p $OUT
MMDEER_CHAIN_DEPTH=8 timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16_layers.py -x -q > $OUT/t_d8.txt 2>&1; echo "pytest d8 rc=$?"; tail -3 $OUT/t_d8.txt
MMDEER_CHAIN_DEPTH=8 timeout -k 10 300 python tools/stackb_chain_check.py 100 1024 > $OUT/check_d8.txt 2>&1; echo "check rc=$?"; grep -v "^  tape" $OUT/check_d8.txt | tail -8
for d in 4 8 4 8; do
  MMDEER_CHAIN_DEPTH=$d timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_d$d.json 2> $OUT/bench_d$d.err; echo "bench d=$d rc=$?"
  python - <<PY
import json
d = json.loads(open("$OUT/bench_d$d.json").read().strip().splitlines()[-1])
print("depth $d:", d["ms_per_step"], {k: v for k, v in d.get("launch_us", {}).items() if "chain" in k})
PY
done
MMDEER_CHAIN_DEPTH=8 timeout -k 10 300 python tools/stackb_fused_time.py > $OUT/sb_time_d8.txt 2> $OUT/sb_time_d8.err; echo "sb d8 rc=$?"; head -4 $OUT/sb_time_d8.txt
MMDEER_CHAIN_DEPTH=4 timeout -k 10 300 python tools/stackb_fused_time.py > $OUT/sb_time_d4.txt 2> $OUT/sb_time_d4.err; echo "sb d4 rc=$?"; head -4 $OUT/sb_time_d4.txt
