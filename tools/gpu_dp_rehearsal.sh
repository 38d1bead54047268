#!/bin/bash
# 1-rank rehearsal of the data-parallel bench line on the one GPU of a gpurun box: the collectives really run (RCCL, one rank).
# bench.py starts its rank itself on a port the kernel hands out (spawn_ranks), stderr is kept beside every line.
#   plans timed inside one run: in-graph all-reduce, in-graph reduce-scatter + all-gather, (MMDEER_DP_OVERLAP=1) overlapped two-phase
OUT=gpurun_out/dp; mkdir -p $OUT
for tag in auto overlap; do
  if [ $tag = overlap ]; then export MMDEER_DP_OVERLAP=1; else unset MMDEER_DP_OVERLAP; fi
  MMDEER_FORCE_COMM=1 timeout -k 10 300 python bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_dp1_$tag.json 2> $OUT/bench_dp1_$tag.err; echo "dp $tag rc=$?"
  python - $OUT/bench_dp1_$tag.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); dp = d["data_parallel"]
    print(dp["plan"], dp["plan_ms"], dp["compute_only_ms"], dp.get("exposed_exchange_us"), dp["exchange_us_by_algo"], dp["backend"])
except Exception as e:      # noqa: BLE001
    print("no line:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-2000:])
PY
done
