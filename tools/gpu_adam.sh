#!/bin/bash
OUT=gpurun_out/adam; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py -x -q > $OUT/t_tr.txt 2>&1; echo "pytest rc=$?"; tail -4 $OUT/t_tr.txt
for f in 1 0; do
  MMDEER_ADAM_FUSED=$f timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_f$f.json 2> $OUT/bench_f$f.err; echo "bench adam_fused=$f rc=$?"
  python - <<PY
import json
d = json.loads(open("$OUT/bench_f$f.json").read().strip().splitlines()[-1])
print("adam_fused=$f:", d["ms_per_step"], {k: d[k] for k in d if "optim" in k or "with_opt" in k})
PY
done
