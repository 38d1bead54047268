#!/bin/bash
OUT=gpurun_out/pro; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16_layers.py tests/test_gpu_stackb.py -x -q > $OUT/t.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/t.txt
OLD=uncertainty-aware-multimodal-emotion-recognition_amd/libmmdeer_var_serialprologue.so
for r in 1 2; do
  timeout -k 10 300 python tools/bench_with_lib.py $OLD bench.py --no-cpu-baseline > $OUT/bench_old$r.json 2> $OUT/bench_old$r.err; echo "old rc=$?"
  timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_new$r.json 2> $OUT/bench_new$r.err; echo "new rc=$?"
done
python - <<'PY'
import json
for n in ("old1", "new1", "old2", "new2"):
    d = json.loads(open(f"gpurun_out/pro/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], {k[:8]: v for k, v in d.get("launch_us", {}).items() if "chain" in k})
PY
timeout -k 10 300 python tools/bench_with_lib.py $OLD tools/stackb_fused_time.py 2>/dev/null | head -2
timeout -k 10 300 python tools/stackb_fused_time.py 2>/dev/null | head -2
