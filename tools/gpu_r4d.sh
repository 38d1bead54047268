set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4d; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16_layers.py tests/test_gpu_configs.py -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
for ts in 0 16; do
  MMDEER_CHAIN_TS=$ts timeout -k 10 300 python bench.py --batch 8192 --no-cpu-baseline --no-autotune > $OUT/b8192_ts$ts.json 2> $OUT/b8192_ts$ts.err; echo "ts $ts rc=$?"
done
MMDEER_CHAIN=0 timeout -k 10 300 python bench.py --batch 8192 --no-cpu-baseline > $OUT/b8192_nochain.json 2> $OUT/b8192_nochain.err
for b in 512 1024 2048; do MMDEER_CHAIN_MIN=1 timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline > $OUT/b${b}_chainmin1.json 2> $OUT/b$b.err; done
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['launch_plan'].get('ms'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
