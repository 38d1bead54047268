set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
for b in 64 128 256; do MMDEER_CHAIN_MIN=1 timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline > $OUT/b${b}.json 2> $OUT/b$b.err; done
for b in 12288 16384 32768; do MMDEER_CHAIN_MAX=1000000 timeout -k 10 300 python bench.py --batch $b --no-cpu-baseline --steps 50 > $OUT/b${b}.json 2> $OUT/b$b.err; done
for f in $OUT/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['launch_plan'].get('ms'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
