set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=gpurun_out/r4b; mkdir -p $OUT
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_d4.json 2> $OUT/bench_d4.err; echo "bench d4 rc=$?"
MMDEER_CHAIN_DEPTH=2 timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_d2.json 2> $OUT/bench_d2.err; echo "bench d2 rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench_d4b.json 2> $OUT/bench_d4b.err; echo "bench d4 again rc=$?"
timeout -k 10 300 python bench.py --batch 8192 --no-cpu-baseline > $OUT/bench_b8192.json 2> $OUT/bench_b8192.err; echo "bench 8192 rc=$?"
( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -- python3 $ROOT/bench.py --no-cpu-baseline --steps 50 > $ROOT/$OUT/rocprof.log 2>&1; echo "rocprof rc=$?" )
python tools/step_trace.py $OUT/prof/*/*_kernel_trace.csv 3 > $OUT/step_trace.txt 2>&1; echo "step trace rc=$?"
cp $OUT/prof/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['ms_per_step'], d['ms_per_step_median'], d['launch_plan'].get('ms'), d['roofline']['avg_launch_us'])
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
cat $OUT/step_trace.txt
