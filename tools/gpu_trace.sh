set -u
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=gpurun_out/trace; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -- python3 $ROOT/bench.py --no-cpu-baseline --steps 50 $* > $ROOT/$OUT/rocprof.log 2>&1; echo "rocprof rc=$?" )
python tools/step_trace.py $OUT/prof/*/*_kernel_trace.csv 3 > $OUT/step_trace.txt 2>&1; echo "step trace rc=$?"
cp $OUT/prof/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
cat $OUT/step_trace.txt
timeout -k 10 200 python tools/chain_stamps.py 4096 > $OUT/chain_stamps.txt 2>&1
