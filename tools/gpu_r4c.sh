set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4c; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 200 python tools/chain_stamps.py 4096 > $OUT/chain_stamps.txt 2>&1; echo "stamps rc=$?"
timeout -k 10 200 python tools/chain_stamps.py 4096 bwd > $OUT/chain_stamps_bwd.txt 2>&1; echo "stamps bwd rc=$?"
bash tools/gpu_dp_rehearsal.sh
