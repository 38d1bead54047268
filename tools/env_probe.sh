run() { echo -n "$1 : "; env $1 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for b in "--batch 7" ""; do
echo "== $b"
run "X=1" "$b"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "$b"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "$b"
run "AMD_OPT_FLUSH=0" "$b"
run "DEBUG_HIP_GRAPH_BATCH_SIZE=64" "$b"
run "GPU_FLUSH_ON_EXECUTION=0" "$b"
run "HIP_FORCE_DEV_KERNARG=0" "$b"
run "X=2" "$b"
done
