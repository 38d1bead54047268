#!/bin/bash
# tests + bench + rocprof kernel stats (no microbenchmarks)
set -u
mkdir -p gpurun_out
run() { local name=$1 to=$2; shift 2; echo "=== $name"; timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?; echo "rc=$rc"; grep -v "amdgpu.ids\|UserWarning\|Consider using\|float(ld\|^$" "gpurun_out/$name.log" | tail -n 12; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
run pytest_gpu 900 python -m pytest tests -m gpu -q -p no:cacheprovider
run smoke 300 python __graft_entry__.py smoke
run bench 600 python bench.py --steps 30 --warmup 10 ${BENCH_ARGS:-}
echo "=== dp rehearsal (1 rank, RCCL collectives forced)"
MMDEER_FORCE_COMM=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/dp1.log 2>&1; echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/dp1.log | tail -n 4
echo "=== dp rehearsal (1 rank, mmdeer_comm_* communicator of the C ABI)"
MMDEER_COMM=rccl MMDEER_FORCE_COMM=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/dp1_rccl.log 2>&1; echo "rc=$?"; grep "^{" gpurun_out/dp1_rccl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['grad_exchange'])"
echo "=== dp rehearsal (1 rank, exact-global loss statistics exchanged inside the graph)"
MMDEER_EXACT_GLOBAL=1 MMDEER_FORCE_COMM=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/dp1_exact.log 2>&1; echo "rc=$?"; grep "^{" gpurun_out/dp1_exact.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['grad_exchange'], d['final_loss'])"
ROOT=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?" )
python tools/prof_summary.py
