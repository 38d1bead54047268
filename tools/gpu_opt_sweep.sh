#!/bin/bash
# bench.py under a list of launch-plan options (each "NAME=VALUE[,NAME=VALUE]"), one line per setting: ms/step and the
# rocprofv3-free in-run roofline figure.  usage: gpu_opt_sweep.sh "" "SPLITK_MAX=4" "KSTEPS=24" ...
set -u
mkdir -p gpurun_out
for cfg in "$@"; do
  envs=""; for kv in ${cfg//,/ }; do envs="$envs MMDEER_$kv"; done
  env $envs timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/sweep.json 2> gpurun_out/sweep.err
  python - "$cfg" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/sweep.json") if l.startswith("{")][-1])
    print(f"{sys.argv[1] or 'default':28s} ms/step {d['ms_per_step']:.4f} (min {d['ms_per_step_min']:.4f} median {d['ms_per_step_median']:.4f})  roofline {d['roofline']['avg_launch_us']} us")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("gpurun_out/sweep.err").read()[-400:])
PY
done
