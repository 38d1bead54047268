#!/bin/bash
# interleaved A/B of an environment switch on the bench line: tools/ab_env.sh MMDEER_XCD_CHAIN [extra bench args]
v=$1; shift
for val in 0 1 0 1 0 1; do
  env $v=$val timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v=$val', d['ms_per_step'], d['train_step_with_optimizer_ms'], d['final_loss'])"
done
