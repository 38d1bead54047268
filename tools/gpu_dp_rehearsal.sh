for ov in 0 1; do MMDEER_FORCE_COMM=1 MMDEER_DP_OVERLAP=$ov timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2954$ov bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); dp=d['data_parallel']; print(dp['plan'], dp['plan_ms'], dp['compute_only_ms'], dp['exposed_exchange_us'])"; done
