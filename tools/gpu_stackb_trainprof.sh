#!/bin/bash
ROOT=$(pwd); export TMPDIR=/tmp; rm -rf gpurun_out/prof_sbt
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_sbt -- python3 $ROOT/tools/stackb_train_prof.py ${1:-4096} ${2:-torch} > $ROOT/gpurun_out/sbt.log 2>&1; echo "rc=$?" )
python - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_sbt/**/*_kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
steps = 24.0   # 3 warm-up + 1 capture-less + 20 replays (approx.)
tot_calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernels per step ~ {tot_calls / steps:.0f}, GPU time per step ~ {tot / steps / 1e3:.0f} us")
for r in rows[:40]:
    n = r["Name"].replace("void mmdeer::(anonymous namespace)::", "").replace("mmdeer::(anonymous namespace)::", "")[:80]
    print(f"{n:80s} n/step={int(r['Calls']) / steps:6.1f} avg={float(r['AverageNs']) / 1e3:7.2f}us step={float(r['TotalDurationNs']) / steps / 1e3:7.1f}us")
PY
