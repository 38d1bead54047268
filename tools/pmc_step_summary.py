#!/usr/bin/env python3
"""Per-kernel averages (over the dispatches of the last 4 of 6 eager steps) of the rocprofv3 --pmc passes under
gpurun_out/pmcs: HBM-side bytes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md 'HBM'), L2 hit rate, MFMA share."""
import csv
import glob
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out/pmcs/p*/**/*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    per = defaultdict(list)
    for r in rows:
        per[(r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, grid, cn), vals in per.items():
        if "mmdeer" not in k:
            continue
        name = k.replace("void mmdeer::(anonymous namespace)::", "").replace("mmdeer::(anonymous namespace)::", "").split("(")[0]
        n = len(vals)
        acc[(name, grid)][cn].extend(vals[n // 3:])          # drop the first third: cold steps
print(f"{'kernel':44s} {'grid':>9s} {'calls':>5s} {'read MB':>8s} {'write MB':>8s} {'L2 hit':>7s} {'MFMA busy/SQ busy':>18s} {'LDS conflict cyc':>17s}")
for key in sorted(acc, key=lambda k: -sum(acc[k].get("FETCH_SIZE", [0]))):
    c = {n: sum(v) / len(v) for n, v in acc[key].items()}
    rd = 2 * c.get("FETCH_SIZE", float("nan")) * 1024 / 1e6
    wr = c.get("WRITE_SIZE", float("nan")) * 1024 / 1e6
    hit = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1) if "TCC_HIT_sum" in c else float("nan")
    mf = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"] if c.get("SQ_BUSY_CYCLES") else float("nan")
    print(f"{key[0][:44]:44s} {key[1]:>9s} {len(acc[key].get('FETCH_SIZE', [])):5d} {rd:8.2f} {wr:8.2f} {hit:7.3f} {mf:18.3f} {c.get('SQ_LDS_BANK_CONFLICT', float('nan')):17.0f}")
