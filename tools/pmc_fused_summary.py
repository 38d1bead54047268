#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 --pmc passes under gpurun_out/pmcf, and profiles/pmc_traffic.json: the HBM-side
bytes per launch of the forward kernel (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, MI355X_MICROARCH.md 'HBM') tied to the
sha of the kernel sources they were measured on -- bench.py reports `roofline.traffic` from it, null on any other code."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out/pmcf/p*/**/*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "tri_fused_kernel" not in k:
            continue
        name = "tri_fused_kernel<0> (forward)" if "<0>" in k else "tri_fused_kernel<1> (backward)"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name in sorted(acc):
    c = {n: sum(v[1:]) / max(len(v) - 1, 1) for n, v in acc[name].items()}     # first launch of each kind: cold caches
    print(name)
    for n in sorted(c):
        print(f"    {n:34s} {c[n]:16.1f}")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = 2 * c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
        print(f"    -> HBM-side traffic per launch: read {rd / 1e6:.2f} MB (FETCH_SIZE x 2), write {wr / 1e6:.2f} MB")
        out[name] = rd + wr
    if "TCC_HIT_sum" in c:
        print(f"    -> L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}; L2 requests {c['TCC_REQ_sum']:.0f}")
    if "TCP_TCC_READ_REQ_LATENCY_sum" in c and c.get("TCP_TCC_READ_REQ_sum"):
        print(f"    -> mean L1->L2 read latency {c['TCP_TCC_READ_REQ_LATENCY_sum'] / c['TCP_TCC_READ_REQ_sum']:.0f} cycles")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
        print(f"    -> MFMA busy cycles / (SQ busy cycles x 4 SIMDs / CU-count basis): {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
fwd = "tri_fused_kernel<0> (forward)"
if fwd in out:
    import bench
    rec = {"kernel": fwd, "dtype": "bf16", "batch": 4096, "traffic_bytes": round(out[fwd]), "src_sha": bench.kernel_sources_sha(),
           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/gpu_pmc_fused.sh), FETCH_SIZE doubled (gfx950)"}
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w"), indent=1)
    print("wrote gpurun_out/pmc_traffic.json (copy to profiles/ when the sources are final):", rec)
