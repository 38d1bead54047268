#!/usr/bin/env python3
"""Copy the summaries of tools/gpu_round_profiles.sh (gpurun_out/round/) into profiles/ under round-numbered names and write
the metadata bench.py checks before quoting them (which kernel sources, batch and dtype they were measured on).
usage: collect_profiles.py r04"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else bench.PROFILE_TAG
src, dst = os.path.join(ROOT, "gpurun_out", "round"), os.path.join(ROOT, "profiles")
sha = bench.kernel_sources_sha()
for name in ("bench_b4096_bf16.json", "bench_b8192_bf16.json", "bench_b1024_fp32.json", "bench_b7_bf16.json", "step_trace.txt",
             "pmc_fused_summary.txt", "pmc_step_summary.txt", "tf_stamps.txt", "chain_stamps.txt", "chain_stamps_bwd.txt", "fwd_only.txt", "calibration.txt",
             "stackb_train.txt", "stackb_train_ops.txt", "stackb_step_trace.txt", "stackb_kernel_stats.csv", "stackb_chain_stamps.txt", "bench_stackb_train.json", "bench_b4096_depth8.json", "bench_b4096_adam_unfused.json", "step_trace_b8192.txt", "kernel_stats_b8192.csv", "bench_dp1_single.json", "bench_dp1_overlap.json", "bench_dp1_auto.json", "bench_b4096_nochain.json"):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
        print("copied", name)
stats = sorted(glob.glob(os.path.join(src, "prof", "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)   # newest run
if stats:
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    line = json.load(open(os.path.join(src, "bench_b4096_bf16.json")))
    meta = {"csv": f"{tag}_bench_kernel_stats.csv", "src_sha": sha, "batch": 4096, "dtype": "bf16",
            "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline",
            "bench_line_of_the_same_box": {k: line[k] for k in ("value", "ms_per_step", "steps", "warmup")}}
    # bytes of the weight-gradient fold (slabs in + gradient out) from the PMC pass over the step, when it ran on these sources
    try:
        for ln in open(os.path.join(src, "pmc_step_summary.txt")):
            f = ln.split()
            if f and f[0] == "reduce_partials_kernel":
                meta["reduce_bytes"] = int((float(f[3]) + float(f[4])) * 1e6)
    except (OSError, ValueError, IndexError):
        pass
    json.dump(meta, open(os.path.join(dst, f"{tag}_bench_kernel_stats.json"), "w"), indent=1)
    print("kernel stats ->", meta["csv"], "sha", sha)
pt = os.path.join(src, "pmc_traffic.json")
if os.path.exists(pt):
    rec = json.load(open(pt))
    if rec.get("src_sha") != sha:
        print("WARNING: pmc_traffic.json was measured on other kernel sources:", rec.get("src_sha"), "!=", sha)
    json.dump(rec, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
    print("pmc traffic ->", rec)
