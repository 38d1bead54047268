#!/usr/bin/env python3
"""Per-step kernel time table from the newest rocprofv3 kernel_stats.csv under gpurun_out/prof."""
import csv
import glob
import os
import sys

# usage: prof_summary.py [steps | profile_dir] [steps]
root = "gpurun_out/prof"
args = sys.argv[1:]
if args and not args[0].isdigit():
    root = args.pop(0)
files = sorted(glob.glob(root + "/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
if not files:
    sys.exit("no kernel_stats.csv")
steps = int(args[0]) if args else 25
rows = list(csv.DictReader(open(files[-1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{files[-1]}: GPU kernel time per step = {tot / steps / 1e3:.1f} us ({steps} steps)")
for r in rows[:24]:
    n = r["Name"].replace("void mmdeer::(anonymous namespace)::", "").replace("mmdeer::(anonymous namespace)::", "")[:72]
    print(f"{n:72s} n/step={int(r['Calls']) / steps:5.1f} avg={float(r['AverageNs']) / 1e3:7.2f}us step={float(r['TotalDurationNs']) / steps / 1e3:7.1f}us")
