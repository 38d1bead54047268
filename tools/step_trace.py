#!/usr/bin/env python3
"""One steady-state train step from a rocprofv3 kernel trace: every launch in order with its duration, the gap to the
previous kernel's end, grid and LDS.  usage: step_trace.py <kernel_trace.csv> [which-step-from-the-end] [name of the step's last kernel]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[3] if len(sys.argv) > 3 else "reduce_partials"
last = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]      # the step's final launch
s, e = last[-back - 1] + 1, last[-back] + 1
prev, tot = None, 0.0
for r in rows[s:e]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev) / 1e3 if prev else 0.0
    n = r["Kernel_Name"].replace("mmdeer::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print(f"{(en - st) / 1e3:7.2f} us  gap {gap:6.2f}  wgs {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):5d} x {r['Workgroup_Size_X']:>3}  lds {int(r['LDS_Block_Size']) // 1024:3d}K  {n}")
    prev = en
    tot += (en - st) / 1e3
print(f"sum of durations {tot:.1f} us, span {(int(rows[e - 1]['End_Timestamp']) - int(rows[s]['Start_Timestamp'])) / 1e3:.1f} us, {e - s} launches")
