#!/usr/bin/env python3
"""Print the kernel timeline of one steady-state bench step from a rocprofv3 kernel trace CSV.
usage: step_timeline.py <kernel_trace.csv> [step_index]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "spread_pad_kernel" in r["Kernel_Name"] or "spread_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[k], idx[k + 1]
# a step starts with the kernels in front of the spread (max|c|, scale): walk back to the previous interp
while a > 0 and "interp" not in rows[a - 1]["Kernel_Name"]:
    a -= 1
while b > 0 and "interp" not in rows[b - 1]["Kernel_Name"]:
    b -= 1
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:6.1f}  {r['Kernel_Name'][:100]}")
    prev_end = e
    busy += e - s
print(f"step span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, {b - a} launches")
