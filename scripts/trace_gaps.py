#!/usr/bin/env python3
"""Kernel durations AND the gaps in front of them from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv):
per kernel name the calls, mean duration and mean idle time between the end of the previous kernel on the device
and this one's start.  What a launch-bound iteration is made of.

  python scripts/trace_gaps.py gpurun_out/x/…_kernel_trace.csv [substring-filter]
"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
        wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "1"
        try:
            grid = str(int(grid) // max(1, int(wg)))
        except ValueError:
            pass
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"] + "", grid))
rows.sort()
flt = sys.argv[2] if len(sys.argv) > 2 else None
acc = defaultdict(lambda: [0, 0, 0, 0])
prev_end = None
for s, e, name, grid in rows:
    short = re.sub(r"\(.*", "", name).replace("void lcgh::", "").replace("lcgh::", "")
    short = re.sub(r"HIP_vector_type<double, 2u>", "double2", short) + f"  [{grid} wg]"
    a = acc[short]
    a[0] += 1; a[1] += e - s
    if prev_end is not None and s - prev_end < 200_000:      # gaps longer than 0.2 ms are host pauses, not launch gaps
        a[2] += max(0, s - prev_end); a[3] += 1
    prev_end = e if prev_end is None else max(prev_end, e)
print(f"{'calls':>7} {'dur_us':>8} {'gap_us':>8}  kernel")
for k, (n, d, g, gn) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if flt and flt not in k:
        continue
    print(f"{n:7d} {d / n / 1e3:8.2f} {g / max(gn, 1) / 1e3:8.2f}  {k[:150]}")
