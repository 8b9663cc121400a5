#!/usr/bin/env python3
"""Timeline of ONE K-step solve from a rocprofv3 --kernel-trace CSV: every kernel with its start relative to the solve's first
kernel, its duration and the idle gap in front of it.  The solve is found as the last run of kernels that holds exactly K
update passes (k_vec<OpCg1UpdateSums> / OpCgUpdate) between two host pauses of > 150 us.

  python scripts/solve_timeline.py <kernel_trace.csv> [K]
"""
import csv
import re
import sys

K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void lcgh::", "").replace("lcgh::", "")))
rows.sort()
segs, cur = [], []
for i, (s, e, name) in enumerate(rows):
    if cur and s - cur[-1][1] > 150_000:
        segs.append(cur); cur = []
    cur.append((s, e, name))
if cur:
    segs.append(cur)
pick = None
for seg in segs:
    if sum(1 for _, _, nm in seg if "Update" in nm) == K:
        pick = seg
if pick is None:
    print("no segment with", K, "update passes; segments:", [sum(1 for _, _, nm in sg if 'Update' in nm) for sg in segs][-20:])
    sys.exit(1)
t0 = pick[0][0]
prev = None
print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel   (solve = {(pick[-1][1] - t0) / 1e3:.1f} us, {len(pick)} kernels)")
for s, e, nm in pick:
    gap = 0.0 if prev is None else (s - prev) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {gap:7.2f}  {nm[:90]}")
    prev = e
