#!/usr/bin/env python3
"""Condense gpurun_out/<tag>/pass*/ (scripts/collect_placement.sh) into profiles/<tag>_counters.csv: per counter pass and per pair of
vectors (the manifest lines of scripts/bin/placement_lab5 tie dispatches to pairs) the mean dispatch duration of the pass itself and
the mean counter values, each pair classed fast / slow by its duration inside ITS pass; then, per counter, the mean over the fast and
over the slow pairs and their ratio -- the counter that differs names the cause."""
import collections
import csv
import glob
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r04_placement2"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", tag)
rows_out, per_counter = [], collections.defaultdict(lambda: {"fast": [], "slow": []})
for p in range(1, 40):
    f = glob.glob(os.path.join(root, f"pass{p}", "**", "*counter_collection.csv"), recursive=True)
    man = os.path.join(root, f"pass{p}.txt")
    if not f or not os.path.exists(man):
        continue
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "k_spmv_ldsp" in r["Kernel_Name"]:
            e = disp.setdefault(int(r["Dispatch_Id"]), {"dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    pairs = []
    for l in open(man):
        m = re.match(r"pair (\S+)\s+dispatches \[(\d+), (\d+)\) us ([\d.]+)", l)
        if not m or int(m.group(3)) > len(ids):
            continue
        sel = [disp[ids[i]] for i in range(int(m.group(2)) + 2, int(m.group(3)))]
        dur = sum(s["dur"] for s in sel) / len(sel) / 1e3
        cs = {k: sum(s.get(k, 0.0) for s in sel) / len(sel) for k in sel[0] if k != "dur"}
        pairs.append((m.group(1), dur, cs))
    if not pairs:
        continue
    lo, hi = min(d for _, d, _ in pairs), max(d for _, d, _ in pairs)
    for name, dur, cs in pairs:
        cls = "fast" if dur < lo + 0.35 * (hi - lo) else ("slow" if dur > lo + 0.65 * (hi - lo) else "between")
        for k, v in cs.items():
            rows_out.append((p, name, cls, round(dur, 1), k, v))
            if cls in ("fast", "slow") and hi - lo > 0.05 * lo:
                per_counter[k][cls].append((v, dur))
out = os.path.join(os.path.dirname(root), "..", "profiles", f"{tag}_counters.csv")
with open(out, "w") as fh:
    w = csv.writer(fh)
    w.writerow(["# scripts/collect_placement.sh + scripts/make_placement_summary.py: one rocprofv3 --pmc pass per counter group, one process each; "
                "class = the pair's dispatch duration inside its own pass"])
    w.writerow(["pass", "pair", "class", "dispatch_us", "counter", "mean_value_per_dispatch"])
    for r in rows_out:
        w.writerow(r)
    w.writerow([])
    w.writerow(["counter", "fast_pairs", "slow_pairs", "mean_fast", "mean_slow", "slow_over_fast", "mean_us_fast", "mean_us_slow"])
    for k, d in sorted(per_counter.items()):
        if d["fast"] and d["slow"]:
            mf = sum(v for v, _ in d["fast"]) / len(d["fast"]); ms = sum(v for v, _ in d["slow"]) / len(d["slow"])
            uf = sum(u for _, u in d["fast"]) / len(d["fast"]); us = sum(u for _, u in d["slow"]) / len(d["slow"])
            w.writerow([k, len(d["fast"]), len(d["slow"]), f"{mf:.6g}", f"{ms:.6g}", f"{ms / mf:.4f}" if mf else "", f"{uf:.1f}", f"{us:.1f}"])
            print(f"{k:48s} fast {mf:14.6g} slow {ms:14.6g} ratio {ms / mf if mf else float('nan'):7.4f}   ({len(d['fast'])} fast {uf:.0f} us, {len(d['slow'])} slow {us:.0f} us)")
print("->", os.path.normpath(out))
