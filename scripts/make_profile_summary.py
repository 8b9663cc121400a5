#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the tracked files under profiles/.

    python scripts/make_profile_summary.py r01 gpurun_out/prof_r01 gpurun_out/pmc_r01_*

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim copy)
  profiles/<tag>_pmc_per_kernel.csv  mean counter value per kernel, one row per (kernel, counter)
  profiles/pmc_summary.json          HBM bytes per launch of the A.x kernel, corrected as
                                     MI355X_MICROARCH.md prescribes (FETCH_SIZE in KB, reports 1/2
                                     of wide coalesced reads on gfx950; WRITE_SIZE exact)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, trace_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    ks = glob.glob(os.path.join(trace_dir, "**", "*_kernel_stats.csv"), recursive=True)
    stats = {}
    if ks:
        shutil.copy(ks[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
        for r in csv.DictReader(open(ks[0])):
            stats[r["Name"]] = r
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in pmc_dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(out, f"{tag}_pmc_per_kernel.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "mean_per_dispatch", "dispatches"])
        for k in sorted(agg):
            for c in sorted(agg[k]):
                v = agg[k][c]
                w.writerow([k, c, sum(v) / len(v), len(v)])
    summary = {"tag": tag, "note": "FETCH_SIZE/WRITE_SIZE are KB per dispatch; on gfx950 FETCH_SIZE counts 1/2 of wide "
                                   "(16 B/lane) coalesced reads (calibrated here on the BLAS-1 kernels, whose byte "
                                   "counts are known exactly); the 8-byte x gather of A.x is uncalibrated, so the "
                                   "doubled figure is an upper bound"}
    for k in agg:
        if "k_spmv_lds" in k and "FETCH_SIZE" in agg[k]:
            f = sum(agg[k]["FETCH_SIZE"]) / len(agg[k]["FETCH_SIZE"]) * 1024
            wsz = sum(agg[k].get("WRITE_SIZE", [0])) / max(1, len(agg[k].get("WRITE_SIZE", [0]))) * 1024
            summary["spmv_kernel"] = k
            summary["spmv_fetch_size_raw_bytes"] = f
            summary["spmv_write_size_bytes"] = wsz
            summary["spmv_hbm_bytes_per_launch"] = 2 * f + wsz
            if "TCC_HIT_sum" in agg[k]:
                h = sum(agg[k]["TCC_HIT_sum"]) / len(agg[k]["TCC_HIT_sum"])
                m = sum(agg[k]["TCC_MISS_sum"]) / len(agg[k]["TCC_MISS_sum"])
                summary["spmv_l2_hit_rate"] = h / (h + m)
        for name, key in (("OpDot1", "calib_dot_160MB_read"), ("OpCgDir", "calib_dir_160MB_read_80MB_write")):
            if name in k and "FETCH_SIZE" in agg[k]:
                summary[key] = {"FETCH_SIZE_KB": sum(agg[k]["FETCH_SIZE"]) / len(agg[k]["FETCH_SIZE"]),
                                "WRITE_SIZE_KB": sum(agg[k].get("WRITE_SIZE", [0])) / max(1, len(agg[k].get("WRITE_SIZE", [0])))}
    for name, r in stats.items():
        if "k_spmv_lds" in name:
            summary["spmv_avg_ns_kernel_trace"] = float(r["AverageNs"])
            summary["spmv_calls_kernel_trace"] = int(r["Calls"])
    json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
