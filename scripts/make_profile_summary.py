#!/usr/bin/env python3
"""Condense what scripts/collect_profiles.sh left under gpurun_out/profiles_<tag>/ into the tracked files under profiles/.

    python scripts/make_profile_summary.py r02

  profiles/<tag>_bench_n1.json             the bench line of that box
  profiles/<tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline` (headline + variants)
  profiles/<tag>_configs_kernel_stats.csv  the same for scripts/bench_configs.py c1 c2 c5 (configs[0], [1], [4])
  profiles/<tag>_configs.jsonl             un-profiled numbers of those configurations
  profiles/<tag>_ax_variants.jsonl         A.x on each pattern with each kernel family (plain / binned / tiled / automatic)
  profiles/<tag>_pmc_per_kernel.csv        mean counter value per kernel, one row per (kernel, counter)
  profiles/pmc_summary.json                per pattern: the A.x kernel(s), HBM bytes per launch corrected as MI355X_MICROARCH.md
                                           prescribes (FETCH_SIZE in KB and worth 1/2 of wide coalesced reads on gfx950,
                                           WRITE_SIZE exact), L2 hit rate; bench.py reads `variants[pattern]`
"""
import collections
import csv
import datetime
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILY = {"constant_diagonals": ("k_spmv_ldsp",), "row_random_band": ("k_tile_spmv",), "scrambled": ("k_bin_expand", "k_bin_reduce")}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)

    def first(pattern):
        g = glob.glob(os.path.join(src, pattern), recursive=True)
        return g[0] if g else None
    for a, b in (("bench.json", f"{tag}_bench_n1.json"), ("configs.jsonl", f"{tag}_configs.jsonl"), ("ax_variants.jsonl", f"{tag}_ax_variants.jsonl")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(out, b))
    stats = {}
    for d, name in (("trace_bench", f"{tag}_kernel_stats.csv"), ("trace_configs", f"{tag}_configs_kernel_stats.csv")):
        f = first(f"{d}/**/*_kernel_stats.csv")
        if f:
            shutil.copy(f, os.path.join(out, name))
            if d == "trace_bench":
                stats = {r["Name"]: r for r in csv.DictReader(open(f))}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))        # every pass (the per-kernel table, the calibration)
    aggv = collections.defaultdict(lambda: collections.defaultdict(list))       # the passes on scripts/ax_variants.py only: the per-pattern traffic
    for f in glob.glob(os.path.join(src, "pmc*", "**", "*_counter_collection.csv"), recursive=True):
        on_variants = os.path.relpath(f, src).startswith("pmc_")
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if on_variants:
                aggv[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(out, f"{tag}_pmc_per_kernel.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "mean_per_dispatch", "dispatches"])
        for k in sorted(agg):
            for c in sorted(agg[k]):
                v = agg[k][c]
                w.writerow([k, c, sum(v) / len(v), len(v)])

    def mean(k, c, table=None):
        v = (agg if table is None else table)[k].get(c)
        return sum(v) / len(v) if v else None
    summary = {"tag": tag, "collected": datetime.date.today().isoformat(),
               "note": "FETCH_SIZE / WRITE_SIZE are KB per dispatch; on gfx950 FETCH_SIZE counts 1/2 of wide (16 B/lane) coalesced reads "
                       "(calibrated on the BLAS-1 kernels below, whose byte counts are known exactly), so streaming kernels are "
                       "corrected x2; for the row-block kernels' 8-byte x gathers the doubled figure is an upper bound",
               "variants": {}}
    sizes = {}      # pattern -> (rows, nnz) of the matrices the counters were collected on
    described = {}  # pattern -> the library's own description of the kernel that ran (lcg_hip_csr_last_kernel)
    for f in glob.glob(os.path.join(src, "pmc_FETCH_SIZE.jsonl")):
        for line in open(f):
            try:
                e = json.loads(line)
                sizes[e["pattern"]] = (e["rows"], e["nnz"])
                described[e["pattern"]] = e.get("kernel", "")
            except (ValueError, KeyError):
                pass
    for pattern, fams in FAMILY.items():
        ks = [k for k in aggv if any(f in k for f in fams) and "FETCH_SIZE" in aggv[k]]
        if not ks:
            continue
        mean_v = lambda k, c: mean(k, c, aggv)
        fetch = sum(mean_v(k, "FETCH_SIZE") for k in ks) * 1024
        write = sum((mean_v(k, "WRITE_SIZE") or 0.0) for k in ks) * 1024
        ent = {"kernel": " + ".join(k.split("(")[0].replace("void ", "") for k in ks), "fetch_size_raw_bytes": fetch, "write_size_bytes": write,
               "hbm_bytes_per_launch": 2 * fetch + write,
               "collected": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on scripts/ax_variants.py, {summary['collected']}"}
        if pattern in sizes:
            ent["rows"], ent["nnz"] = sizes[pattern]
            ent["kernel_description"] = described.get(pattern, "")
        hit = sum((mean_v(k, "TCC_HIT_sum") or 0.0) for k in ks); miss = sum((mean_v(k, "TCC_MISS_sum") or 0.0) for k in ks)
        if hit + miss > 0:
            ent["l2_hit_rate"] = hit / (hit + miss)
        req = sum((mean_v(k, "TCC_EA0_RDREQ_sum") or 0.0) for k in ks)
        if req:
            ent["l2_fabric_read_requests"] = req
        for name, r in stats.items():
            if any(f in name for f in fams):
                ent.setdefault("avg_ns_kernel_trace", {})[name.split("(")[0].replace("void ", "")] = float(r["AverageNs"])
        summary["variants"][pattern] = ent
    for k in agg:
        for name, key in (("OpDot1", "calib_dot_160MB_read"), ("OpCgDir", "calib_dir_160MB_read_80MB_write")):
            if name in k and "FETCH_SIZE" in agg[k]:
                summary[key] = {"FETCH_SIZE_KB": mean(k, "FETCH_SIZE"), "WRITE_SIZE_KB": mean(k, "WRITE_SIZE")}
    json.dump(summary, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
