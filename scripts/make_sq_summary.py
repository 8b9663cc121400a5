#!/usr/bin/env python3
"""Condense what scripts/collect_sq.sh left under gpurun_out/<tag>/pass*/ into profiles/<tag>.csv: one row per
(kernel, counter) with the mean value per launch and the number of launches it was taken over, plus derived rows
(`derived:*`) for the A.x kernels: share of wave cycles spent waiting, LDS bank-conflict share, instructions per wave.

    python scripts/make_sq_summary.py r03_tiled_sq [kernel substring ...]
"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03_tiled_sq"
    keep = sys.argv[2:] or ["k_tile_spmv", "k_spmv_", "k_bin_"]
    src = os.path.join(ROOT, "gpurun_out", tag)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(src, "pass*", "**", "*_counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(k in name for k in keep):
                continue
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = os.path.join(ROOT, "profiles", f"{tag}.csv")
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "mean_per_launch", "launches"])
        for name in sorted(agg):
            c = {k: sum(v) / len(v) for k, v in agg[name].items()}
            short = name.replace("void lcgh::", "").split("(")[0]
            for k in sorted(c):
                w.writerow([short, k, f"{c[k]:.6g}", len(agg[name][k])])

            def ratio(a, b):
                return c[a] / c[b] if a in c and b in c and c[b] else None
            for label, a, b in (("derived:wait_any_share_of_wave_cycles", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
                                ("derived:wait_inst_any_share_of_wave_cycles", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
                                ("derived:wait_inst_lds_share_of_wave_cycles", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"),
                                ("derived:active_inst_any_share_of_wave_cycles", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
                                ("derived:lds_bank_conflict_share_of_lds_active", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS"),
                                ("derived:lds_insts_per_wave", "SQ_INSTS_LDS", "SQ_WAVES"),
                                ("derived:valu_insts_per_wave", "SQ_INSTS_VALU", "SQ_WAVES"),
                                ("derived:vmem_rd_insts_per_wave", "SQ_INSTS_VMEM_RD", "SQ_WAVES"),
                                ("derived:l2_hit_rate", "TCC_HIT_sum", None)):
                if b is None:
                    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
                        w.writerow([short, label, f"{c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.4f}", ""])
                    continue
                v = ratio(a, b)
                if v is not None:
                    w.writerow([short, label, f"{v:.4f}", ""])
    print(out)
    print(open(out).read())


if __name__ == "__main__":
    main()
