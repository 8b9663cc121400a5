#!/usr/bin/env python3
"""A.x on the three column patterns of the synthetic family at the benchmark size, each with the row-block kernels
and with the binned two-pass product: time per product (HIP events over `reps` back-to-back calls), algorithmic
GB/s (12 nnz + 4 (N+1) + 16 N) and its fraction of the 8 TB/s peak.  One JSON line each.

    python scripts/ax_variants.py [--rows 10000000] [--band 131072] [--reps 20] [--patterns 1,2,0] [--modes plain,binned]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from liblcg_amd import _lib, api

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--band", type=int, default=131072)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--patterns", default="1,2,0")
ap.add_argument("--modes", default="plain,binned")
ap.add_argument("--symmetric", type=int, default=1)
ap.add_argument("--dot", type=int, default=0, help="1: the product the solver loops run -- lcg_hip_spmv_dot, y.x carried where the kernel family can")
args = ap.parse_args()
lib = _lib.load()
NAMES = {0: "scrambled", 1: "constant_diagonals", 2: "row_random_band"}
n = args.rows
x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, x)
y = torch.empty_like(x); yref = torch.empty_like(x)
_res = (_lib.C.c_double * 2)()


def product(A, x, y):
    if args.dot:
        assert lib.lcg_hip_spmv_dot(A.h, x.data_ptr(), y.data_ptr(), x.data_ptr(), None) == 0
    else:
        A.spmv(x, y)


for pat in [int(p) for p in args.patterns.split(",")]:
    A = api.CsrMatrix.generate(n, 16, args.band if pat else 0, bool(args.symmetric), 1, 0.01, pattern=pat)
    nnz = A.nnz
    byts = 12 * nnz + 4 * (n + 1) + 16 * n
    for mode in args.modes.split(","):
        assert lib.lcg_hip_csr_set_binned(A.h, {"binned": 1, "auto": -1}.get(mode, 0)) == 0
        assert lib.lcg_hip_csr_set_tiled(A.h, {"tiled": 1, "auto": -1}.get(mode, 0)) == 0
        api.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        api.use_torch_stream()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record(); product(A, x, y); t1.record(); torch.cuda.synchronize()      # includes building packed columns / the plan
        first_ms = t0.elapsed_time(t1)
        for _ in range(3):
            product(A, x, y)
        e0.record()
        for _ in range(args.reps):
            product(A, x, y)
        e1.record(); torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / args.reps
        if mode == "plain":
            yref.copy_(y)
        err = float(((y - yref).abs().max() / yref.abs().max()).item())
        print(json.dumps({"pattern": NAMES[pat], "rows": n, "nnz": nnz, "band": args.band if pat else 0, "mode": mode,
                          "kernel": lib.lcg_hip_csr_last_kernel(A.h).decode(), "ax_us": round(us, 1),
                          "algorithmic_GBs": round(byts / us / 1e3, 1), "frac_of_8TBs": round(byts / us / 1e3 / 8000, 4),
                          "first_call_ms": round(first_ms, 1), "streamed_bytes_model": lib.lcg_hip_csr_last_traffic_model(A.h),
                          "max_rel_diff_vs_plain": err}), flush=True)
    A.destroy()
