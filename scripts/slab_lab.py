#!/usr/bin/env python3
"""What would an Infinity-Cache slab walk of the binned product cost?  

The walk would run the two passes slab of rows by slab of rows, so that a slab's expanded `xg` (16 B per entry of round trip) is still
in the 256 MB Infinity Cache when pass 2 reads it, at the price of loading every 64 KB slice of x once per SLAB.  One slab of that
walk is exactly the binned product of a RECTANGULAR matrix -- the slab's rows x all 10M columns -- so the existing kernels measure it
without a line of new device code: for slab heights of 1/110 ... 1/7 of the rows, the scattered product of the slab from cold caches
(a 1 GiB fill in front of every product: the matrix streams of a slab come from HBM in the walk too, x is re-warmed as it would be)
x the number of slabs = the walk's time, against 1/slabs of the whole product as it runs today.

    python3 scripts/slab_lab.py [--rows 10000000]
"""
import argparse
import json
import sys
import time

sys.path.insert(0, ".")
import torch
from liblcg_amd import _lib, api

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--per-row", type=int, default=33)
ap.add_argument("--slabs", default="110,55,28,14,7")
args = ap.parse_args()
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
api.use_torch_stream()
n = args.rows
dev = "cuda"
torch.manual_seed(1)
x = torch.rand(n, dtype=torch.float64, device=dev)
flush = torch.empty(1 << 27, dtype=torch.float64, device=dev)       # 1 GiB


def ev_time(fn, reps, pre=None):
    ts = []
    for _ in range(reps):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def scattered(rows):
    """rows x n CSR, --per-row columns per row drawn uniformly (sorted inside the row), values in (0, 1)."""
    k = args.per_row
    col = torch.randint(0, n, (rows, k), device=dev, dtype=torch.int32)
    col, _ = torch.sort(col, dim=1)
    rp = torch.arange(0, rows * k + 1, k, device=dev, dtype=torch.int32)
    val = torch.rand(rows * k, dtype=torch.float64, device=dev)
    return api.CsrMatrix.from_csr(rp, col.reshape(-1).contiguous(), val, n_cols=n)


out = []
# the whole product as it runs today (10M x 10M, binned)
A = scattered(n)
lib.lcg_hip_csr_set_binned(A.h, 1)
y = torch.empty(n, dtype=torch.float64, device=dev)
A.spmv(x, y); api.synchronize()
whole = ev_time(lambda: A.spmv(x, y), 7)
whole_cold = ev_time(lambda: A.spmv(x, y), 5, pre=lambda: flush.fill_(1.0))
kern = lib.lcg_hip_csr_last_kernel(A.h).decode()
nnz = A.nnz
print(json.dumps({"whole_product_us": whole, "whole_product_cold_us": whole_cold, "nnz": nnz, "kernel": kern}), flush=True)
A.destroy(); del A, y


def warm_x():       # what the previous slab leaves behind in the walk: x was just read through
    x.sum()


for ns in [int(s) for s in args.slabs.split(",")]:
    rows = (n + ns - 1) // ns
    S = scattered(rows)
    lib.lcg_hip_csr_set_binned(S.h, 1)
    ys = torch.empty(rows, dtype=torch.float64, device=dev)
    S.spmv(x, ys); api.synchronize()
    k = lib.lcg_hip_csr_last_kernel(S.h).decode()
    back_to_back = ev_time(lambda: S.spmv(x, ys), 9)
    cold = ev_time(lambda: S.spmv(x, ys), 7, pre=lambda: flush.fill_(1.0))
    cold_xwarm = ev_time(lambda: S.spmv(x, ys), 7, pre=lambda: (flush.fill_(1.0), warm_x()))
    ent = {"slabs": ns, "slab_rows": rows, "slab_entries": S.nnz, "xg_MB": S.nnz * 8 / 1e6, "kernel": k.split(" (")[0],
           "slab_us_back_to_back": back_to_back, "slab_us_cold": cold, "slab_us_cold_x_warm": cold_xwarm,
           "walk_us_cold": cold * ns, "walk_us_cold_x_warm": cold_xwarm * ns, "walk_us_back_to_back": back_to_back * ns,
           "whole_product_us": whole, "walk_over_whole_cold_x_warm": cold_xwarm * ns / whole}
    print(json.dumps(ent), flush=True)
    out.append(ent)
    S.destroy(); del S, ys
