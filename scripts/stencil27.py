#!/usr/bin/env python3
"""A.x and CG on a REAL stencil instead of the synthetic family: the 27-point stencil of an nx x ny x nz grid (default 200^3 =
8M rows, 2.1e8 entries), built as COO on the device with torch, handed to lcg_hip_csr_from_coo through the host, multiplied by
whatever kernel the automatic choice takes.  Prints the kernel, time per product, algorithmic GB/s and the CG rate.

  python scripts/stencil27.py [nx ny nz [27|7]]
"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from liblcg_amd import _lib, api

lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
nx, ny, nz = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (200, 200, 200)
points = int(sys.argv[4]) if len(sys.argv) >= 5 else 27        # 27 (full cube) or 7 (faces only)
dof = int(sys.argv[5]) if len(sys.argv) >= 6 else 1            # unknowns per grid point (full dof x dof coupling blocks)
n = nx * ny * nz
dev = "cuda"
idx = torch.arange(n, device=dev, dtype=torch.int64).reshape(nz, ny, nx)
rows, cols, vals = [], [], []
for dz in (-1, 0, 1):
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if points == 7 and abs(dx) + abs(dy) + abs(dz) > 1:
                continue
            src = idx[max(0, -dz):nz - max(0, dz), max(0, -dy):ny - max(0, dy), max(0, -dx):nx - max(0, dx)].reshape(-1)
            dst = idx[max(0, dz):nz - max(0, -dz), max(0, dy):ny - max(0, -dy), max(0, dx):nx - max(0, -dx)].reshape(-1)
            rows.append(src); cols.append(dst)
            vals.append(torch.full((src.numel(),), float(points) if (dx, dy, dz) == (0, 0, 0) else -1.0, device=dev, dtype=torch.float64))
r = torch.cat(rows); c = torch.cat(cols); v = torch.cat(vals)
if dof > 1:         # every node coupling becomes a dof x dof block (values scaled so that the diagonal still dominates)
    a = torch.arange(dof, device=dev, dtype=torch.int64)
    r = (r[:, None, None] * dof + a[None, :, None]).expand(-1, dof, dof).reshape(-1)
    c = (c[:, None, None] * dof + a[None, None, :]).expand(-1, dof, dof).reshape(-1)
    v = v[:, None, None].expand(-1, dof, dof).reshape(-1).clone()
    off = (r % dof) != (c % dof)
    v[off] = v[off] / (2.0 * dof)
    n = n * dof
order = torch.argsort(r * n + c)
r, c, v = r[order], c[order], v[order]
rp = torch.zeros(n + 1, dtype=torch.int64, device=dev); rp[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
A = api.CsrMatrix.from_csr(rp.to(torch.int32), c.to(torch.int32), v)
nnz = A.nnz
import os
if os.environ.get("KERNEL"):
    A.set_kernel(int(os.environ["KERNEL"]))
if os.environ.get("PACKED"):
    assert lib.lcg_hip_csr_set_packed(A.h, int(os.environ["PACKED"])) == 0
del r, c, v, rows, cols, vals, order
x = torch.rand(n, dtype=torch.float64, device=dev); y = torch.empty_like(x)
t0 = time.perf_counter(); A.spmv(x, y); api.synchronize(); first = time.perf_counter() - t0
for _ in range(3):
    A.spmv(x, y)
api.synchronize()
t0 = time.perf_counter()
reps = 30
for _ in range(reps):
    A.spmv(x, y)
api.synchronize()
t = (time.perf_counter() - t0) / reps
byts = 12 * nnz + 4 * (n + 1) + 16 * n
runs = lib.lcg_hip_csr_packed_runs(A.h, None)
print(f"{points}-point stencil {nx}x{ny}x{nz} x {dof} dof: rows {n}, entries {nnz}; kernel: {lib.lcg_hip_csr_last_kernel(A.h).decode()} ({runs} run blocks of {(n + 63) // 64})")
print(f"A.x {t * 1e6:.1f} us = {byts / t / 1e9:.0f} GB/s algorithmic = {byts / t / 8e12:.3f} of 8 TB/s; must move {lib.lcg_hip_csr_last_traffic_model(A.h) / t / 8e12:.3f}; first call {first * 1e3:.1f} ms")
xt = torch.rand(n, dtype=torch.float64, device=dev); b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
m = torch.zeros_like(xt)
t0 = time.perf_counter()
info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=1e-300, max_iterations=200), A, api.LCG_CG)
api.synchronize()
dt = time.perf_counter() - t0
print(f"CG: {info.iterations} iterations in {dt * 1e3:.1f} ms = {info.iterations / dt:.0f} it/s; |x - x_true| / |x_true| = {((m - xt).norm() / xt.norm()).item():.2e}")
