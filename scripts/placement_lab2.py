#!/usr/bin/env python3
"""Continuation of placement_lab.py: ONE matrix; x / y pairs (a) allocated one by one (16 pairs, each its own allocation) and (b) cut out
of one 4 GB allocation at 2 MB-aligned offsets.  How many pairs are slow in each family?"""
import sys, time; sys.path.insert(0, ".")
import torch
from liblcg_amd import _lib, api
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
n = 10_000_000
A = api.CsrMatrix.generate(n, 16, 131072, True, 1, 0.01, pattern=api.GEN_DIAGONALS)


def t(xx, yy, reps=10):
    A.spmv(xx, yy); A.spmv(xx, yy); api.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): A.spmv(xx, yy)
    api.synchronize(); return (time.perf_counter() - t0) / reps * 1e6


keep = []
own = []
for i in range(16):
    keep.append(torch.empty((i * 53 + 7) * 1031, dtype=torch.float64, device="cuda"))
    xi = torch.rand(n, dtype=torch.float64, device="cuda"); yi = torch.empty_like(xi); keep += [xi, yi]
    own.append(t(xi, yi))
print("own allocations :", " ".join(f"{v:5.0f}" for v in own), flush=True)
slab = torch.empty(1 << 29, dtype=torch.float64, device="cuda")        # 4 GB
step = ((n * 8 + (2 << 20) - 1) // (2 << 20) * (2 << 20)) // 8          # 2 MB-aligned stride in doubles
base = (-(slab.data_ptr() % (2 << 20)) % (2 << 20)) // 8
cut = []
for i in range(16):
    xi = slab[base + (2 * i) * step: base + (2 * i) * step + n]; yi = slab[base + (2 * i + 1) * step: base + (2 * i + 1) * step + n]
    xi.uniform_()
    cut.append(t(xi, yi))
print("cut from one 4 GB:", " ".join(f"{v:5.0f}" for v in cut), flush=True)
# x from the slab, y own and the other way round (which of the two vectors is it?)
xi = slab[base: base + n]; yo = keep[2]; xo = keep[1]
print("x slab + y own:", f"{t(xi, yo):5.0f}", " x own + y slab:", f"{t(xo, slab[base + step: base + step + n]):5.0f}")
