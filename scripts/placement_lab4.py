#!/usr/bin/env python3
"""placement_lab 4: 40 separately allocated y vectors (sizes vary a little so that the allocator cannot hand a freed block back), the
product's time with each and the address -- is there a pattern?  Then the slow / fast ones again (is it stable?)."""
import sys, time; sys.path.insert(0, ".")
import torch
from liblcg_amd import _lib, api
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
n = 10_000_000
y0 = torch.empty(n, dtype=torch.float64, device="cuda")          # allocated BEFORE the matrix
A = api.CsrMatrix.generate(n, 16, 131072, True, 1, 0.01, pattern=api.GEN_DIAGONALS)
x = torch.rand(n, dtype=torch.float64, device="cuda")


def t(yy, reps=6):
    A.spmv(x, yy); api.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): A.spmv(x, yy)
    api.synchronize(); return (time.perf_counter() - t0) / reps * 1e6


print(f"y allocated before the matrix: {t(y0):.0f} us at {y0.data_ptr():#x}")
ys = []
for i in range(40):
    ys.append(torch.empty(n + i * 300_000, dtype=torch.float64, device="cuda"))
res = [(t(y[:n]), y.data_ptr()) for y in ys]
for i, (tt, p) in enumerate(res):
    print(f"{i:2d} {tt:5.0f} us  {p:#x}  (addr >> 21) % 64 = {(p >> 21) % 64:2d}  addr / 1 GiB = {p / (1 << 30):.3f}")
print("again:", " ".join(f"{t(y[:n]):.0f}" for y in ys[:12]))
print(f"y allocated before the matrix, again: {t(y0):.0f} us")
