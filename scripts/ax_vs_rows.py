"""A.x wall time per call against the number of rows (same band, packed columns): is there a fixed cost per launch that
would hurt the 8-way shards?  (No: 89.8 us at 1.25M rows against 698.3 / 8 = 87.3 us at 10M; 625K rows, whose matrix fits
the Infinity Cache, run faster per entry.)

  python scripts/ax_vs_rows.py
"""
import sys, time
sys.path.insert(0, ".")
import torch
from liblcg_amd import _lib, api
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
for n in (312500, 625000, 1250000, 2500000, 5000000, 10000000):
    A = api.CsrMatrix.generate(n, 16, 131072, True, 1, 0.01)
    lib.lcg_hip_csr_set_packed(A.h, 1)
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for _ in range(10): A.spmv(x, y)
    api.synchronize()
    reps = 300 if n < 5_000_000 else 100
    t0 = time.perf_counter()
    for _ in range(reps): A.spmv(x, y)
    api.synchronize()
    t = (time.perf_counter() - t0) / reps * 1e6
    print(f"n={n:9d} nnz={A.nnz:10d} {t:8.1f} us  {t / (A.nnz / 1e6):.3f} us per M entries")
    A.destroy()
