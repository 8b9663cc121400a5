#!/usr/bin/env python3
"""A banded SPD system with three dense rows / columns (300,000 / 120,000 / 60,000 entries in rows of a 300,000-row system): A.x with the
dense rows cut out into ranges of their own (the automatic choice) and, with LCG_HIP_RANGES=0, left in the part.
  python scripts/arrow_lab.py ; LCG_HIP_RANGES=0 python scripts/arrow_lab.py"""
import sys, time; sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp, torch
from liblcg_amd import _lib, api
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
rng = np.random.default_rng(314); n = 300_000
offs = np.unique(np.concatenate([[0], rng.integers(1, 3000, 8)]))
B = sp.diags([rng.standard_normal(n - o) * 0.1 for o in offs], offs, shape=(n, n), format="coo")
rows = [B.row, B.col[B.row != B.col]]; cols = [B.col, B.row[B.row != B.col]]; vals = [B.data, B.data[B.row != B.col]]
for r, cnt in {n - 1: n - 1, n // 2 + 17: 120_000, 70_001: 60_000}.items():
    c = rng.choice(np.setdiff1d(np.arange(n), [r], assume_unique=True), cnt, replace=False) if cnt < n - 1 else np.delete(np.arange(n), r)
    v = rng.standard_normal(len(c)) * 1e-3
    rows += [np.full(len(c), r), c]; cols += [c, np.full(len(c), r)]; vals += [v, v]
M = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr(); M.sum_duplicates()
M = (M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + 1.0)).tocsr(); M.sort_indices()
A = api.CsrMatrix.from_csr(M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data)
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
A.spmv(x, y); api.synchronize()
t0 = time.perf_counter()
for _ in range(20): A.spmv(x, y)
api.synchronize()
print(f"{M.nnz} entries: A.x {(time.perf_counter() - t0) / 20 * 1e6:.0f} us; {lib.lcg_hip_csr_last_kernel(A.h).decode()[:300]}")
