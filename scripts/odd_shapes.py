#!/usr/bin/env python3
"""A.x on shapes no benchmark has (looking for cliffs, not for records): rows of power-law lengths, dense diagonal blocks, very long
uniform rows, tridiagonal, diagonal only, a dense column.   python scripts/odd_shapes.py"""
import sys, time; sys.path.insert(0, ".")
import numpy as np, scipy.sparse as sp, torch
from liblcg_amd import _lib, api
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
rng = np.random.default_rng(5)


def run(name, M):
    M = M.tocsr(); M.sort_indices()
    n = M.shape[0]
    A = api.CsrMatrix.from_csr(M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64))
    xh = rng.standard_normal(n); x = torch.from_numpy(xh).cuda(); y = torch.empty_like(x)
    t0 = time.perf_counter(); A.spmv(x, y); api.synchronize(); first = time.perf_counter() - t0
    err = float(np.max(np.abs(y.cpu().numpy() - M @ xh) / np.maximum(abs(M) @ np.abs(xh), 1e-300)))
    A.spmv(x, y); api.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): A.spmv(x, y)
    api.synchronize(); t = (time.perf_counter() - t0) / 10
    byts = 12 * M.nnz + 20 * n
    print(f"{name:34s} rows {n:9d} entries {M.nnz:10d} longest row {int(np.diff(M.indptr).max()):7d}: {t * 1e6:8.1f} us = {byts / t / 8e12:.3f} of the peak, "
          f"first call {first * 1e3:6.1f} ms, err {err:.1e}, {lib.lcg_hip_csr_last_kernel(A.h).decode()[:70]}", flush=True)
    A.destroy()


n = 1_000_000
lens = np.minimum((rng.pareto(1.3, n) * 4 + 1).astype(np.int64), 5000)
rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum(lens)
ci = (np.repeat(np.arange(n), lens) + rng.integers(-20000, 20000, rp[-1])) % n
run("power-law row lengths", sp.csr_matrix((rng.standard_normal(rp[-1]), ci, rp), shape=(n, n)))
nb = 4000; bs = 128
run("dense 128 x 128 diagonal blocks", sp.block_diag([sp.csr_matrix(rng.standard_normal((bs, bs))) for _ in range(200)] * (nb // 200), format="csr"))
n = 200_000
run("200 diagonals", sp.diags([rng.standard_normal(n - o) for o in range(0, 2000, 10)], list(range(0, 2000, 10)), shape=(n, n)))
n = 10_000_000
run("tridiagonal", sp.diags([np.ones(n - 1), 2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1], shape=(n, n)))
run("diagonal only", sp.diags([np.arange(1.0, n + 1)], [0], shape=(n, n)))
n = 2_000_000
run("diagonal + one dense column", sp.diags([np.ones(n)], [0], shape=(n, n), format="csr") + sp.csr_matrix((np.ones(n), (np.arange(n), np.zeros(n, np.int64))), shape=(n, n)))
