"""Worker of tests/test_gpu_kernels.py::test_lds_window_does_not_change_a_bit: multiplies a fixed menu of matrices by the row-block
kernels and prints, per case, the kernel that ran and the SHA-256 of y.  The parent runs it under different LCG_HIP_PACKED_WINDOW
settings (the window is read once per process) and compares the lines."""
import hashlib
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from liblcg_amd import _lib, api

lib = _lib.load(); assert lib.lcg_hip_init(0) == 0


def stencil(dims, faces, dof):
    n0 = int(np.prod(dims)); idx = np.arange(n0).reshape(dims)
    rows, cols = [], []
    for d in itertools.product(*([range(-1, 2)] * len(dims))):
        if faces and sum(1 for k in d if k) > 1:
            continue
        src = idx[tuple(slice(max(0, -k), m - max(0, k)) for k, m in zip(d, dims))].ravel()
        dst = idx[tuple(slice(max(0, k), m - max(0, -k)) for k, m in zip(d, dims))].ravel()
        for a in range(dof):
            for b in range(dof):
                rows.append(src * dof + a); cols.append(dst * dof + b)
    r = np.concatenate(rows); c = np.concatenate(cols)
    order = np.lexsort((c, r)); r, c = r[order], c[order]
    n = n0 * dof
    rp = np.zeros(n + 1, np.int64); np.add.at(rp, r + 1, 1)
    return n, np.cumsum(rp).astype(np.int32), c.astype(np.int32)


def run(name, A, n, packed):
    if packed is not None:          # True: packed columns forced; False: switched off (k_spmv_lds1); None: the automatic choice
        assert lib.lcg_hip_csr_set_packed(A.h, 1 if packed else 0) == 0
    assert lib.lcg_hip_csr_set_tiled(A.h, 0) == 0 and lib.lcg_hip_csr_set_binned(A.h, 0) == 0
    x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 5, 0, n, x)
    u = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 6, 0, n, u)
    y = torch.empty_like(x); A.spmv(x, y); api.synchronize()
    k = lib.lcg_hip_csr_last_kernel(A.h).decode().split(" (")[0]
    import ctypes as C
    sums = (C.c_double * 2)(); y2 = torch.empty_like(x)
    assert lib.lcg_hip_spmv_dot(A.h, x.data_ptr(), y2.data_ptr(), u.data_ptr(), sums) == 0
    assert torch.equal(y, y2)
    print(name, k, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest(), repr(sums[0]), repr(sums[1]), flush=True)
    A.destroy()


rng = np.random.default_rng(3)
# 27-point stencil (1728 entries per 64-row block: the 1872 window), 7-point x 3 unknowns (1344: 1696), constant diagonals at 33 per
# row (2112: 2208; the full window is what LCG_HIP_PACKED_WINDOW=2240 runs them all with), and long rows (>= 4M entries) on the plain LDS-staged kernel and, by the automatic choice, in blocks of 32 rows with packed columns
for name, dims, faces, dof, packed in (("stencil27", (40, 41, 42), False, 1, True), ("stencil7x3", (30, 31, 32), True, 3, True),
                                       ("stencil27x2", (35, 36, 37), False, 2, False), ("stencil27x2p", (35, 36, 37), False, 2, None)):
    n, rp, ci = stencil(dims, faces, dof)
    v = np.random.default_rng(sum(dims) + dof).standard_normal(len(ci))       # (the same values for the same system under two names)
    A = api.CsrMatrix.from_csr(rp, ci, v)
    run(name, A, n, packed)
A = api.CsrMatrix.generate(200_000, 16, 3000, True, 4, 0.01, pattern=api.GEN_DIAGONALS)
run("diagonals33", A, 200_000, True)
A = api.CsrMatrix.generate(150_000, 16, 3000, True, 4, 0.01, pattern=api.GEN_ROW_RANDOM_BAND)
run("band", A, 150_000, True)
