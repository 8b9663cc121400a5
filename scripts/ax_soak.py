#!/usr/bin/env python3
"""Soak of the A.x kernel choice on ONE GPU (not a pytest test): random matrices from a menu of structures -- stencils of random
shape / reach / unknowns per point, constant diagonals, per-row random bands, scattered columns, and row-wise mixes of two of
them -- each multiplied (a) by whatever the library chooses with the big-matrix families forced on where eligible (packed
columns, row ranges) and in pure automatic mode, and (b) by the lanes-per-row kernel, which shares no code with (a); the two
must agree row by row within 1e-12 |A||x|.

  python scripts/ax_soak.py [cases=40] [seed=1]
"""
import itertools
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from liblcg_amd import _lib, api

lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def stencil(dims, reach, faces, dof):
    n0 = int(np.prod(dims)); idx = np.arange(n0).reshape(dims)
    rows, cols = [], []
    for d in itertools.product(*([range(-reach, reach + 1)] * len(dims))):
        if faces and sum(1 for k in d if k) > 1:
            continue
        rows.append(idx[tuple(slice(max(0, -k), m - max(0, k)) for k, m in zip(d, dims))].ravel())
        cols.append(idx[tuple(slice(max(0, k), m - max(0, -k)) for k, m in zip(d, dims))].ravel())
    r = np.concatenate(rows); c = np.concatenate(cols)
    if dof > 1:
        a = np.arange(dof)
        r = (r[:, None, None] * dof + a[None, :, None] + 0 * a[None, None, :]).ravel()
        c = (c.repeat(dof * dof).reshape(-1, dof, dof) * dof + a[None, None, :]).ravel()
    return n0 * dof, r, c


def piece(kind, h, n, r0):
    """rows r0 .. r0 + h of an n-column matrix, as (rows, cols)"""
    rr = np.arange(r0, r0 + h)
    if kind == "diag":
        offs = np.unique(np.concatenate([[0], rng.integers(-min(3000, n - 1), min(3000, n - 1), int(rng.integers(3, 33)))]))
        r = np.repeat(rr, len(offs)); c = (rr[:, None] + offs[None, :]).ravel()
    elif kind == "band":
        W = int(rng.integers(500, 200000)); k = int(rng.integers(4, 34))
        r = np.repeat(rr, k); c = (rr[:, None] + rng.integers(-W, W + 1, (h, k))).ravel()
    else:
        k = int(rng.integers(2, 34))
        r = np.repeat(rr, k); c = rng.integers(0, n, h * k)
    ok = (c >= 0) & (c < n)
    return r[ok], c[ok]


def to_csr(n, r, c):
    key = np.unique(r.astype(np.int64) * n + c)
    r, c = key // n, key % n
    rp = np.zeros(n + 1, np.int64); np.add.at(rp, r + 1, 1)
    return np.cumsum(rp).astype(np.int32), c.astype(np.int32)


worst = 0.0
t0 = time.time()
for case in range(cases):
    what = rng.choice(["stencil", "mix", "diag", "band", "scat"])
    if what == "stencil":
        nd = int(rng.integers(2, 4)); dof = int(rng.choice([1, 1, 2, 3])); reach = int(rng.choice([1, 1, 2])) if nd == 2 else 1
        faces = bool(rng.integers(0, 2))
        dims = tuple(int(rng.integers(8, 90 if nd == 3 else 900)) for _ in range(nd))
        while np.prod(dims) * dof > 1_500_000:
            dims = tuple(max(4, d // 2) for d in dims)
        n, r, c = stencil(dims, reach, faces, dof)
        desc = f"stencil {dims} reach {reach} faces {faces} dof {dof}"
    elif what == "mix":
        n = int(rng.integers(200_000, 1_200_000)); cut = int(rng.integers(n // 10, 9 * n // 10)) // 64 * 64
        k1, k2 = rng.choice(["diag", "band", "scat"], 2, replace=False)
        r1, c1 = piece(k1, cut, n, 0); r2, c2 = piece(k2, n - cut, n, cut)
        r = np.concatenate([r1, r2]); c = np.concatenate([c1, c2])
        desc = f"mix {k1} | {k2} at {cut} of {n}"
    else:
        n = int(rng.integers(100_000, 1_500_000))
        r, c = piece(what, n, n, 0)
        desc = f"{what} {n}"
    rp, ci = to_csr(n, r, c)
    val = rng.standard_normal(len(ci))
    A = api.CsrMatrix.from_csr(rp, ci, val); B = api.CsrMatrix.from_csr(rp, ci, np.abs(val))
    x = torch.from_numpy(rng.standard_normal(n)).cuda()
    yref = torch.empty_like(x); bound = torch.empty_like(x); y = torch.empty_like(x)
    A.set_kernel(4); A.spmv(x, yref); B.set_kernel(4); B.spmv(x.abs(), bound); api.synchronize()
    A.set_kernel(0)
    names = []
    for packed, ranges in ((-1, -1), (1, 1), (1, 0)):
        assert lib.lcg_hip_csr_set_packed(A.h, packed) == 0 and lib.lcg_hip_csr_set_ranges(A.h, ranges) == 0
        y.fill_(float("nan"))
        A.spmv(x, y); api.synchronize()
        err = float(((y - yref).abs() / (bound + 1e-300)).max().item())
        names.append(lib.lcg_hip_csr_last_kernel(A.h).decode().split(" (")[0][:60])
        worst = max(worst, err)
        if not err <= 1e-12:
            print(f"FAIL case {case}: {desc}: packed {packed} ranges {ranges}: {lib.lcg_hip_csr_last_kernel(A.h).decode()}: err {err:.3e}")
            sys.exit(1)
    print(f"case {case:3d} {desc[:70]:70s} nnz {len(ci):9d}  {' / '.join(names)}", flush=True)
    A.destroy(); B.destroy()
print(f"ax soak: {cases} cases ok, worst error {worst:.2e} of |A||x|, {time.time() - t0:.0f} s")
