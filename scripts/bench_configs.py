#!/usr/bin/env python3
"""Secondary measurements: every BASELINE.json config that fits one GPU, one JSON line each.

  config 1  case_10K_A, CG/PCG/CGS/BiCGStab to eps=1e-10 (latency-bound: it/s only)
  config 2  5-point Laplacian 1000x1000, PCG + Jacobi, 500 iterations
  config 3  10M-row banded SPD, CG (the headline; also the scrambled variant and other W)
  config 5  10M-row non-symmetric BiCGStab; case_10K_cA TFQMR
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from liblcg_amd import _lib, api
from liblcg_amd.coo_io import read_coo_system, read_solution

lib = _lib.load()
G = os.path.join(ROOT, "tests", "golden")


def spmv_bytes(n, nnz, cplx=False):
    return (20 if cplx else 12) * nnz + 4 * (n + 1) + (32 if cplx else 16) * n


def timed(fn, reps=1):
    torch.cuda.synchronize(); api.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    api.synchronize()
    return (time.perf_counter() - t0) / reps, r


def emit(**kw):
    print(json.dumps(kw), flush=True)


def fixed_iter_run(name, A, n, solver, iters, words, ax_per_it, jacobi=False):
    xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    m = torch.zeros_like(xt)
    p = api.lcg_default_parameters(epsilon=1e-300, max_iterations=iters)

    def go():
        m.zero_()
        if jacobi:
            return api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, p, A)
        return api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, p, A, solver)
    go()
    lib.lcg_hip_set_profiling(1)
    dt, info = timed(go)
    ax_us = lib.lcg_hip_last_ax_mean_us(); lib.lcg_hip_set_profiling(0)
    nnz = A.nnz
    byts = ax_per_it * spmv_bytes(n, nnz) + 8 * words * n
    emit(config=name, rows=n, nnz=nnz, iterations=info.iterations, ret=info.ret, it_per_s=iters / dt, ms_per_it=1e3 * dt / iters,
         algorithmic_GBs=byts / (dt / iters) / 1e9, ax_mean_us=ax_us, ax_GBs=spmv_bytes(n, nnz) / (ax_us * 1e-6) / 1e9 if ax_us else None,
         rel_err=((m - xt).norm() / xt.norm()).item(), ax_kernel=lib.lcg_hip_csr_last_kernel(A.h).decode())


def main():
    what = sys.argv[1:] or ["c1", "c2", "c3", "c5"]
    if "c1" in what:
        n, row, col, val, b = read_coo_system(os.path.join(G, "case_10K_A")); xs = read_solution(os.path.join(G, "case_10K_B"))
        A = api.CsrMatrix.from_coo(n, row, col, val); A.build_jacobi()
        bd = torch.from_numpy(b).cuda()
        for name, sid in (("CG", 0), ("CG one-reduction schedule", 0), ("PCG", 1), ("CGS", 2), ("BICGSTAB", 3)):
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            p = api.lcg_default_parameters(epsilon=1e-10, abs_diff=1)
            api.set_cg_schedule(api.CG_ONE_REDUCTION if "one-reduction" in name else api.CG_AUTO)

            def go():
                m.zero_()
                if sid == 1:
                    return api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, bd, n, p, A)
                return api.lcg_solver("lcg_hip_csr_ax", None, m, bd, n, p, A, sid)
            go()
            dt, info = timed(go, 5)
            emit(config=f"1: case_10K_A {name} eps=1e-10", iterations=info.iterations, ret=info.ret, solve_ms=1e3 * dt,
                 us_per_it=1e6 * dt / info.iterations, err_vs_case_10K_B=float(np.linalg.norm(m.cpu().numpy() - xs)))
    if "c2" in what:
        A = api.CsrMatrix.laplace2d(1000, 1000); A.build_jacobi()
        fixed_iter_run("2: Laplace2D 1000x1000 PCG+Jacobi 500 its", A, 1_000_000, api.LCG_PCG, 500, 18, 1, jacobi=True)
        fixed_iter_run("2b: Laplace2D 1000x1000 CG 500 its", A, 1_000_000, api.LCG_CG, 500, 13, 1)
    if "c3" in what:
        n = 10_000_000
        for pattern, band, label in ((1, 131072, "33 constant diagonals W=131072"), (1, 2048, "33 constant diagonals W=2048"),
                                     (2, 131072, "row-random band W=131072"), (2, 16384, "row-random band W=16384"),
                                     (2, 1048576, "row-random band W=1048576"), (0, 0, "scrambled")):
            A = api.CsrMatrix.generate(n, 16, band, True, 1, 0.01, pattern=pattern)
            fixed_iter_run(f"3: 10M {label} CG 50 its", A, n, api.LCG_CG, 50, 13, 1)
            A.destroy()
    if "c5" in what:
        n = 10_000_000
        A = api.CsrMatrix.generate(n, 16, 131072, False, 1, 0.01)
        fixed_iter_run("5: 10M non-symmetric BiCGStab 100 its", A, n, api.LCG_BICGSTAB, 100, 22, 2)
        fixed_iter_run("5b: 10M non-symmetric CGS 100 its", A, n, api.LCG_CGS, 100, 21, 2)
        A.destroy()
        n, row, col, val, b = read_coo_system(os.path.join(G, "case_10K_cA"), True); xs = read_solution(os.path.join(G, "case_10K_cB"), True)
        A = api.CsrMatrix.from_coo(n, row, col, val)
        bd = torch.from_numpy(b).cuda()
        for name, sid in (("BICG_SYM", 1), ("CGS", 2), ("TFQMR", 4)):
            m = torch.zeros(n, dtype=torch.complex128, device="cuda")
            p = api.clcg_default_parameters(epsilon=1e-10, abs_diff=1)

            def go():
                m.zero_()
                return api.clcg_solver("clcg_hip_csr_ax", None, m, bd, n, p, A, sid)
            go()
            dt, info = timed(go, 3)
            emit(config=f"5: case_10K_cA {name} eps=1e-10", iterations=info.iterations, ret=info.ret, solve_ms=1e3 * dt,
                 us_per_it=1e6 * dt / info.iterations, err_vs_case_10K_cB=float(np.linalg.norm(m.cpu().numpy() - xs)))


if __name__ == "__main__":
    main()
