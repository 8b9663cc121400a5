#!/usr/bin/env python3
"""Generate tests/golden/ref_*.npz from the REAL liblcg native back-end.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Each record holds the inputs' identity (fixture name, solver, parameters) and the
reference's outputs (return code, iteration count seen by the progress callback,
last monitored residual, solution vector).  For the complex solvers whose shadow
residual the reference draws from srand(time(0)) (clcg.cpp:399-403,556-560,721-725)
the wall-clock second that seeded the draw is recorded too, so the restatement can
replay the same vector (oracle.orc_clcg_vecrnd) and must then match bit for bit.

The fixtures case_* in this directory are verbatim copies of the reference's bundled
data files (data/README:1-10): data, not code.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from liblcg_amd.coo_io import coo_to_csr_host, read_coo_system  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

REAL_CASES = [  # (tag, solver_id, jacobi, epsilon, abs_diff)
    ("cg_e6", po.LCG_CG, 0, 1e-6, 0), ("pcg_e6", po.LCG_PCG, 1, 1e-6, 0),
    ("cgs_e6", po.LCG_CGS, 0, 1e-6, 0), ("bicgstab_e6", po.LCG_BICGSTAB, 0, 1e-6, 0),
    ("cg_e10", po.LCG_CG, 0, 1e-10, 1), ("pcg_e10", po.LCG_PCG, 1, 1e-10, 1),
    ("cgs_e10", po.LCG_CGS, 0, 1e-10, 1), ("bicgstab_e10", po.LCG_BICGSTAB, 0, 1e-10, 1),
    ("cg_e20", po.LCG_CG, 0, 1e-20, 1), ("pcg_e12", po.LCG_PCG, 1, 1e-12, 1),
    ("cgs_e12", po.LCG_CGS, 0, 1e-12, 1), ("bicgstab_e12", po.LCG_BICGSTAB, 0, 1e-12, 1),
    ("cg_e12", po.LCG_CG, 0, 1e-12, 1),
    ("cg_max25", po.LCG_CG, 0, 1e-12, 1),  # max_iterations = 25 -> LCG_REACHED_MAX_ITERATIONS
    ("bicgstab2_e10", 4, 0, 1e-10, 1), ("bicgstab2_e6", 4, 0, 1e-6, 0), ("bicgstab2_max31", 4, 0, 1e-10, 1),
]
BOX_CASES = [  # (tag, solver_id, epsilon, abs_diff, max_iterations); box = [-5, 8] on case_10K_A
    ("pg_40", 5, 1e-10, 1, 40), ("spg_40", 6, 1e-10, 1, 40), ("pg_150", 5, 1e-6, 0, 150), ("spg_60", 6, 1e-6, 0, 60),
]
CPLX_CASES = [  # (tag, fixture, solver_id, epsilon, abs_diff)
    ("bicg_1K", "1K", po.CLCG_BICG, 1e-10, 1), ("bicg_10K", "10K", po.CLCG_BICG, 1e-10, 1),
    ("bicgsym_1K", "1K", po.CLCG_BICG_SYM, 1e-10, 1), ("cgs_1K", "1K", po.CLCG_CGS, 1e-10, 1),
    ("tfqmr_1K", "1K", po.CLCG_TFQMR, 1e-10, 1), ("bicgstab_1K", "1K", po.CLCG_BICGSTAB, 1e-10, 1),
    ("bicgsym_10K", "10K", po.CLCG_BICG_SYM, 1e-10, 1), ("cgs_10K", "10K", po.CLCG_CGS, 1e-10, 1),
    ("tfqmr_10K", "10K", po.CLCG_TFQMR, 1e-10, 1),
]


def main():
    ref = po.Oracle("reference")
    out = {}
    n, row, col, val, b = read_coo_system(os.path.join(HERE, "case_10K_A"))
    rp, ci, v = coo_to_csr_host(n, row, col, val)
    for tag, sid, jac, eps, ad in REAL_CASES:
        para = po.default_para(epsilon=eps, abs_diff=ad)
        if tag == "cg_max25":
            para.max_iterations = 25
        if tag == "bicgstab2_max31":
            para.max_iterations = 31
        r = ref.solve(sid, rp, ci, v, b, para=para, jacobi=bool(jac))
        out[f"real/{tag}/x"] = r["x"]
        out[f"real/{tag}/meta"] = np.array([r["ret"], r["iters"], sid, jac, ad, para.max_iterations], np.int64)
        out[f"real/{tag}/fl"] = np.array([eps, r["residual"]])
        print(tag, r["ret"], r["iters"], r["residual"])
    low, hig = np.full(n, -5.0), np.full(n, 8.0)
    for tag, sid, eps, ad, maxit in BOX_CASES:
        para = po.default_para(epsilon=eps, abs_diff=ad, max_iterations=maxit)
        r = ref.solve_box(sid, rp, ci, v, b, low, hig, para=para)
        out[f"box/{tag}/x"] = r["x"]
        out[f"box/{tag}/meta"] = np.array([r["ret"], r["iters"], sid, ad, maxit, r["n_ax"]], np.int64)
        out[f"box/{tag}/fl"] = np.array([eps, r["residual"]])
        print(tag, r["ret"], r["iters"], r["residual"], r["n_ax"])
    for tag, fx, sid, eps, ad in CPLX_CASES:
        n, row, col, val, b = read_coo_system(os.path.join(HERE, f"case_{fx}_cA"), True)
        rp, ci, v = coo_to_csr_host(n, row, col, val)
        para = po.default_cpara(epsilon=eps, abs_diff=ad)
        if sid == po.CLCG_BICGSTAB:
            para.max_iterations = 300   # does not converge on these systems
        while True:
            r = ref.csolve(sid, rp, ci, v, b, para=para)
            if r["seed_before"] == r["seed_after"]:
                break
        out[f"cplx/{tag}/x"] = r["x"]
        out[f"cplx/{tag}/meta"] = np.array([r["ret"], r["iters"], sid, ad, para.max_iterations,
                                            r["seed_before"]], np.int64)
        out[f"cplx/{tag}/fl"] = np.array([eps, r["residual"]])
        print(tag, r["ret"], r["iters"], r["residual"], r["seed_before"])
    np.savez_compressed(os.path.join(HERE, "ref_goldens.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_goldens.npz"))


if __name__ == "__main__":
    main()
