"""The oracle (C restatement) against the reference's own outputs.

tests/golden/ref_goldens.npz was produced by the REAL liblcg native back-end
(tests/golden/make_golden.py).  The restatement must reproduce it bit for bit:
same return code, same iteration count, identical solution vector.  This is the
pin that lets the oracle stand in for the reference on the GPU box.
"""
import numpy as np
import pytest

from oracle import pyoracle as po

REAL_TAGS = ["cg_e6", "pcg_e6", "cgs_e6", "bicgstab_e6", "cg_e10", "pcg_e10", "cgs_e10",
             "bicgstab_e10", "cg_e20", "pcg_e12", "cgs_e12", "bicgstab_e12", "cg_e12", "cg_max25",
             "bicgstab2_e10", "bicgstab2_e6", "bicgstab2_max31"]
BOX_TAGS = ["pg_40", "spg_40", "pg_150", "spg_60"]
CPLX_TAGS = ["bicg_1K", "bicg_10K", "bicgsym_1K", "cgs_1K", "tfqmr_1K", "bicgstab_1K", "bicgsym_10K", "cgs_10K", "tfqmr_10K"]


@pytest.mark.parametrize("tag", REAL_TAGS)
def test_real_solver_bit_exact(tag, port, goldens, case10k):
    n, rp, ci, v, b, _ = case10k
    ret, iters, sid, jac, ad, maxit = goldens[f"real/{tag}/meta"]
    eps, resid = goldens[f"real/{tag}/fl"]
    para = po.default_para(epsilon=float(eps), abs_diff=int(ad), max_iterations=int(maxit))
    r = port.solve(int(sid), rp, ci, v, b, para=para, jacobi=bool(jac))
    assert r["ret"] == ret
    assert r["iters"] == iters
    assert r["residual"] == resid
    assert np.array_equal(r["x"], goldens[f"real/{tag}/x"])


@pytest.mark.parametrize("tag", BOX_TAGS)
def test_box_constrained_solver_bit_exact(tag, port, goldens, case10k):
    """lpg / lspg (lcg.cpp:1054-1446) on case_10K_A with the box [-5, 8] (1200 active bounds)."""
    n, rp, ci, v, b, _ = case10k
    ret, iters, sid, ad, maxit, n_ax = goldens[f"box/{tag}/meta"]
    eps, resid = goldens[f"box/{tag}/fl"]
    para = po.default_para(epsilon=float(eps), abs_diff=int(ad), max_iterations=int(maxit))
    r = port.solve_box(int(sid), rp, ci, v, b, np.full(n, -5.0), np.full(n, 8.0), para=para)
    assert (r["ret"], r["iters"], r["n_ax"], r["residual"]) == (ret, iters, n_ax, resid)
    assert np.array_equal(r["x"], goldens[f"box/{tag}/x"])


@pytest.mark.parametrize("tag", CPLX_TAGS)
def test_complex_solver_bit_exact(tag, port, goldens, case1kc, case10kc):
    n, rp, ci, v, b, _ = case1kc if tag.endswith("1K") else case10kc
    ret, iters, sid, ad, maxit, seed = goldens[f"cplx/{tag}/meta"]
    eps, resid = goldens[f"cplx/{tag}/fl"]
    para = po.default_cpara(epsilon=float(eps), abs_diff=int(ad), max_iterations=int(maxit))
    rbar0 = port.vecrnd(n, int(seed))      # replay of the reference's srand(time(0)) draw
    r = port.csolve(int(sid), rp, ci, v, b, para=para, rbar0=rbar0)
    assert r["ret"] == ret
    assert r["iters"] == iters
    assert r["residual"] == resid
    assert np.array_equal(r["x"], goldens[f"cplx/{tag}/x"])


def test_clpcg_restatement_against_known_answers(port, case1kc, case10kc):
    """orc_clpcg is the one oracle function WITHOUT a reference pin (clpcg exists only in liblcg's CUDA and
    Eigen back-ends, neither buildable here).  What can be checked: it solves the bundled complex systems
    to their known answers, and its monitored value is |r|/N recomputed independently."""
    for (n, rp, ci, v, b, xs), iters in ((case1kc, 377), (case10kc, 526)):
        r = port.csolve_pcg(rp, ci, v, b, para=po.default_cpara(epsilon=1e-10, abs_diff=1))
        assert r["ret"] == 0 and r["iters"] == iters
        assert np.linalg.norm(r["x"] - xs) <= 2e-7
        res = np.linalg.norm(port.csr_matvec(rp, ci, v, r["x"]) - b) / n
        assert res <= 1.05e-10


def test_clpbicg_restatement_against_known_answers(port, case1kc, case10kc):
    """orc_clpbicg (restated from clcg_eigen.cpp:685-802) has no reference pin either: the Eigen back-end is the only
    place it exists.  What can be checked: it solves the bundled complex systems -- its stop rule is the CPU loops'
    4th-power one, so eps = 1e-10 / abs_diff stops at |r|^2 / N <= 1e-10 -- ; its monitored value is |r|^2 / N of its
    answer recomputed independently; it takes two products per iteration (A.p and conj(A).ps) plus the set-up one."""
    for (n, rp, ci, v, b, xs), iters, dist in ((case1kc, 322, 1e-4), (case10kc, 453, 5e-4)):
        r = port.csolve_pbicg(rp, ci, v, b, para=po.default_cpara(epsilon=1e-10, abs_diff=1))
        assert r["ret"] == 0 and r["iters"] == iters and r["n_ax"] == 2 * iters + 1
        assert np.linalg.norm(r["x"] - xs) <= dist
        res = np.linalg.norm(port.csr_matvec(rp, ci, v, r["x"]) - b) ** 2 / n
        assert res <= 1.05e-10 and abs(res - r["residual"]) <= 1e-3 * r["residual"]
    # capped: the REAL enum's code, as every CPU complex loop returns it (clcg_eigen.cpp:751-755)
    n, rp, ci, v, b, xs = case1kc
    r = port.csolve_pbicg(rp, ci, v, b, para=po.default_cpara(epsilon=1e-10, abs_diff=1, max_iterations=7))
    assert r["ret"] == -1019 and r["iters"] == 7


def test_known_answer_case_10K(port, case10k):
    """CG at eps=1e-20/abs_diff reaches the fp64 floor of case_10K_B."""
    n, rp, ci, v, b, xs = case10k
    r = port.solve(po.LCG_CG, rp, ci, v, b, para=po.default_para(epsilon=1e-20, abs_diff=1))
    assert r["ret"] == 0 and r["iters"] == 367
    assert np.linalg.norm(r["x"] - xs) < 1e-11


def test_argument_checks(port, case10k):
    """lcg.cpp:150-155: error codes, checked in this order."""
    n, rp, ci, v, b, _ = case10k
    assert port.solve(0, rp, ci, v, b, para=po.default_para(max_iterations=-1))["ret"] == -1022
    assert port.solve(0, rp, ci, v, b, para=po.default_para(epsilon=0.0))["ret"] == -1021
    assert port.solve(0, rp, ci, v, b, para=po.default_para(epsilon=1.0))["ret"] == -1021


def test_already_optimized(port, case10k):
    """lcg.cpp:186-203: a start vector that already meets the tolerance returns 2."""
    n, rp, ci, v, b, xs = case10k
    for sid in (po.LCG_CG, po.LCG_CGS, po.LCG_BICGSTAB):
        r = port.solve(sid, rp, ci, v, b, m0=xs, para=po.default_para(epsilon=1e-6))
        assert r["ret"] == 2 and r["iters"] == 0
    assert port.solve(po.LCG_PCG, rp, ci, v, b, m0=xs, jacobi=True)["ret"] == 2


def test_unknown_solver_runs_cgs(port, case10k):
    """lcg.cpp:76-78: LCG_PCG/PG/SPG handed to lcg_solver silently run CGS."""
    n, rp, ci, v, b, _ = case10k
    a = port.solve(po.LCG_CGS, rp, ci, v, b)
    for sid in (1, 5, 6):
        c = port.solve(sid, rp, ci, v, b)
        assert c["iters"] == a["iters"] and np.array_equal(c["x"], a["x"])


def test_coo_matvec_equals_csr(port):
    """algebra.cpp:195-221 on the row-sorted fixture equals the CSR product exactly."""
    import os
    from conftest import GOLDEN
    from liblcg_amd.coo_io import coo_to_csr_host, read_coo_system
    n, row, col, val, b = read_coo_system(os.path.join(GOLDEN, "case_10K_A"))
    rp, ci, v = coo_to_csr_host(n, row, col, val)
    x = np.random.default_rng(3).standard_normal(n)
    assert np.array_equal(port.coo_matvec(row, col, val, x), port.csr_matvec(rp, ci, v, x))
    rp2, perm = port.coo_to_csr(row, col, n)
    assert np.array_equal(rp2, rp) and np.array_equal(col[perm], ci)
    d = port.csr_diag(rp, ci, v)
    assert set(np.unique(d)) <= {2.0, 3.0, 4.0}      # SURVEY.md section 4 fixture facts
