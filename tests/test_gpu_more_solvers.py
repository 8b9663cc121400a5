"""-m gpu: the remaining real solvers of lcg.cpp through the C ABI:
BiCGStab with restart (lcg.cpp:812-1034) and the box-constrained PG / SPG
(lcg_solver_constrained, lcg.cpp:1054-1446), against the oracle and the real liblcg's goldens."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def api():
    from liblcg_amd import api as a
    assert torch.cuda.is_available()
    return a


@pytest.fixture(scope="module")
def A10k(api, case10k):
    n, rp, ci, v, b, xs = case10k
    return api.CsrMatrix.from_csr(rp, ci, v)


def _run(api, A, sid, b, n, para, pfp=None):
    m = torch.zeros(n, dtype=torch.float64, device="cuda")
    info = api.lcg_solver("lcg_hip_csr_ax", pfp, m, torch.from_numpy(b).cuda(), n, para, A, sid)
    return info, m.cpu().numpy()


def test_bicgstab2_counts_two_iterations_per_pass_with_abs_diff(api, port, goldens, case10k, A10k):
    """abs_diff: the mid-iteration test advances t a second time (lcg.cpp:910-939)."""
    from oracle import pyoracle as po
    n, rp, ci, v, b, xs = case10k
    # capped run: deterministic count, tight agreement with the real liblcg
    info, x = _run(api, A10k, api.LCG_BICGSTAB2, b, n, api.lcg_default_parameters(epsilon=1e-10, abs_diff=1, max_iterations=31))
    gold = goldens["real/bicgstab2_max31/x"]
    assert info.ret == -1019 and info.iterations == 31
    assert np.linalg.norm(x - gold) / np.linalg.norm(gold) <= 1e-7
    # to convergence: BiCGStab-type sensitivity band (tests/test_oracle_sensitivity.py)
    info, x = _run(api, A10k, api.LCG_BICGSTAB2, b, n, api.lcg_default_parameters(epsilon=1e-10, abs_diff=1))
    ret, iters = goldens["real/bicgstab2_e10/meta"][:2]
    assert info.ret == ret == 0 and info.residual <= 1e-10
    # the restart test |r.r0| < 1e-6 makes the count chaotic: the oracle itself lands anywhere in
    # 280..364 under 1-ulp changes of b (tests/test_oracle_sensitivity.py)
    assert 0.5 * iters <= info.iterations <= 1.6 * iters
    assert np.linalg.norm(x - xs) <= 1e-3
    # without abs_diff one count per pass
    info, x = _run(api, A10k, api.LCG_BICGSTAB2, b, n, api.lcg_default_parameters(epsilon=1e-6, abs_diff=0))
    ret, iters = goldens["real/bicgstab2_e6/meta"][:2]
    assert info.ret == ret == 0 and abs(info.iterations - iters) <= max(2, 0.15 * iters)
    assert np.linalg.norm(x - goldens["real/bicgstab2_e6/x"]) / np.linalg.norm(x) <= 5e-3
    # progress callback sees both halves (k = 0, 1, 2, ...) and the oracle's first residuals
    seen = []
    _run(api, A10k, api.LCG_BICGSTAB2, b, n, api.lcg_default_parameters(epsilon=1e-10, abs_diff=1, max_iterations=6),
         pfp=lambda i, m, c, p, nn, k: seen.append((k, c)) or 0)
    trace = []
    import ctypes as C
    PROG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int)
    cb = PROG(lambda i, m, c, p, nn, k: trace.append((k, c)) or 0)
    inst, keep = port._inst(rp, ci, v, None, 1)
    mm = np.zeros(n); para = po.default_para(epsilon=1e-10, abs_diff=1, max_iterations=6)
    port.lib.orc_lbicgstab2(port.lib.orc_csr_ax, cb, mm.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_int(n),
                            C.byref(para), C.byref(inst))
    assert [k for k, _ in seen] == [k for k, _ in trace] == list(range(7))
    assert np.allclose([c for _, c in seen], [c for _, c in trace], rtol=1e-9)


@pytest.mark.parametrize("case", ["1K", "10K"])
def test_complex_pbicg_jacobi(api, port, case1kc, case10kc, case):
    """clpbicg (clcg_eigen.cpp:685-802; CLCG_PBICG through clcg_solver_preconditioned, the Eigen entry's default,
    clcg_eigen.h:87-92) with the complex Jacobi.  Oracle = restatement of the Eigen source (parity unpinned), plus the known
    answers.  The first dozen iterates track the oracle tightly; converged runs within the bands of the other complex
    BiCG-type loops; the fused built-in Jacobi and a user-written preconditioner walk the same iterates."""
    from oracle import pyoracle as po
    n, rp, ci, v, b, xs = case1kc if case == "1K" else case10kc
    A = api.CsrMatrix.from_csr(rp, ci, v)
    A.build_jacobi()
    bd = torch.from_numpy(b).cuda()
    m = torch.zeros(n, dtype=torch.complex128, device="cuda")
    info = api.clcg_solver_preconditioned("clcg_hip_csr_ax", "clcg_hip_jacobi_mx", None, m, bd, n,
                                          api.clcg_default_parameters(epsilon=1e-10, abs_diff=1), A, api.CLCG_PBICG)
    ref = port.csolve_pbicg(rp, ci, v, b, para=po.default_cpara(epsilon=1e-10, abs_diff=1))
    x = m.cpu().numpy()
    assert info.ret == ref["ret"] == 0 and info.residual <= 1e-10
    assert abs(info.iterations - ref["iters"]) <= 0.12 * ref["iters"]
    assert np.linalg.norm(x - ref["x"]) / np.linalg.norm(ref["x"]) <= 5e-5
    assert np.linalg.norm(x - xs) <= 1e-3
    p12 = api.clcg_default_parameters(epsilon=1e-10, abs_diff=1, max_iterations=12)
    m1 = torch.zeros_like(m); m2 = torch.zeros_like(m)
    i1 = api.clcg_solver_preconditioned("clcg_hip_csr_ax", "clcg_hip_jacobi_mx", None, m1, bd, n, p12, A, api.CLCG_PBICG)
    r12 = port.csolve_pbicg(rp, ci, v, b, para=po.default_cpara(epsilon=1e-10, abs_diff=1, max_iterations=12))
    assert i1.ret == r12["ret"] == -1019 and i1.iterations == 12
    assert np.linalg.norm(m1.cpu().numpy() - r12["x"]) <= 1e-9 * np.linalg.norm(r12["x"])
    from liblcg_amd import _lib
    lib = _lib.load()
    i2 = api.clcg_solver_preconditioned("clcg_hip_csr_ax", lambda inst, xp, zp, nn, lay, cj: lib.clcg_hip_jacobi_mx(A.h, xp, zp, nn, 0, 0),
                                        None, m2, bd, n, p12, A, api.CLCG_PBICG)
    assert i2.ret == -1019 and (m1 - m2).abs().max().item() <= 1e-8 * m1.abs().max().item()
    assert api.clcg_solver_preconditioned("clcg_hip_csr_ax", None, None, m1, bd, n, p12, A, api.CLCG_PBICG).ret == -1018


@pytest.mark.parametrize("case", ["1K", "10K"])
def test_complex_pcg_jacobi_sample10_workload(api, port, case1kc, case10kc, case):
    """clpcg (clcg_cuda.cu:403-558) with the complex Jacobi of sample10.cu:117,193 -- the reference's own
    GPU workload for these fixtures.  Oracle = restatement of the CUDA source (parity unpinned: no CPU
    implementation exists in the reference), plus the known answers case_*_cB."""
    from oracle import pyoracle as po
    n, rp, ci, v, b, xs = case1kc if case == "1K" else case10kc
    A = api.CsrMatrix.from_csr(rp, ci, v)
    A.build_jacobi()
    bd = torch.from_numpy(b).cuda()
    for eps, ad in ((1e-10, 1), (1e-6, 0)):
        m = torch.zeros(n, dtype=torch.complex128, device="cuda")
        info = api.clcg_solver_preconditioned("clcg_hip_csr_ax", "clcg_hip_jacobi_mx", None, m, bd, n,
                                              api.clcg_default_parameters(epsilon=eps, abs_diff=ad), A)
        ref = port.csolve_pcg(rp, ci, v, b, para=po.default_cpara(epsilon=eps, abs_diff=ad))
        x = m.cpu().numpy()
        assert info.ret == ref["ret"] == 0 and info.residual <= eps
        assert abs(info.iterations - ref["iters"]) <= 0.12 * ref["iters"]
        assert np.linalg.norm(x - ref["x"]) / np.linalg.norm(ref["x"]) <= (5e-5 if ad else 5e-3)
        if ad:
            assert np.linalg.norm(x - xs) <= 1e-5                   # far tighter than the shadow-residual solvers
    # a user-written preconditioner (Python callable launching the library's kernel) takes the unfused path
    from liblcg_amd import _lib
    lib = _lib.load()
    m1 = torch.zeros(n, dtype=torch.complex128, device="cuda"); m2 = torch.zeros_like(m1)
    p12 = api.clcg_default_parameters(epsilon=1e-10, abs_diff=1, max_iterations=12)
    i1 = api.clcg_solver_preconditioned("clcg_hip_csr_ax", lambda inst, xp, zp, nn, lay, cj: lib.clcg_hip_jacobi_mx(A.h, xp, zp, nn, 0, 0),
                                        None, m1, bd, n, p12, A)
    i2 = api.clcg_solver_preconditioned("clcg_hip_csr_ax", "clcg_hip_jacobi_mx", None, m2, bd, n, p12, A)
    assert i1.ret == i2.ret == -1019 and i1.iterations == i2.iterations == 12
    assert (m1 - m2).abs().max().item() <= 1e-8 * m2.abs().max().item()      # fused vs separate passes: rounding only
    assert api.clcg_solver_preconditioned("clcg_hip_csr_ax", None, None, m1, bd, n, p12, A).ret == -1018


def test_bicgstab2_restart_and_argument_checks(api, port, case10k, A10k):
    from oracle import pyoracle as po
    n, rp, ci, v, b, xs = case10k
    # a huge restart_epsilon restarts every pass (lcg.cpp:982-997)
    info, x = _run(api, A10k, api.LCG_BICGSTAB2, b, n,
                   api.lcg_default_parameters(epsilon=1e-10, abs_diff=1, restart_epsilon=1e3, max_iterations=40))
    ref = port.solve(4, rp, ci, v, b, para=po.default_para(epsilon=1e-10, abs_diff=1, restart_epsilon=1e3, max_iterations=40))
    assert info.ret == ref["ret"] == -1019 and info.iterations == ref["iters"] == 40
    assert np.linalg.norm(x - ref["x"]) / np.linalg.norm(ref["x"]) <= 1e-9
    # lcg.cpp:820-821: epsilon >= 1 is reported as a restart-epsilon error by this solver
    assert _run(api, A10k, api.LCG_BICGSTAB2, b, n, api.lcg_default_parameters(epsilon=1.5))[0].ret == -1020
    assert _run(api, A10k, api.LCG_BICGSTAB2, b, n, api.lcg_default_parameters(restart_epsilon=0.0))[0].ret == -1020
    assert _run(api, A10k, api.LCG_BICGSTAB2, b, n, api.lcg_default_parameters(epsilon=0.0))[0].ret == -1021


@pytest.mark.parametrize("tag", ["pg_40", "spg_40", "pg_150", "spg_60"])
def test_box_constrained_vs_golden(api, goldens, case10k, A10k, tag):
    n, rp, ci, v, b, xs = case10k
    ret, iters, sid, ad, maxit, n_ax = goldens[f"box/{tag}/meta"]
    eps, resid = goldens[f"box/{tag}/fl"]
    gold = goldens[f"box/{tag}/x"]
    low = torch.full((n,), -5.0, dtype=torch.float64, device="cuda"); hig = torch.full((n,), 8.0, dtype=torch.float64, device="cuda")
    m = torch.zeros(n, dtype=torch.float64, device="cuda")
    from liblcg_amd import _lib
    lib = _lib.load()
    lib.lcg_hip_set_profiling(1)
    info = api.lcg_solver_constrained("lcg_hip_csr_ax", None, m, torch.from_numpy(b).cuda(), low, hig, n,
                                      api.lcg_default_parameters(epsilon=float(eps), abs_diff=int(ad), max_iterations=int(maxit)),
                                      A10k, int(sid))
    calls = lib.lcg_hip_last_ax_calls(); lib.lcg_hip_set_profiling(0)
    x = m.cpu().numpy()
    assert info.ret == ret and info.iterations == iters
    assert calls == n_ax                                            # same number of A.x calls: same line-search path
    assert x.min() >= -5.0 and x.max() <= 8.0
    assert np.array_equal(x <= -5.0, gold <= -5.0) and np.array_equal(x >= 8.0, gold >= 8.0)    # same active set
    assert np.linalg.norm(x - gold) / np.linalg.norm(gold) <= 1e-9
    assert abs(info.residual - resid) <= 1e-8 * resid


def test_box_host_memory_progress_and_errors(api, port, case10k, A10k):
    from oracle import pyoracle as po
    n, rp, ci, v, b, xs = case10k
    low, hig = np.full(n, -5.0), np.full(n, 8.0)
    for sid in (api.LCG_PG, api.LCG_SPG):
        m = np.full(n, 20.0)                                        # start outside the box: projected first
        seen = []
        info = api.lcg_solver_constrained("lcg_hip_csr_ax", lambda i, mp, c, p, nn, k: seen.append(k) or 0, m, b, low, hig, n,
                                          api.lcg_default_parameters(epsilon=1e-10, abs_diff=1, max_iterations=15), A10k, sid)
        ref = port.solve_box(sid, rp, ci, v, b, low, hig, m0=np.full(n, 20.0),
                             para=po.default_para(epsilon=1e-10, abs_diff=1, max_iterations=15))
        assert info.ret == ref["ret"] == -1019 and seen == list(range(16))
        assert np.linalg.norm(m - ref["x"]) / np.linalg.norm(ref["x"]) <= 1e-9
    p = api.lcg_default_parameters
    run = lambda sid, **kw: api.lcg_solver_constrained("lcg_hip_csr_ax", None, np.zeros(n), b, low, hig, n, p(**kw), A10k, sid).ret
    assert run(api.LCG_PG, step=0.0) == -1015 and run(api.LCG_PG, epsilon=1.0) == -1015     # lcg.cpp:1063
    assert run(api.LCG_SPG, epsilon=1.0) == -1021 and run(api.LCG_SPG, step=-1.0) == -1015
    assert run(api.LCG_SPG, sigma=1.0) == -1014 and run(api.LCG_SPG, beta=0.0) == -1013 and run(api.LCG_SPG, maxi_m=0) == -1012
    from liblcg_amd import _lib
    lib = _lib.load()
    a = torch.linspace(-10, 10, 1001, dtype=torch.float64, device="cuda")
    lo = torch.full_like(a, -2.0); hi = torch.full_like(a, 3.0)
    assert lib.lcg_hip_set2box(1001, lo.data_ptr(), hi.data_ptr(), a.data_ptr()) == 0
    api.synchronize()
    assert torch.equal(a, torch.linspace(-10, 10, 1001, dtype=torch.float64, device="cuda").clamp(-2.0, 3.0))


def test_cg_one_reduction_schedule_matches_reference_cg(api, goldens, case10k, A10k):
    """lcg_hip_set_cg_schedule(ONE_REDUCTION): the Chronopoulos-Gear arrangement of lcg.cpp:206-264 that
    the sharded path uses (one all-reduce per iteration).  Same iterates in exact arithmetic, so on the
    bundled system it must meet the bands of the classic schedule: the real liblcg's x to 1e-9 (tight) /
    1e-6 (loose), its iteration counts to +-3 / +-2, the 25-iteration iterate to 1e-9, and the progress
    callback contract (k = 0..t, residual of the classic recurrence)."""
    n, rp, ci, v, b, xs = case10k
    api.set_cg_schedule(api.CG_ONE_REDUCTION)
    try:
        for tag, kw, xtol, itol in (("e12", dict(epsilon=1e-12, abs_diff=1), 1e-9, 3), ("e6", dict(epsilon=1e-6), 1e-6, 2)):
            info, x = _run(api, A10k, api.LCG_CG, b, n, api.lcg_default_parameters(**kw))
            ret, iters = goldens[f"real/cg_{tag}/meta"][:2]
            gold = goldens[f"real/cg_{tag}/x"]
            assert info.ret == ret == 0 and abs(info.iterations - iters) <= itol
            assert np.linalg.norm(x - gold) / np.linalg.norm(gold) <= xtol
            assert np.linalg.norm(x - xs) <= 1.1 * np.linalg.norm(gold - xs) + 1e-9
        seen = []
        pfp = lambda i, mp, c, p, nn, k: seen.append((k, c)) or 0
        info, x = _run(api, A10k, api.LCG_CG, b, n, api.lcg_default_parameters(epsilon=1e-12, abs_diff=1, max_iterations=25), pfp)
        gold = goldens["real/cg_max25/x"]
        assert info.ret == -1019 and info.iterations == 25 and [k for k, _ in seen] == list(range(26))
        assert np.linalg.norm(x - gold) / np.linalg.norm(gold) <= 1e-9
        assert abs(seen[-1][1] - goldens["real/cg_max25/fl"][1]) <= 1e-8 * seen[-1][1]
        # x0 already solves the system: same early exit as the classic schedule (lcg.cpp:186-203)
        m = torch.from_numpy(xs.copy()).cuda()
        bd = torch.from_numpy(np.asarray(A10k_matvec(rp, ci, v, xs))).cuda()
        info = api.lcg_solver("lcg_hip_csr_ax", None, m, bd, n, api.lcg_default_parameters(epsilon=1e-10), A10k, api.LCG_CG)
        assert info.ret == 2 and info.iterations == 0                 # LCG_ALREADY_OPTIMIZIED
    finally:
        api.set_cg_schedule(api.CG_AUTO)


def A10k_matvec(rp, ci, v, x):
    import scipy.sparse as sp
    return sp.csr_matrix((v, ci, rp), shape=(len(rp) - 1, len(rp) - 1)) @ x


def test_tiny_systems_bicgstab2_and_the_box_solvers(api, port):
    """n = 1 ... 257 (ragged SPD, with a right-hand side and with b = 0): lbicgstab2 under both stop rules, lpg and lspg between
    bounds that cut the solution off -- return code, iteration count (+-2) and solution as the oracle's."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 5, 63, 64, 65, 129, 257):
        off = rng.standard_normal(n - 1) * (rng.random(n - 1) < 0.7) if n > 1 else np.zeros(0)
        d = np.full(n, 0.5); rows, cols, ov = list(range(n)), list(range(n)), []
        for i in range(n - 1):
            if off[i] != 0.0:
                rows += [i, i + 1]; cols += [i + 1, i]; ov += [off[i], off[i]]
                d[i] += abs(off[i]); d[i + 1] += abs(off[i])
        row = np.array(rows, np.int32); col = np.array(cols, np.int32); val = np.concatenate([d, np.array(ov, dtype=np.float64)])
        o = np.lexsort((col, row)); row, col, val = row[o], col[o], val[o]
        rp = np.zeros(n + 1, np.int32); np.add.at(rp, row + 1, 1); rp = np.cumsum(rp).astype(np.int32)
        A = api.CsrMatrix.from_csr(rp, col, val)
        b = port.csr_matvec(rp, col, val, rng.standard_normal(n))
        low = np.full(n, -0.5); hig = np.full(n, 0.8)
        for bh in (b, np.zeros(n)):
            bd = torch.from_numpy(bh).cuda()
            cases = [("bicgstab2", api.LCG_BICGSTAB2, eps, ad) for eps, ad in ((1e-12, 1), (1e-10, 0))] + [("pg", api.LCG_PG, 1e-10, 1), ("spg", api.LCG_SPG, 1e-10, 1)]
            for name, sid, eps, ad in cases:
                m = torch.zeros(n, dtype=torch.float64, device="cuda")
                cap = 300 if sid == api.LCG_BICGSTAB2 else 500
                para = api.lcg_default_parameters(epsilon=eps, abs_diff=ad, max_iterations=cap)
                opara = po.default_para(epsilon=eps, abs_diff=ad, max_iterations=cap)
                if sid == api.LCG_BICGSTAB2:
                    info = api.lcg_solver("lcg_hip_csr_ax", None, m, bd, n, para, A, sid)
                    ref = port.solve(4, rp, col, val, bh, para=opara)
                else:
                    info = api.lcg_solver_constrained("lcg_hip_csr_ax", None, m, bd, torch.from_numpy(low).cuda(), torch.from_numpy(hig).cuda(), n, para, A, sid)
                    ref = port.solve_box(sid, rp, col, val, bh, low, hig, para=opara)
                tag = (n, name, eps, ad, "b = 0" if not bh.any() else "b", info.ret, info.iterations, ref["ret"], ref["iters"])
                assert info.ret == ref["ret"] and abs(info.iterations - ref["iters"]) <= 2, tag
                if np.all(np.isfinite(ref["x"])):
                    assert np.linalg.norm(m.cpu().numpy() - ref["x"]) <= 1e-7 * max(1.0, np.linalg.norm(ref["x"])), tag
        A.destroy()
