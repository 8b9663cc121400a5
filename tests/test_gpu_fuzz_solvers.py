"""-m gpu: the insensitive recurrences (CG, PCG + Jacobi, CGS) on RANDOM generated systems against the oracle --
sizes from a few hundred to 60,000 rows, bands from 1 to n/3, scrambled columns, both stop rules -- with the plain and
with the packed-column A.x.  Bands: x to 1e-9 of the oracle's, iteration counts within +-3 (tests/test_gpu_solvers.py
derives them on the bundled system; the generated ones are better conditioned)."""
import numpy as np
import pytest

from conftest import FUZZ_SEED_OFFSET, check_converged_run

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_random_systems_against_the_oracle(port):
    from liblcg_amd import _lib, api
    from oracle import pyoracle as po
    lib = _lib.load()
    rng = np.random.default_rng(99 + FUZZ_SEED_OFFSET)
    for case in range(10):
        n = int(rng.integers(300, 60000))
        band = 0 if case % 4 == 3 else int(rng.integers(1, max(2, n // 3)))
        abs_diff = int(case % 2)
        eps = 1e-12 if abs_diff else 1e-14
        seed = int(rng.integers(1, 1000))
        A = api.CsrMatrix.generate(n, 16, band, True, seed, 0.01)
        A.build_jacobi()
        rp, ci, v = A.arrays_to_host()
        xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, seed, 0, n, xt)
        b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
        bh = b.cpu().numpy(); xth = xt.cpu().numpy()
        for sid, name in ((api.LCG_CG, "cg"), (api.LCG_PCG, "pcg"), (api.LCG_CGS, "cgs")):
            shared = {}
            for packed in (0, 1):
                assert lib.lcg_hip_csr_set_packed(A.h, packed) == 0

                def solve_gpu(cap, sid=sid):
                    m = torch.zeros(n, dtype=torch.float64, device="cuda")
                    para = api.lcg_default_parameters(epsilon=eps, abs_diff=abs_diff, max_iterations=cap)
                    if sid == api.LCG_PCG:
                        info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, para, A)
                    else:
                        info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, sid)
                    return info.ret, info.iterations, info.residual, m.cpu().numpy()
                check_converged_run(port, solve_gpu, sid, rp, ci, v, bh, eps, abs_diff, jacobi=(sid == api.LCG_PCG),
                                    tag=(case, n, band, abs_diff, name, packed), wide=(sid == api.LCG_CGS), cache=shared, xt=xth)
                if sid != api.LCG_CGS and packed == 1:      # the reference's own recurrence: late iterates too
                    api.set_cg_schedule(api.CG_CLASSIC)
                    try:
                        check_converged_run(port, solve_gpu, sid, rp, ci, v, bh, eps, abs_diff, jacobi=(sid == api.LCG_PCG),
                                            tag=(case, n, band, abs_diff, name, packed, "classic"), cache=shared, xt=xth, late=True)
                    finally:
                        api.set_cg_schedule(api.CG_AUTO)
        A.destroy()
        # the non-symmetric twin: lbicgstab (lcg.cpp:629-794) and lcgs (lcg.cpp:437-612) on A != A^T
        A = api.CsrMatrix.generate(n, 16, band, False, seed, 0.01)
        rp, ci, v = A.arrays_to_host()
        b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
        bh = b.cpu().numpy()
        prng = np.random.default_rng(case)
        for sid, name in ((api.LCG_BICGSTAB, "bicgstab"), (api.LCG_CGS, "cgs")):
            shared = {}
            for packed in (0, 1):
                assert lib.lcg_hip_csr_set_packed(A.h, packed) == 0

                def solve_gpu(cap, sid=sid):
                    m = torch.zeros(n, dtype=torch.float64, device="cuda")
                    info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=eps, abs_diff=abs_diff, max_iterations=cap), A, sid)
                    return info.ret, info.iterations, info.residual, m.cpu().numpy()
                check_converged_run(port, solve_gpu, sid, rp, ci, v, bh, eps, abs_diff, tag=(case, n, band, abs_diff, name, packed, "non-symmetric"), wide=True, cache=shared, xt=xth)
        A.destroy()


def test_short_row_systems_against_the_oracle(port):
    """The same on systems with 5 and 11 entries per row (2 and 5 offset pairs + the diagonal): the one-wavefront-per-block
    product with one and with two partial sums per row (k_spmv_run1 / k_spmv_run1d<1>, <2>) under CG (two launches per
    iteration below 2^17 rows, classic above), PCG + Jacobi, CGS and BiCGStab; constant diagonals (runs) and scrambled
    columns (no runs: the LDS-staged kernel and its dot-carrying twin)."""
    from liblcg_amd import _lib, api
    from oracle import pyoracle as po
    lib = _lib.load()
    rng = np.random.default_rng(7 + FUZZ_SEED_OFFSET)
    seen = set()
    for case, (npairs, n, band) in enumerate(((2, 3000, 40), (2, 150000, 700), (5, 20000, 300), (5, 140000, 0), (2, 50000, 0), (5, 9000, 2000))):
        seed = int(rng.integers(1, 1000))
        abs_diff = case % 2
        eps = 1e-12 if abs_diff else 1e-14
        A = api.CsrMatrix.generate(n, npairs, band, True, seed, 0.01)
        A.build_jacobi()
        rp, ci, v = A.arrays_to_host()
        xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, seed, 0, n, xt)
        b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
        bh = b.cpu().numpy(); xth = xt.cpu().numpy()
        for sid, name in ((api.LCG_CG, "cg"), (api.LCG_PCG, "pcg"), (api.LCG_CGS, "cgs"), (api.LCG_BICGSTAB, "bicgstab")):
            def solve_gpu(cap, sid=sid):
                m = torch.zeros(n, dtype=torch.float64, device="cuda")
                para = api.lcg_default_parameters(epsilon=eps, abs_diff=abs_diff, max_iterations=cap)
                if sid == api.LCG_PCG:
                    info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, para, A)
                else:
                    info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, sid)
                seen.add(lib.lcg_hip_csr_last_kernel(A.h).decode().split(" ")[0])
                return info.ret, info.iterations, info.residual, m.cpu().numpy()
            tag = (case, npairs, n, band, abs_diff, name)
            # (bands from the oracle's own response to 1-ulp changes of b on THIS system: the recurrences amplify rounding differently
            #  from system to system -- conftest.check_converged_run)
            shared = {}
            check_converged_run(port, solve_gpu, sid, rp, ci, v, bh, eps, abs_diff, jacobi=(sid == api.LCG_PCG), tag=tag,
                                wide=sid in (api.LCG_CGS, api.LCG_BICGSTAB), cache=shared, xt=xth)
            if sid in (api.LCG_CG, api.LCG_PCG):            # the reference's own recurrence: late iterates too
                api.set_cg_schedule(api.CG_CLASSIC)
                try:
                    check_converged_run(port, solve_gpu, sid, rp, ci, v, bh, eps, abs_diff, jacobi=(sid == api.LCG_PCG), tag=tag + ("classic",),
                                        cache=shared, xt=xth, late=True)
                finally:
                    api.set_cg_schedule(api.CG_AUTO)
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            # and six capped iterations walk the oracle's iterates to rounding, whatever the system
            m.zero_()
            p6 = api.lcg_default_parameters(epsilon=eps, abs_diff=abs_diff, max_iterations=6)
            if sid == api.LCG_PCG:
                i6 = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, p6, A)
            else:
                i6 = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, p6, A, sid)
            r6 = port.solve(sid, rp, ci, v, bh, para=po.default_para(epsilon=eps, abs_diff=abs_diff, max_iterations=6), jacobi=(sid == api.LCG_PCG))
            assert i6.ret == r6["ret"] == -1019 and i6.iterations == 6, tag
            assert np.linalg.norm(m.cpu().numpy() - r6["x"]) <= 1e-13 * np.linalg.norm(r6["x"]), tag
        A.destroy()
    assert {"k_spmv_run1d", "k_spmv_lds1d"} <= seen, seen
