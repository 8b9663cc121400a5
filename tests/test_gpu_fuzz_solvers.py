"""-m gpu: the insensitive recurrences (CG, PCG + Jacobi, CGS) on RANDOM generated systems against the oracle --
sizes from a few hundred to 60,000 rows, bands from 1 to n/3, scrambled columns, both stop rules -- with the plain and
with the packed-column A.x.  Bands: x to 1e-9 of the oracle's, iteration counts within +-3 (tests/test_gpu_solvers.py
derives them on the bundled system; the generated ones are better conditioned)."""
import numpy as np
import pytest

from conftest import FUZZ_SEED_OFFSET

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
        bh = b.cpu().numpy()
        for sid, name in ((api.LCG_CG, "cg"), (api.LCG_PCG, "pcg"), (api.LCG_CGS, "cgs")):
            ref = port.solve(sid, rp, ci, v, bh, para=po.default_para(epsilon=eps, abs_diff=abs_diff), jacobi=(sid == api.LCG_PCG))
            for packed in (0, 1):
                assert lib.lcg_hip_csr_set_packed(A.h, packed) == 0
                m = torch.zeros(n, dtype=torch.float64, device="cuda")
                para = api.lcg_default_parameters(epsilon=eps, abs_diff=abs_diff)
                if sid == api.LCG_PCG:
                    info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, para, A)
                else:
                    info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, sid)
                x = m.cpu().numpy()
                tag = (case, n, band, abs_diff, name, packed)
                assert info.ret == ref["ret"], (tag, info.ret, ref["ret"])
                assert abs(info.iterations - ref["iters"]) <= 3, (tag, info.iterations, ref["iters"])
                assert np.linalg.norm(x - ref["x"]) <= 1e-9 * np.linalg.norm(ref["x"]), tag
        A.destroy()
        # the non-symmetric twin: lbicgstab (lcg.cpp:629-794) and lcgs (lcg.cpp:437-612) on A != A^T
        A = api.CsrMatrix.generate(n, 16, band, False, seed, 0.01)
        rp, ci, v = A.arrays_to_host()
        b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
        bh = b.cpu().numpy()
        prng = np.random.default_rng(case)
        for sid, name in ((api.LCG_BICGSTAB, "bicgstab"), (api.LCG_CGS, "cgs")):
            opara = po.default_para(epsilon=eps, abs_diff=abs_diff)
            ref = port.solve(sid, rp, ci, v, bh, para=opara)
            sens, dit = 0.0, 0
            for _ in range(3):
                alt = port.solve(sid, rp, ci, v, bh * (1.0 + 1e-16 * prng.standard_normal(n)), para=opara)
                sens = max(sens, np.linalg.norm(alt["x"] - ref["x"]) / np.linalg.norm(ref["x"]))
                dit = max(dit, abs(alt["iters"] - ref["iters"]))
            for packed in (0, 1):
                assert lib.lcg_hip_csr_set_packed(A.h, packed) == 0
                m = torch.zeros(n, dtype=torch.float64, device="cuda")
                info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=eps, abs_diff=abs_diff), A, sid)
                x = m.cpu().numpy()
                tag = (case, n, band, abs_diff, name, packed, info.iterations, ref["iters"], sens, dit)
                assert info.ret == ref["ret"] == 0, tag
                assert abs(info.iterations - ref["iters"]) <= max(3, 3 * dit, 0.05 * ref["iters"]), tag
                assert np.linalg.norm(x - ref["x"]) <= max(1e-9, 50 * sens) * np.linalg.norm(ref["x"]), tag
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
        bh = b.cpu().numpy()
        for sid, name in ((api.LCG_CG, "cg"), (api.LCG_PCG, "pcg"), (api.LCG_CGS, "cgs"), (api.LCG_BICGSTAB, "bicgstab")):
            opara = po.default_para(epsilon=eps, abs_diff=abs_diff)
            ref = port.solve(sid, rp, ci, v, bh, para=opara, jacobi=(sid == api.LCG_PCG))
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            para = api.lcg_default_parameters(epsilon=eps, abs_diff=abs_diff)
            if sid == api.LCG_PCG:
                info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, para, A)
            else:
                info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, sid)
            seen.add(lib.lcg_hip_csr_last_kernel(A.h).decode().split(" ")[0])
            x = m.cpu().numpy()
            tag = (case, npairs, n, band, abs_diff, name, info.iterations, ref["iters"])
            assert info.ret == ref["ret"] == 0, tag
            # bands from the oracle's own response to 1-ulp changes of b on THIS system (the recurrences amplify rounding
            # differently from system to system: on the first one here CG moves by 2e-8, PCG by 1e-15, BiCGStab by 5e-6)
            sens, dit = 0.0, 0
            for k in range(2):
                alt = port.solve(sid, rp, ci, v, bh * (1.0 + 1e-16 * np.random.default_rng(10 * case + k).standard_normal(n)), para=opara,
                                 jacobi=(sid == api.LCG_PCG))
                sens = max(sens, np.linalg.norm(alt["x"] - ref["x"]) / np.linalg.norm(ref["x"]))
                dit = max(dit, abs(alt["iters"] - ref["iters"]))
            tag = tag + (sens, dit)
            assert abs(info.iterations - ref["iters"]) <= max(3, 3 * dit, 0.05 * ref["iters"]), tag
            assert np.linalg.norm(x - ref["x"]) <= max(1e-9, 20 * sens) * np.linalg.norm(ref["x"]), tag
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
