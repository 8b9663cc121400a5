"""-m gpu: the BASELINE.json configurations that round 1 only ran from scripts/, at their full sizes, plus oracle parity of
the non-symmetric solvers on matrices that really are non-symmetric.

  configs[1]     5-point Laplacian 1000 x 1000 (1M rows), PCG + Jacobi           lcg.cpp:293-434, sample1.cpp:55-62,98-107
  configs[4](i)  10M-row non-symmetric CSR, BiCGStab                             lcg.cpp:629-794
  BiCGStab / CGS vs the oracle on generated non-symmetric systems (<= 60K rows)  lcg.cpp:437-612, 629-794

Bands.  PCG on the Laplacian is insensitive (the oracle moves by 1e-13 under a 1-ulp change of b, 1427 iterations
unchanged): x to 1e-9, counts +-3.  The non-symmetric recurrences stop one iteration earlier or later under the same
change, so two correct runs differ by about the error left at the stop: BiCGStab 4e-8 (eps = 1e-12, abs_diff = 1),
CGS 3e-10, counts within a few percent -- tests/test_oracle_sensitivity.py measures those responses on the oracle itself
and pins the bands used here (>= 20x headroom); the first 6 iterates are compared at 1e-13.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

# (pattern, n, band, seed): constant diagonals, scrambled, row-random band -- the systems of test_oracle_sensitivity.py
NONSYM_SYSTEMS = [(1, 60000, 3000, 5), (0, 40000, 0, 6), (2, 50000, 2048, 7)]
NONSYM_BANDS = {3: (1e-6, 0.15), 2: (1e-8, 0.15)}       # solver id -> (rel x vs oracle, iteration band)


@pytest.fixture(scope="module")
def api():
    from liblcg_amd import api as a
    assert torch.cuda.is_available(), "GPU tests need the MI355X; there is no CPU fallback"
    return a


def test_config1_laplacian_pcg_jacobi_full_size(api, port):
    """BASELINE configs[1]: laplace2d(1000, 1000) + build_jacobi, PCG to abs_diff = 1, eps = 1e-10 and a run capped at
    500 iterations, both against the oracle's lpcg with the same reciprocal-diagonal Jacobi."""
    from oracle import pyoracle as po
    nx = ny = 1000
    n = nx * ny
    A = api.CsrMatrix.laplace2d(nx, ny)
    assert A.nnz == 4_996_000
    A.build_jacobi()
    rp, ci, v = A.arrays_to_host()
    xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    bh = b.cpu().numpy()
    assert np.array_equal(bh, port.csr_matvec(rp, ci, v, xt.cpu().numpy()))     # integers times doubles in one order: exact
    for eps, ad, cap, want_ret in ((1e-10, 1, 0, 0), (1e-300, 0, 500, -1019)):
        ref = port.solve(po.LCG_PCG, rp, ci, v, bh, para=po.default_para(epsilon=eps, abs_diff=ad, max_iterations=cap), jacobi=True, threads=8)
        m = torch.zeros_like(xt)
        info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n,
                                             api.lcg_default_parameters(epsilon=eps, abs_diff=ad, max_iterations=cap), A)
        x = m.cpu().numpy()
        assert info.ret == ref["ret"] == want_ret, (cap, info.ret, ref["ret"])
        assert abs(info.iterations - ref["iters"]) <= (0 if cap else 3), (info.iterations, ref["iters"])
        assert np.linalg.norm(x - ref["x"]) <= 1e-9 * np.linalg.norm(ref["x"]), cap
        assert abs(info.residual - ref["residual"]) <= 1e-6 * ref["residual"]
        if not cap:
            assert info.residual <= 1e-10 and np.linalg.norm(x - xt.cpu().numpy()) <= 1e-4 * np.linalg.norm(x)


def test_config4_nonsymmetric_bicgstab_full_size(api, port):
    """BASELINE configs[4](i): the 10M-row generated system with unmirrored values, BiCGStab to abs_diff = 1,
    eps = 1e-10 -- size-independent properties (no CPU reference at this size): A really is non-symmetric, the solve
    returns convergence, recovers x_true, and the residual it monitored is the residual of its answer (second A.x)."""
    from oracle import pyoracle as po
    n = 10_000_000
    A = api.CsrMatrix.generate(n, 16, 131072, False, 1, 0.01)
    assert 32.5 * n < A.nnz <= 33 * n
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    y = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    Ax, Ay = torch.empty_like(x), torch.empty_like(x)
    A.spmv(x, Ax); A.spmv(y, Ay); api.synchronize()
    assert abs(api.dot(x, Ay) - api.dot(y, Ax)) >= 1e-6 * abs(api.dot(x, Ay))       # A != A^T
    del y, Ay
    xt = torch.empty_like(x); api.gen_xtrue(n, 1, 0, n, xt)
    b = torch.empty_like(x); A.spmv(xt, b); api.synchronize()
    m = torch.zeros_like(x)
    info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=1e-10, abs_diff=1), A, api.LCG_BICGSTAB)
    assert info.ret == 0 and info.residual <= 1e-10 and 5 <= info.iterations <= 200
    # how close to x_true the stop leaves the answer: |e| <= |A^-1| |r| with |r| = residual * N <= 1e-3 and |A^-1| <= 1 / 0.01
    # (rows are diagonally dominant by the generator's 0.01 margin) => |e| / |x_true| <= 0.1 / 1826 = 5.5e-5; which iteration
    # happens to cross the threshold decides where below that the run lands (seen: 0.7e-5 .. 1.2e-5)
    assert ((m - xt).norm() / xt.norm()).item() <= 5.5e-5
    A.spmv(m, Ax); api.synchronize()
    true_res = (Ax - b).norm().item() / n
    assert true_res <= 1.05e-10 and abs(true_res - info.residual) <= 0.05 * info.residual
    # CGS, the reference's default solver (lcg.h:72), on the same system
    m.zero_()
    info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=1e-10, abs_diff=1), A, api.LCG_CGS)
    assert info.ret == 0 and ((m - xt).norm() / xt.norm()).item() <= 5.5e-5
    # The ARITHMETIC at full size: four capped iterations of both loops against the oracle on the very same
    # 3.3e8 entries.  The band is rounding, not a stop: the oracle adds its 1e7-term inner products left to right, the device in
    # blocks -- a few 1e-13 per sum -- so the iterates agree to 1e-10 where a wrong coefficient would show at 1e-1.
    rp, ci, v = A.arrays_to_host()
    bh = b.cpu().numpy()
    for sid in (api.LCG_BICGSTAB, api.LCG_CGS):
        ref = port.solve(sid, rp, ci, v, bh, para=po.default_para(epsilon=1e-300, abs_diff=1, max_iterations=4))
        m.zero_()
        info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=1e-300, abs_diff=1, max_iterations=4), A, sid)
        x = m.cpu().numpy()
        rel = float(np.linalg.norm(x - ref["x"]) / np.linalg.norm(ref["x"]))
        assert info.ret == ref["ret"] == -1019 and info.iterations == ref["iters"] == 4 and rel <= 1e-10, (sid, info.ret, ref["ret"], rel)
        if ref["residual"] > 0.0:
            assert abs(info.residual - ref["residual"]) <= 1e-9 * ref["residual"], (sid, info.residual, ref["residual"])


@pytest.mark.parametrize("pattern,band,family", [(1, 131072, "k_spmv_ldsp (LDS-staged, run blocks"), (2, 131072, "k_tile_spmv"),
                                                 (0, 0, "k_bin_expand + k_bin_reduce")])
def test_config2_against_the_oracle_at_full_size(api, port, pattern, band, family):
    """BASELINE configs[2] AT ITS OWN SIZE against the oracle (missing 3): the 10M-row, 3.3e8-entry SPD system of
    bench.py -- the headline's constant diagonals, the row-random band and the scrambled columns, each multiplied by the kernel
    family the library chooses for it -- (1) A.x row by row against the oracle's serial CSR product at 1e-13 |A||x| (the oracle's
    row-wise bound), plain and as the solver runs it (carrying the dot: the first capped iterate below), and (2) four capped
    iterations of lcg() (lcg.cpp:206-264) against the oracle's loop on the very same arrays at 1e-10 (rounding of 1e7-term inner
    products added in another order -- a wrong coefficient shows at 1e-1), monitored residual included."""
    from liblcg_amd import _lib
    from oracle import pyoracle as po
    lib = _lib.load()
    n = 10_000_000
    A = api.CsrMatrix.generate(n, 16, band, True, 1, 0.01, pattern=pattern)
    assert 32.5 * n < A.nnz <= 33 * n
    rp, ci, v = A.arrays_to_host()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) - 0.5
    y = torch.empty_like(x)
    A.spmv(x, y); api.synchronize()
    name = lib.lcg_hip_csr_last_kernel(A.h).decode()
    assert name.startswith(family), name
    xh = x.cpu().numpy()
    ref = port.csr_matvec(rp, ci, v, xh)
    bound = port.csr_matvec(rp, ci, np.abs(v), np.abs(xh))
    worst = float(np.max(np.abs(y.cpu().numpy() - ref) / bound))
    assert worst <= 1e-13, (name, worst)
    del ref, bound, xh, x, y
    xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    bh = b.cpu().numpy()
    for ad in (1, 0):
        ref = port.solve(po.LCG_CG, rp, ci, v, bh, para=po.default_para(epsilon=1e-300, abs_diff=ad, max_iterations=4), threads=8)
        m = torch.zeros_like(xt)
        ws = [torch.empty_like(xt) for _ in range(3)]
        info = api.lcg("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=1e-300, abs_diff=ad, max_iterations=4), A, *ws)
        xg = m.cpu().numpy()
        rel = float(np.linalg.norm(xg - ref["x"]) / np.linalg.norm(ref["x"]))
        assert info.ret == ref["ret"] == -1019 and info.iterations == ref["iters"] == 4, (info.ret, ref["ret"], info.iterations, ref["iters"])
        assert rel <= 1e-10, (name, ad, rel)
        assert abs(info.residual - ref["residual"]) <= 1e-9 * ref["residual"], (ad, info.residual, ref["residual"])
        # ... and once more the way bench.py times it: the reference's plain call, workspaces left at nullptr (lcg.h:135-137) -- the
        # library's own vectors, placement AUTOMATIC (roles dealt by the clock, a walk where the box asks for one), the zero-guess skip
        m2 = torch.zeros_like(xt)
        info2 = api.lcg("lcg_hip_csr_ax", None, m2, b, n, api.lcg_default_parameters(epsilon=1e-300, abs_diff=ad, max_iterations=4), A)
        assert info2.ret == -1019 and info2.iterations == 4
        assert torch.equal(m2, m), (name, ad)           # placement and the pool change no bit
        assert abs(info2.residual - ref["residual"]) <= 1e-9 * ref["residual"]
        del m2
    # the product inside the solve was the family's dot-carrying form where it has one
    inside = lib.lcg_hip_csr_last_kernel(A.h).decode()
    assert inside.startswith(family.split(" (")[0]), inside
    A.destroy()


@pytest.mark.parametrize("pattern,n,band,seed", NONSYM_SYSTEMS)
def test_nonsymmetric_bicgstab_and_cgs_against_the_oracle(api, port, pattern, n, band, seed):
    """lbicgstab (lcg.cpp:629-794) and lcgs (lcg.cpp:437-612) exist for A != A^T: here they meet the oracle on such
    matrices -- three column patterns, both stop rules, plain and packed A.x -- and their first six iterates at 1e-13."""
    from liblcg_amd import _lib
    from oracle import pyoracle as po
    lib = _lib.load()
    A = api.CsrMatrix.generate(n, 16, band, False, seed, 0.01, pattern=pattern)
    rp, ci, v = A.arrays_to_host()
    g = port.gen_init(n, 16, band, False, seed, 0.01, pattern=pattern)
    rp0, ci0, v0 = port.gen_rows(g)
    assert np.array_equal(rp, rp0) and np.array_equal(ci, ci0) and np.array_equal(v, v0)       # same matrix on both sides
    import scipy.sparse as sp
    M = sp.csr_matrix((v, ci, rp), shape=(n, n))
    assert abs(M - M.T).max() > 0.1                                                              # and it is not symmetric
    xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, seed, 0, n, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    bh = b.cpu().numpy()
    for sid in (api.LCG_BICGSTAB, api.LCG_CGS):
        tol, band_it = NONSYM_BANDS[sid]
        for eps, ad, cap, loose in ((1e-12, 1, 0, 1.0), (1e-14, 0, 0, 300.0), (1e-300, 1, 6, None)):
            ref = port.solve(sid, rp, ci, v, bh, para=po.default_para(epsilon=eps, abs_diff=ad, max_iterations=cap))
            for packed in (0, 1):
                assert lib.lcg_hip_csr_set_packed(A.h, packed) == 0
                m = torch.zeros(n, dtype=torch.float64, device="cuda")
                info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=eps, abs_diff=ad, max_iterations=cap), A, sid)
                x = m.cpu().numpy()
                tag = (pattern, sid, eps, cap, packed, info.iterations, ref["iters"])
                assert info.ret == ref["ret"], tag
                rel = np.linalg.norm(x - ref["x"]) / np.linalg.norm(ref["x"])
                if cap:         # six iterations: arithmetic parity, nothing to do with where a run stops
                    assert info.iterations == ref["iters"] == cap and rel <= 1e-13, (tag, rel)
                else:
                    assert abs(info.iterations - ref["iters"]) <= max(3, band_it * ref["iters"]), tag
                    assert rel <= tol * loose, (tag, rel)
    A.destroy()
