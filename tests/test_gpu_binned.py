"""-m gpu: the two-pass binned A.x (liblcg_amd/csrc/csr_binned.hip) for scattered columns -- the arbitrary user CSR
of the reference's cudaAx callback (sample8.cu:96-103) at its worst -- and the one-pass tiled A.x (csr_tiled.hip) for
row-random bands, against the oracle's row-by-row product and against the row-block kernels on the same matrix.

Band: the binned product rounds every a*x before adding it and sums a row in column order, the row-block kernels run
four FMA chains per row and a tree: both are within a few ulp of |A||x| per row, so the test bounds
|y - y_oracle| <= 1e-13 * (|A| |x|) row by row (a bound relative to max|y| would let a small row be wrong)."""
import numpy as np
import pytest

from conftest import FUZZ_SEED_OFFSET

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def api():
    from liblcg_amd import api as a
    assert torch.cuda.is_available()
    return a


@pytest.fixture(scope="module")
def lib():
    from liblcg_amd import _lib
    return _lib.load()


def _ragged(rng, n, ncols, max_len, long_rows=()):
    lens = rng.integers(0, max_len + 1, n)
    lens[rng.integers(0, n, n // 10)] = 0
    for r, ln in long_rows:
        lens[r] = ln
    rp = np.zeros(n + 1, np.int32); np.cumsum(lens, out=rp[1:])
    col = rng.integers(0, ncols, rp[-1]).astype(np.int32)
    return rp, col


def _check(api, lib, port, rp, col, val, x, ncols, expect_binned=True, fmt="binned"):
    setter = lib.lcg_hip_csr_set_binned if fmt == "binned" else lib.lcg_hip_csr_set_tiled
    status = lib.lcg_hip_csr_binned_status if fmt == "binned" else lib.lcg_hip_csr_tiled_status
    name = b"k_bin_expand" if fmt == "binned" else b"k_tile_spmv"
    n = len(rp) - 1
    ref = port.csr_matvec(rp, col, val, x)
    bound = port.csr_matvec(rp, col, np.abs(val), np.abs(x))
    A = api.CsrMatrix.from_csr(rp, col, val, n_cols=ncols)
    xd = torch.from_numpy(x).cuda()
    y0 = torch.full((n,), 7.0, dtype=torch.float64, device="cuda"); y1 = y0.clone(); y2 = y0.clone()
    assert lib.lcg_hip_csr_set_binned(A.h, 0) == 0 and lib.lcg_hip_csr_set_tiled(A.h, 0) == 0
    A.spmv(xd, y0); api.synchronize()
    assert b"k_spmv" in lib.lcg_hip_csr_last_kernel(A.h)
    assert setter(A.h, 1) == 0
    A.spmv(xd, y1); A.spmv(xd, y2); api.synchronize()
    assert (name in lib.lcg_hip_csr_last_kernel(A.h)) == expect_binned, status(A.h)
    assert (status(A.h) == b"ready") == expect_binned
    assert torch.equal(y1, y2)                                      # call to call: same bits
    for y in (y0, y1):
        assert np.all(np.abs(y.cpu().numpy() - ref) <= 1e-13 * bound + 1e-300)
    if expect_binned:                                               # plan to plan: same bits
        B = api.CsrMatrix.from_csr(rp, col, val, n_cols=ncols)
        assert lib.lcg_hip_csr_set_binned(B.h, 0) == 0 and lib.lcg_hip_csr_set_tiled(B.h, 0) == 0 and setter(B.h, 1) == 0
        y3 = torch.empty_like(y1)
        B.spmv(xd, y3); api.synchronize()
        assert torch.equal(y1, y3)
        B.destroy()
    A.destroy()
    return y1.cpu().numpy()


def test_ragged_fuzz_against_the_oracle(api, lib, port):
    """Sizes on both sides of every boundary of the format: rows around multiples of the 2048-row chunk, columns
    around multiples of the 8192-column tile, empty rows, a row of several thousand entries (groups of hundreds),
    repeated (row, col) pairs (random columns repeat), rectangular shapes, one-row and one-column matrices."""
    rng = np.random.default_rng(2024 + FUZZ_SEED_OFFSET)
    shapes = [(1, 1, 1), (5, 3, 3), (2047, 8191, 9), (2048, 8192, 9), (2049, 8193, 9), (4096, 70000, 33), (10000, 300, 20),
              (300, 100000, 40), (30011, 2000003, 33), (6500, 50000, 5)]
    for case, (n, ncols, mx) in enumerate(shapes):
        long_rows = [(n // 2, min(5000, 4 * ncols))] if n > 100 else []
        rp, col = _ragged(rng, n, ncols, mx, long_rows)
        if rp[-1] == 0:
            rp, col = np.array([0] + [1] * n, np.int32), np.zeros(1, np.int32)
        val = rng.standard_normal(rp[-1])
        x = rng.standard_normal(ncols)
        _check(api, lib, port, rp, col, val, x, ncols)
        _check(api, lib, port, rp, col, val, x, ncols, fmt="tiled")


def test_tiled_band_shapes(api, lib, port):
    """The tiled product on what it is for: rows that draw their columns inside a band -- chunk (1024 rows), workgroup
    (4096 rows) and tile (4096 columns) boundaries, groups longer than the six prefetched steps (768 entries) and
    shorter than one, an unaligned x, odd group lengths (the padding pair), empty rows and an empty trailing chunk."""
    rng = np.random.default_rng(77)
    for n, band, per_row in ((4096, 300, 5), (4097, 9000, 33), (20000, 2000, 40), (9000, 70000, 3), (1023, 50, 7), (5000, 4096, 64)):
        ncols = n
        lens = rng.integers(0, per_row + 1, n); lens[rng.integers(0, n, n // 20)] = 0
        rp = np.zeros(n + 1, np.int32); np.cumsum(lens, out=rp[1:])
        rows = np.repeat(np.arange(n), lens)
        col = np.clip(rows + rng.integers(-band, band + 1, rp[-1]), 0, ncols - 1).astype(np.int32)
        val = rng.standard_normal(rp[-1])
        x = rng.standard_normal(ncols + 1)[1:]                      # 8-byte aligned only once on the device? (copy is aligned) -- see below
        _check(api, lib, port, rp, col, val, x, ncols, fmt="tiled")
    # an x that is only 8-byte aligned on the device
    n = 10000
    rp = np.arange(0, 8 * n + 1, 8, dtype=np.int32)
    col = np.clip(np.repeat(np.arange(n), 8) + rng.integers(-3000, 3001, 8 * n), 0, n - 1).astype(np.int32)
    val = rng.standard_normal(8 * n); x = rng.standard_normal(n)
    A = api.CsrMatrix.from_csr(rp, col, val)
    assert lib.lcg_hip_csr_set_tiled(A.h, 1) == 0
    buf = torch.zeros(n + 1, dtype=torch.float64, device="cuda"); buf[1:] = torch.from_numpy(x).cuda()
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    A.spmv(buf[1:], y); api.synchronize()
    assert b"k_tile_spmv" in lib.lcg_hip_csr_last_kernel(A.h)
    ref = port.csr_matvec(rp, col, val, x); bound = port.csr_matvec(rp, col, np.abs(val), np.abs(x))
    assert np.all(np.abs(y.cpu().numpy() - ref) <= 1e-13 * bound + 1e-300)
    A.destroy()


def test_padding_never_touches_x_it_does_not_own(api, lib, port):
    """A NaN / inf in x may only reach the rows that reference that column (the padding entries of a group read
    column 0 of their tile and must not contribute)."""
    rng = np.random.default_rng(5)
    n, ncols = 9000, 40000
    rp, col = _ragged(rng, n, ncols, 12)
    val = rng.standard_normal(rp[-1])
    x = rng.standard_normal(ncols)
    for bad in (0, 4096, 8192, 16384, 39999):
        for setter in (lib.lcg_hip_csr_set_binned, lib.lcg_hip_csr_set_tiled):
            x2 = x.copy(); x2[bad] = np.nan
            A = api.CsrMatrix.from_csr(rp, col, val, n_cols=ncols)
            assert lib.lcg_hip_csr_set_binned(A.h, 0) == 0 and lib.lcg_hip_csr_set_tiled(A.h, 0) == 0 and setter(A.h, 1) == 0
            y = torch.empty(n, dtype=torch.float64, device="cuda")
            A.spmv(torch.from_numpy(x2).cuda(), y); api.synchronize()
            assert b"k_spmv" not in lib.lcg_hip_csr_last_kernel(A.h)
            touched = np.zeros(n, bool)
            rows = np.repeat(np.arange(n), np.diff(rp))
            touched[rows[col == bad]] = True
            assert np.array_equal(np.isnan(y.cpu().numpy()), touched), bad
            A.destroy()


def test_dense_groups_and_out_of_range_columns(api, lib, port):
    """Every entry of a 3000-row matrix in ONE column tile (groups of 8192 entries, heavy row collisions inside an LDS
    instruction) is still summed correctly; a column index outside [0, n_cols) makes the plan refuse (the row-block
    kernel answers, as before)."""
    rng = np.random.default_rng(6)
    n = 3000
    rp = np.arange(0, 4 * n + 1, 4, dtype=np.int32)
    col = rng.integers(0, 500, rp[-1]).astype(np.int32)
    val = rng.standard_normal(rp[-1]); x = rng.standard_normal(9000)
    _check(api, lib, port, rp, col, val, x, 9000)
    A = api.CsrMatrix.from_csr(rp, col, val, n_cols=400)           # lies about the width: columns up to 499
    assert lib.lcg_hip_csr_set_binned(A.h, 1) == 0
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    A.spmv(torch.from_numpy(x).cuda(), y); api.synchronize()
    assert b"outside" in lib.lcg_hip_csr_binned_status(A.h) and b"k_bin" not in lib.lcg_hip_csr_last_kernel(A.h)
    assert np.abs(y.cpu().numpy() - port.csr_matvec(rp, col, val, x)).max() <= 1e-12 * np.abs(x).max() * 10


def test_automatic_choice_and_solvers(api, lib, port):
    """Automatic mode takes the binned product for a large scattered system and leaves a banded one alone; CG, PCG,
    CGS and BiCGStab on a scattered system through the binned product meet the oracle's solutions."""
    from oracle import pyoracle as po
    n = 1_500_000
    # (row-random bands: the tiled product from a line ratio of 0.18 on -- bands of >= 8192 columns at 33 entries per row; narrower ones
    #  keep the packed row blocks, whose gathers then share their cache lines: profiles/r03_band_sweep.txt; below 4M rows the caches help the
        #  row blocks and the tiled product starts at a line ratio of 0.3, the band of 16384 columns: profiles/r04_choice_regret.txt)
    for pattern, band, want in ((api.GEN_SCRAMBLED, 0, b"k_bin_expand"), (api.GEN_DIAGONALS, 40000, b"k_spmv_ldsp"),
                                (api.GEN_ROW_RANDOM_BAND, 40000, b"k_tile_spmv"), (api.GEN_ROW_RANDOM_BAND, 16384, b"k_tile_spmv"),
                                (api.GEN_ROW_RANDOM_BAND, 8192, b"k_spmv_ldsp"), (api.GEN_ROW_RANDOM_BAND, 2048, b"k_spmv_ldsp")):
        A = api.CsrMatrix.generate(n, 16, band, True, 3, 0.01, pattern=pattern)
        x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 3, 0, n, x)
        y = torch.empty_like(x); y2 = torch.empty_like(x)
        A.spmv(x, y); api.synchronize()
        assert want in lib.lcg_hip_csr_last_kernel(A.h), (pattern, lib.lcg_hip_csr_last_kernel(A.h), lib.lcg_hip_csr_binned_status(A.h), lib.lcg_hip_csr_tiled_status(A.h))
        if want != b"k_spmv_ldsp":
            # what the kernels stream by construction: binned 28.5 B per entry; tiled 2 KB per step of 192 entries (three
            # 21-bit (row, column) pairs per 64-bit word: 10.67 B per entry) + x, y and the tile lists
            model = lib.lcg_hip_csr_last_traffic_model(A.h)
            assert (12 if want == b"k_bin_expand" else 10.66) * A.nnz <= model <= 40 * A.nnz, (want, model / A.nnz)
            assert lib.lcg_hip_csr_set_binned(A.h, 0) == 0 and lib.lcg_hip_csr_set_tiled(A.h, 0) == 0
            A.spmv(x, y2); api.synchronize()
            assert b"k_spmv" in lib.lcg_hip_csr_last_kernel(A.h)
            assert ((y - y2).abs().max() <= 1e-13 * y2.abs().max()).item()
        A.destroy()
    n = 50_000
    for sym, sids in ((True, (api.LCG_CG, api.LCG_PCG, api.LCG_CGS)), (False, (api.LCG_BICGSTAB, api.LCG_CGS))):
        A = api.CsrMatrix.generate(n, 16, 0, sym, 9, 0.01)
        A.build_jacobi()
        assert lib.lcg_hip_csr_set_binned(A.h, 1) == 0
        rp, ci, v = A.arrays_to_host()
        xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 9, 0, n, xt)
        b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
        assert b"k_bin_expand" in lib.lcg_hip_csr_last_kernel(A.h)
        bh = b.cpu().numpy()
        for sid in sids:
            ref = port.solve(sid, rp, ci, v, bh, para=po.default_para(epsilon=1e-12, abs_diff=1), jacobi=(sid == api.LCG_PCG))
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            para = api.lcg_default_parameters(epsilon=1e-12, abs_diff=1)
            if sid == api.LCG_PCG:
                info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, para, A)
            else:
                info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, sid)
            xs = m.cpu().numpy()
            assert info.ret == ref["ret"] == 0, (sym, sid)
            assert abs(info.iterations - ref["iters"]) <= 3, (sym, sid, info.iterations, ref["iters"])
            assert np.linalg.norm(xs - ref["x"]) <= 1e-9 * np.linalg.norm(ref["x"]), (sym, sid)
        A.destroy()


def test_tiled_product_carries_the_dot():
    """k_tile_spmv2<DOT>: every consumer wavefront leaves its 1024 rows' share of y.u (and y.y); y must not change by a bit against the
    plain tiled product, the sums must agree with numpy's on the same y, and CG through it must meet the oracle-checked plain run."""
    import ctypes as C
    from liblcg_amd import _lib, api
    lib = _lib.load()
    for n, band in ((20000, 3000), (300000, 40000), (5 * 8192 + 77, 9000)):
        A = api.CsrMatrix.generate(n, 16, band, True, 4, 0.01, pattern=api.GEN_ROW_RANDOM_BAND)
        assert lib.lcg_hip_csr_set_tiled(A.h, 1) == 0
        x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 9, 0, n, x)
        u = torch.rand(n, dtype=torch.float64, device="cuda") - 0.5
        y0 = torch.empty_like(x); y1 = torch.full_like(x, 3.0)
        A.spmv(x, y0); api.synchronize()
        assert "k_tile_spmv" in lib.lcg_hip_csr_last_kernel(A.h).decode()
        sums = (C.c_double * 2)()
        assert lib.lcg_hip_spmv_dot(A.h, x.data_ptr(), y1.data_ptr(), u.data_ptr(), sums) == 0
        name = lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert "k_tile_spmv" in name and "carrying the dot" in name, name
        assert torch.equal(y0, y1), n
        yh, uh = y0.cpu().numpy(), u.cpu().numpy()
        assert abs(sums[0] - float(yh @ uh)) <= 1e-12 * float(np.abs(yh) @ np.abs(uh)) and abs(sums[1] - float(yh @ yh)) <= 1e-12 * float(yh @ yh), n
        # the same bits from call to call
        s2 = (C.c_double * 2)()
        assert lib.lcg_hip_spmv_dot(A.h, x.data_ptr(), y1.data_ptr(), u.data_ptr(), s2) == 0 and s2[0] == sums[0] and s2[1] == sums[1]
        # CG with and without the carried dot: same iteration count, same answer to rounding
        b = torch.empty_like(x); A.spmv(x, b); api.synchronize()
        m1 = torch.zeros_like(x)
        i1 = api.lcg_solver("lcg_hip_csr_ax", None, m1, b, n, api.lcg_default_parameters(epsilon=1e-12, abs_diff=1), A, api.LCG_CG)
        api.set_cg_schedule(api.CG_CLASSIC)
        try:
            m2 = torch.zeros_like(x)
            i2 = api.lcg_solver("lcg_hip_csr_ax", None, m2, b, n, api.lcg_default_parameters(epsilon=1e-12, abs_diff=1), A, api.LCG_CG)
        finally:
            api.set_cg_schedule(api.CG_AUTO)
        assert i1.ret == i2.ret == 0 and abs(i1.iterations - i2.iterations) <= 3
        assert ((m1 - x).norm() / x.norm()).item() <= 1e-8 and ((m2 - x).norm() / x.norm()).item() <= 1e-8
        A.destroy()


def test_wide_bands_at_full_size_take_the_tiled_product(api, lib):
    """The rule scripts/choice_regret.py moved (profiles/r04_choice_regret.txt): at 10M rows a band of +-524288 columns drawn per row -- a mean
    block span of 2^20, "scattered" by round 3's rule, which cut the matrix into ranges with a binned middle (1828 us) -- is ONE tiled
    product (1027 us); at +-2097152 the binned product has long taken over (1599 against 3115 us; around +-1048576 the two are within 5 % of
    each other and either may be chosen); y against the plain row-block kernel's."""
    n = 10_000_000
    for band, want in ((524288, b"k_tile_spmv"), (2097152, b"k_bin_expand")):
        A = api.CsrMatrix.generate(n, 16, band, True, 1, 0.01, pattern=api.GEN_ROW_RANDOM_BAND)
        x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, x)
        y = torch.empty_like(x); y2 = torch.empty_like(x)
        A.spmv(x, y); api.synchronize()
        kern = lib.lcg_hip_csr_last_kernel(A.h)
        assert kern.startswith(want), (band, kern)
        assert lib.lcg_hip_csr_set_tiled(A.h, 0) == 0 and lib.lcg_hip_csr_set_binned(A.h, 0) == 0 and lib.lcg_hip_csr_set_ranges(A.h, 0) == 0
        A.spmv(x, y2); api.synchronize()
        assert b"k_spmv" in lib.lcg_hip_csr_last_kernel(A.h)
        assert ((y - y2).abs().max() <= 1e-13 * y2.abs().max()).item(), band
        A.destroy()
        del x, y, y2
        torch.cuda.empty_cache()


def test_choice_rules_below_four_million_rows(api, lib):
    """The rules scripts/choice_regret.py moved in round 5 (profiles/r05_choice_regret.txt; 12-38 % lost before): (1) at 1M rows a band as
    wide as the matrix is no band -- x is 8 MB, the packed row blocks gather from the caches (170 us) where the tiled product pays for
    every tile (231); (2) the tiled product takes the band of 8192 columns from ~1.5M rows on (2M rows: 150 against 198 us), not only
    from 4M; (3) 800K rows of constant diagonals + 200K scattered rows are ONE packed product (102 us), not two ranges (116).
    Every choice against the plain row-block kernel's y."""
    import ctypes as C

    def check(A, n, want, never=()):
        x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, x)
        y = torch.empty_like(x); y2 = torch.empty_like(x)
        A.spmv(x, y); api.synchronize()
        kern = lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert want in kern and not any(w in kern for w in never), kern
        assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0 and lib.lcg_hip_csr_set_tiled(A.h, 0) == 0 and lib.lcg_hip_csr_set_binned(A.h, 0) == 0 and lib.lcg_hip_csr_set_ranges(A.h, 0) == 0
        A.spmv(x, y2); api.synchronize()
        assert ((y - y2).abs().max() <= 1e-13 * y2.abs().max()).item(), kern
        A.destroy()

    n = 1_000_000
    check(api.CsrMatrix.generate(n, 16, 524288, True, 1, 0.01, pattern=api.GEN_ROW_RANDOM_BAND), n, "k_spmv_ldsp", never=("k_tile", "k_bin", "rows ["))
    check(api.CsrMatrix.generate(n, 16, 131072, True, 1, 0.01, pattern=api.GEN_ROW_RANDOM_BAND), n, "k_tile_spmv")
    check(api.CsrMatrix.generate(n, 16, 8192, True, 1, 0.01, pattern=api.GEN_ROW_RANDOM_BAND), n, "k_spmv_ldsp", never=("k_tile",))
    n = 2_000_000
    check(api.CsrMatrix.generate(n, 16, 8192, True, 1, 0.01, pattern=api.GEN_ROW_RANDOM_BAND), n, "k_tile_spmv")
    check(api.CsrMatrix.generate(n, 16, 524288, True, 1, 0.01, pattern=api.GEN_ROW_RANDOM_BAND), n, "k_tile_spmv")
    # 800K rows of constant diagonals above 200K rows with scattered columns (inside their own block)
    n, n1 = 1_000_000, 800_000
    parts = []
    for rows, band, pat, seed in ((n1, 131072, api.GEN_DIAGONALS, 3), (n - n1, 0, api.GEN_SCRAMBLED, 5)):
        G = api.CsrMatrix.generate(rows, 16, band, True, seed, 0.01, pattern=pat)
        pr, pc, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        assert lib.lcg_hip_csr_arrays(G.h, C.byref(pr), C.byref(pc), C.byref(pv)) == 0
        nz = G.nnz
        rp = torch.empty(rows + 1, dtype=torch.int32, device="cuda"); ci = torch.empty(nz, dtype=torch.int32, device="cuda")
        vv = torch.empty(nz, dtype=torch.float64, device="cuda")
        for dst, src in ((rp, pr), (ci, pc), (vv, pv)):
            assert lib.lcg_hip_memcpy(dst.data_ptr(), src, dst.numel() * dst.element_size(), 3) == 0
        G.destroy()
        parts.append((rp, ci, vv))
    (rp1, c1, v1), (rp2, c2, v2) = parts
    M = api.CsrMatrix.from_csr(torch.cat([rp1, rp2[1:] + int(rp1[-1].item())]), torch.cat([c1, c2 + n1]), torch.cat([v1, v2]), n_cols=n)
    check(M, n, "k_spmv_ldsp", never=("rows [",))
