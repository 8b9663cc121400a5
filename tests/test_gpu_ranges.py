"""-m gpu: row ranges (csr_choice.hip, "row ranges"; lcg_hip_csr_set_ranges) -- matrices whose rows fall into different column-pattern
classes: a 27-point stencil in 90 % of the rows, long-range random couplings in the rest.

  * <= 60,000 rows, built on the host, ranges forced: A.x row by row against the oracle's product (1e-13 |A||x|), the cut where
    the generator put it, CG (block-diagonal SPD mix) and BiCGStab / CGS (non-symmetric mix, couplings anywhere) against the
    oracle with the bands of tests/test_gpu_fuzz_solvers.py and six capped iterations to 1e-13;
  * 10M rows, composed on the device (8M rows of constant diagonals + 2M rows of scrambled columns, block diagonal), automatic
    mode: two ranges, the run-block kernel on the first and the binned product on the second -- asserted --, and y BIT-EQUAL to
    the products of the two blocks as matrices of their own (a range is multiplied by the kernel it would get alone).
The A.x contract is the user's callback of the reference (lcg.h:37-38): any CSR must be multiplied correctly whatever mix of
patterns it holds."""
import ctypes as C

import numpy as np
import pytest

from conftest import FUZZ_SEED_OFFSET, check_converged_run

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def stencil27(nx, ny, nz):
    """CSR of a 27-point stencil on an nx x ny x nz grid (row-major, x fastest): -1 off the diagonal, 27 on it (SPD)."""
    n = nx * ny * nz
    idx = np.arange(n).reshape(nz, ny, nx)
    rows, cols, vals = [], [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                src = idx[max(0, -dz):nz - max(0, dz), max(0, -dy):ny - max(0, dy), max(0, -dx):nx - max(0, dx)]
                dst = idx[max(0, dz):nz - max(0, -dz), max(0, dy):ny - max(0, -dy), max(0, dx):nx - max(0, -dx)]
                rows.append(src.ravel()); cols.append(dst.ravel())
                vals.append(np.full(src.size, 27.0 if (dx, dy, dz) == (0, 0, 0) else -1.0))
    return n, np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)


def to_csr(n, r, c, v):
    order = np.lexsort((c, r))
    r, c, v = r[order], c[order], v[order]
    rp = np.zeros(n + 1, np.int64); np.add.at(rp, r + 1, 1)
    return np.cumsum(rp).astype(np.int32), c.astype(np.int32), v.astype(np.float64)


def mixed_system(rng, symmetric, dims=None):
    """90 % stencil rows followed by 10 % rows with 24 long-range couplings each.  symmetric: the couplings stay inside the
    last block and are mirrored (block-diagonal SPD); otherwise they go anywhere in the matrix (A != A^T)."""
    nx, ny, nz = dims or (int(rng.integers(24, 40)), int(rng.integers(24, 40)), int(rng.integers(20, 34)))
    n1, r, c, v = stencil27(nx, ny, nz)
    n2 = max(2048, (n1 // 9) // 2048 * 2048 + 2048)
    n1p = (n1 + 2047) // 2048 * 2048          # the stencil block padded with diagonal-only rows up to a chunk boundary
    pad = np.arange(n1, n1p)
    n = n1p + n2
    rr = np.repeat(np.arange(n1p, n), 24)
    if symmetric:
        cc = rng.integers(n1p, n, rr.size)
        keep = cc != rr
        rr, cc = rr[keep], cc[keep]
        w = -rng.random(rr.size)
        R = np.concatenate([rr, cc]); Cc = np.concatenate([cc, rr]); W = np.concatenate([w, w])
    else:
        cc = rng.integers(0, n, rr.size)
        keep = cc != rr
        R, Cc, W = rr[keep], cc[keep], -rng.random(int(keep.sum()))
    # duplicates are summed into one entry; the diagonal of the coupled rows dominates their row
    key = R.astype(np.int64) * n + Cc
    uk, inv = np.unique(key, return_inverse=True)
    Wk = np.zeros(uk.size); np.add.at(Wk, inv, W)
    R, Cc = (uk // n).astype(np.int64), (uk % n).astype(np.int64)
    dsum = np.zeros(n); np.add.at(dsum, R, np.abs(Wk))
    drows = np.concatenate([pad, np.arange(n1p, n)])
    r = np.concatenate([r, R, drows]); c = np.concatenate([c, Cc, drows])
    v = np.concatenate([v, Wk, np.concatenate([np.full(pad.size, 1.0), dsum[n1p:] + 0.5])])
    return n, n1p, to_csr(n, r, c, v)


def test_mixed_rows_against_the_oracle(port):
    from liblcg_amd import _lib, api
    from oracle import pyoracle as po
    lib = _lib.load()
    rng = np.random.default_rng(2026 + FUZZ_SEED_OFFSET)
    for case in range(4):
        symmetric = case % 2 == 0
        n, cut, (rp, ci, v) = mixed_system(rng, symmetric)
        assert n <= 60000
        A = api.CsrMatrix.from_csr(rp, ci, v)
        assert lib.lcg_hip_csr_set_ranges(A.h, 1) == 0
        xh = rng.standard_normal(n)
        x = torch.from_numpy(xh).cuda(); y = torch.empty_like(x)
        A.spmv(x, y); api.synchronize()
        first = (C.c_int * 8)()
        nr = lib.lcg_hip_csr_ranges(A.h, 8, first)
        name = lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert nr == 2 and list(first[:2]) == [0, cut], (case, nr, list(first[:nr]), cut, name)
        assert name.startswith("rows [0, %d): " % cut) and (" | rows [%d, %d): " % (cut, n)) in name, name
        ref = port.csr_matvec(rp, ci, v, xh)
        bound = port.csr_matvec(rp, ci, np.abs(v), np.abs(xh))
        assert float(np.max(np.abs(y.cpu().numpy() - ref) / bound)) <= 1e-13, (case, name)
        # one kernel family for all rows gives the same product to rounding
        assert lib.lcg_hip_csr_set_ranges(A.h, 0) == 0
        y0 = torch.empty_like(x); A.spmv(x, y0); api.synchronize()
        assert lib.lcg_hip_csr_ranges(A.h, 8, first) == 0
        assert float(np.max(np.abs(y0.cpu().numpy() - ref) / bound)) <= 1e-13, case
        assert lib.lcg_hip_csr_set_ranges(A.h, 1) == 0
        # solvers through the split product
        xt = torch.from_numpy(rng.standard_normal(n)).cuda()
        b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
        bh = b.cpu().numpy()
        eps, abs_diff = 1e-12, case // 2
        for sid, nm in (((api.LCG_CG, "cg"),) if symmetric else ((api.LCG_BICGSTAB, "bicgstab"), (api.LCG_CGS, "cgs"))):
            def solve_gpu(cap, sid=sid):
                m = torch.zeros(n, dtype=torch.float64, device="cuda")
                info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=eps, abs_diff=abs_diff, max_iterations=cap), A, sid)
                assert lib.lcg_hip_csr_ranges(A.h, 8, first) == 2
                return info.ret, info.iterations, info.residual, m.cpu().numpy()
            tag = (case, n, nm)
            check_converged_run(port, solve_gpu, sid, rp, ci, v, bh, eps, abs_diff, tag=tag, wide=not symmetric, xt=xt.cpu().numpy())
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            m.zero_()
            i6 = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=eps, abs_diff=abs_diff, max_iterations=6), A, sid)
            r6 = port.solve(sid, rp, ci, v, bh, para=po.default_para(epsilon=eps, abs_diff=abs_diff, max_iterations=6))
            assert i6.ret == r6["ret"] == -1019 and i6.iterations == 6, tag
            assert np.linalg.norm(m.cpu().numpy() - r6["x"]) <= 1e-13 * np.linalg.norm(r6["x"]), tag
        A.destroy()


@pytest.mark.parametrize("grid,cuts", [((64, 64, 64), False), ((128, 128, 128), True)])
def test_automatic_mode_cuts_only_where_a_cut_pays(port, grid, cuts):
    """Stencil rows + 10 % coupled rows, automatic mode.  64^3 (7.6M entries, 293K columns): every stretch would end in row-block
    kernels and the whole matrix can be packed -- ONE product (49 us) beats one per stretch (60 us: scripts/ranges_ab.py), so nothing is cut
    (round 5; round 4 cut here).  128^3 (61M entries, 2.3M columns): the coupled rows' blocks span 2^21 columns and more, which costs the
    WHOLE matrix its packed columns (243 us) -- cut, the stencil rows get them back (220 us).  A.x against the oracle's product."""
    from liblcg_amd import _lib, api
    lib = _lib.load()
    rng = np.random.default_rng(64 + FUZZ_SEED_OFFSET)
    n, cut, (rp, ci, v) = mixed_system(rng, False, grid)
    assert len(ci) >= 4_000_000
    A = api.CsrMatrix.from_csr(rp, ci, v)
    xh = rng.standard_normal(n)
    x = torch.from_numpy(xh).cuda(); y = torch.empty_like(x)
    A.spmv(x, y); api.synchronize()
    first = (C.c_int * 8)()
    nr = lib.lcg_hip_csr_ranges(A.h, 8, first)
    name = lib.lcg_hip_csr_last_kernel(A.h).decode()
    if cuts:
        assert nr == 2 and list(first[:2]) == [0, cut], (nr, list(first[:nr]), cut, name)
        assert "rows [0, %d): k_spmv_ldsp (LDS-staged" % cut in name, name
    else:
        # (a 64-row block of this grid is one x-line with its two boundary rows: template blocks; the coupled rows ride along as packed columns)
        assert nr == 0 and name.startswith("k_spmv_ldsp (LDS-staged") and "packed columns" in name, (nr, name)
    ref = port.csr_matvec(rp, ci, v, xh)
    bound = port.csr_matvec(rp, ci, np.abs(v), np.abs(xh))
    assert float(np.max(np.abs(y.cpu().numpy() - ref) / bound)) <= 1e-13, name
    A.destroy()


def device_arrays(A):
    """(rowptr, col, val) of a handle as torch tensors on the device (copies)."""
    from liblcg_amd import _lib
    lib = _lib.load()
    pr, pc, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.lcg_hip_csr_arrays(A.h, C.byref(pr), C.byref(pc), C.byref(pv)) == 0
    nnz = A.nnz
    rp = torch.empty(A.n + 1, dtype=torch.int32, device="cuda"); ci = torch.empty(nnz, dtype=torch.int32, device="cuda")
    v = torch.empty(nnz, dtype=torch.float64, device="cuda")
    for dst, src in ((rp, pr), (ci, pc), (v, pv)):
        assert lib.lcg_hip_memcpy(dst.data_ptr(), src, dst.numel() * dst.element_size(), 3) == 0
    return rp, ci, v


def test_description_does_not_outlive_the_range_plan(port):
    """the range-by-range product's description was a pointer into the plan; dropping the plan (set_ranges, set_packed,
    set_binned, set_tiled all do) and asking lcg_hip_csr_last_kernel / _last_traffic_model BEFORE the next product read freed
    memory.  Now the description is reset with the plan."""
    from liblcg_amd import _lib, api
    lib = _lib.load()
    rng = np.random.default_rng(77)
    n, n1p, (rp, ci, v) = mixed_system(rng, True, dims=(28, 26, 22))
    A = api.CsrMatrix.from_csr(rp, ci, v)
    assert lib.lcg_hip_csr_set_ranges(A.h, 1) == 0
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    A.spmv(x, y); api.synchronize()
    first = (C.c_int * 8)()
    assert lib.lcg_hip_csr_ranges(A.h, 8, first) >= 2
    split = lib.lcg_hip_csr_last_kernel(A.h).decode()
    assert split.startswith("rows [0, ") and " | " in split
    assert lib.lcg_hip_csr_last_traffic_model(A.h) > 0
    for drop in (lambda: lib.lcg_hip_csr_set_ranges(A.h, 0), lambda: lib.lcg_hip_csr_set_packed(A.h, 0),
                 lambda: lib.lcg_hip_csr_set_binned(A.h, 0), lambda: lib.lcg_hip_csr_set_tiled(A.h, 0)):
        assert lib.lcg_hip_csr_set_ranges(A.h, 1) == 0
        A.spmv(x, y); api.synchronize()
        assert " | " in lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert drop() == 0
        junk = [torch.full((4096,), 3.0, device="cuda") for _ in range(64)]       # whatever was freed is likely overwritten by now
        assert lib.lcg_hip_csr_last_kernel(A.h).decode() == ""                  # no product since the plan went
        assert lib.lcg_hip_csr_last_traffic_model(A.h) == 0
        del junk
    assert lib.lcg_hip_csr_set_ranges(A.h, 0) == 0
    A.spmv(x, y); api.synchronize()
    assert " | " not in lib.lcg_hip_csr_last_kernel(A.h).decode() and lib.lcg_hip_csr_last_kernel(A.h).decode() != ""
    yo = port.csr_matvec(rp, ci, v, x.cpu().numpy())
    assert np.max(np.abs(y.cpu().numpy() - yo)) <= 1e-12 * np.max(np.abs(yo))
    A.destroy()


def test_ten_million_rows_two_classes():
    from liblcg_amd import _lib, api
    lib = _lib.load()
    n1, n2 = 8_000_000, 2_000_000
    n = n1 + n2
    G1 = api.CsrMatrix.generate(n1, 16, 131072, True, 3, 0.01, pattern=api.GEN_DIAGONALS)
    G2 = api.CsrMatrix.generate(n2, 16, 0, True, 5, 0.01, pattern=api.GEN_SCRAMBLED)
    rp1, c1, v1 = device_arrays(G1)
    rp2, c2, v2 = device_arrays(G2)
    nnz1 = int(rp1[-1].item())
    rp = torch.cat([rp1, rp2[1:] + nnz1]); ci = torch.cat([c1, c2 + n1]); v = torch.cat([v1, v2])
    # the scattered block as a matrix of its own WITH the columns it has in the mix (2M x 10M): the binned product sums a row in
    # an order that depends on where its columns fall in the 8192-column tiles
    G2.destroy()
    G2 = api.CsrMatrix.from_csr(rp2, c2 + n1, v2, n_cols=n)
    del rp1, c1, v1, rp2, c2, v2
    A = api.CsrMatrix.from_csr(rp, ci, v, n_cols=n)
    x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 11, 0, n, x)
    y = torch.empty_like(x)
    A.spmv(x, y); api.synchronize()
    first = (C.c_int * 8)()
    nr = lib.lcg_hip_csr_ranges(A.h, 8, first)
    name = lib.lcg_hip_csr_last_kernel(A.h).decode()
    assert nr == 2 and list(first[:2]) == [0, n1], (nr, list(first[:nr]), name)
    a, b = name.split(" | ")
    assert a.startswith("rows [0, %d): k_spmv_ldsp (LDS-staged, run blocks" % n1), name
    assert b.startswith("rows [%d, %d): k_bin_expand + k_bin_reduce" % (n1, n)), name
    # each block as a matrix of its own: the same kernels, the same bits
    y1 = torch.empty(n1, dtype=torch.float64, device="cuda"); y2 = torch.empty(n2, dtype=torch.float64, device="cuda")
    G1.spmv(x[:n1].contiguous(), y1); G2.spmv(x, y2); api.synchronize()
    assert lib.lcg_hip_csr_last_kernel(G1.h).decode().startswith("k_spmv_ldsp (LDS-staged, run blocks")
    assert lib.lcg_hip_csr_last_kernel(G2.h).decode().startswith("k_bin_expand")
    assert torch.equal(y[:n1], y1) and torch.equal(y[n1:], y2)
    # and against one kernel family for all rows (what the library did before): rounding apart, and slower
    import time

    def timed(out, reps=10):
        A.spmv(x, out); api.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            A.spmv(x, out)
        api.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    t_split = timed(y)
    assert lib.lcg_hip_csr_set_ranges(A.h, 0) == 0
    y0 = torch.empty_like(x)
    t_whole = timed(y0)
    whole = lib.lcg_hip_csr_last_kernel(A.h).decode()
    assert lib.lcg_hip_csr_ranges(A.h, 8, first) == 0
    print(f"\n10M mixed rows: ranges {t_split:.0f} us ({name}) vs one family {t_whole:.0f} us ({whole})")
    scale = float(torch.max(torch.abs(y)).item())
    assert float(torch.max(torch.abs(y0 - y)).item()) <= 1e-11 * scale
    assert t_split < t_whole
    for M in (A, G1, G2):
        M.destroy()


def test_random_class_layouts(port):
    """Fuzz: 2-5 stretches of rows, each of a random class -- tridiagonal / 9-point / 27-point stencil pieces (structured), columns
    drawn per row inside a band (banded random), columns anywhere (scattered), empty rows in between, stretch heights that are
    NOT multiples of the 64-row block or the 2048-row chunk -- forced ranges: the product row by row against the oracle's
    (1e-13 |A||x|), wherever the cuts fall, and equal to rounding with the split switched off.  The cuts themselves are a
    matter of speed, not of correctness: only their sanity is asserted (ascending, inside the matrix)."""
    from liblcg_amd import _lib, api
    lib = _lib.load()
    rng = np.random.default_rng(777 + FUZZ_SEED_OFFSET)
    seen_ranges = 0
    for case in range(12):
        nstretch = int(rng.integers(2, 6))
        heights = [int(rng.integers(1500, 30000)) for _ in range(nstretch)]
        n = sum(heights)
        rows, cols, vals = [], [], []
        r0 = 0
        for h in heights:
            kind = int(rng.integers(0, 4))
            rr = np.arange(r0, r0 + h)
            if kind == 0:       # structured: a few constant diagonals
                offs = np.unique(np.concatenate([[0], rng.integers(-min(2000, n - 1), min(2000, n - 1), int(rng.integers(3, 12)))]))
                for o in offs:
                    cc = rr + o
                    ok = (cc >= 0) & (cc < n)
                    rows.append(rr[ok]); cols.append(cc[ok]); vals.append(rng.standard_normal(int(ok.sum())))
            elif kind == 1:     # banded random: every row draws its own columns within +-W
                W = int(rng.integers(200, 5000)); k = int(rng.integers(4, 20))
                cc = np.clip(rr[:, None] + rng.integers(-W, W + 1, (h, k)), 0, n - 1)
                rows.append(np.repeat(rr, k)); cols.append(cc.ravel()); vals.append(rng.standard_normal(h * k))
            elif kind == 2:     # scattered
                k = int(rng.integers(2, 16))
                rows.append(np.repeat(rr, k)); cols.append(rng.integers(0, n, h * k)); vals.append(rng.standard_normal(h * k))
            else:               # mostly empty rows
                keep = rr[rng.random(h) < 0.05]
                rows.append(keep); cols.append(rng.integers(0, n, keep.size)); vals.append(rng.standard_normal(keep.size))
            r0 += h
        r = np.concatenate(rows); c = np.concatenate(cols); v = np.concatenate(vals)
        key = r.astype(np.int64) * n + c
        uk, inv = np.unique(key, return_inverse=True)
        vk = np.zeros(uk.size); np.add.at(vk, inv, v)
        rp, ci, vv = to_csr(n, (uk // n), (uk % n), vk)
        A = api.CsrMatrix.from_csr(rp, ci, vv)
        xh = rng.standard_normal(n)
        x = torch.from_numpy(xh).cuda(); y = torch.empty_like(x)
        ref = port.csr_matvec(rp, ci, vv, xh)
        bound = port.csr_matvec(rp, ci, np.abs(vv), np.abs(xh)) + 1e-300
        for mode in (1, 0, -1):
            assert lib.lcg_hip_csr_set_ranges(A.h, mode) == 0
            y.fill_(float("nan"))
            A.spmv(x, y); api.synchronize()
            first = (C.c_int * 8)()
            nr = lib.lcg_hip_csr_ranges(A.h, 8, first)
            name = lib.lcg_hip_csr_last_kernel(A.h).decode()
            assert float(np.max(np.abs(y.cpu().numpy() - ref) / bound)) <= 1e-13, (case, mode, name)
            if mode == 1:
                seen_ranges += nr
                cuts = list(first[:nr])
                assert nr == 0 or (2 <= nr <= 8 and cuts[0] == 0 and cuts == sorted(set(cuts)) and cuts[-1] < n), (case, cuts)
            else:
                assert nr == 0 or mode == -1
        A.destroy()
    assert seen_ranges >= 12        # the split really was exercised


def test_arrow_matrix_dense_rows_get_their_own_range(port):
    """A banded SPD matrix with dense rows and columns (constraint / mean-value rows: 300,000, 120,000 and 60,000 entries in rows of a
    300,000-row system).  One such row makes its 64-row block larger than any LDS window; left in the part, it sends ALL rows to the
    window-by-window kernel and one workgroup walks the dense row alone.  The row ranges cut the blocks that hold them out (class 3) and
    multiply those rows in chunks; the rest keeps the one-window kernels.  A.x and CG against the oracle."""
    import time
    import scipy.sparse as sp
    from liblcg_amd import _lib, api
    from oracle import pyoracle as po
    lib = _lib.load()
    rng = np.random.default_rng(314 + FUZZ_SEED_OFFSET)
    n = 300_000
    offs = np.unique(np.concatenate([[0], rng.integers(1, 3000, 8)]))
    diags = [rng.standard_normal(n - o) * 0.1 for o in offs]
    B = sp.diags(diags, offs, shape=(n, n), format="coo")
    rows = [B.row, B.col[B.row != B.col]]; cols = [B.col, B.row[B.row != B.col]]; vals = [B.data, B.data[B.row != B.col]]
    dense = {n - 1: n - 1, n // 2 + 17: 120_000, 70_001: 60_000}       # row -> number of off-diagonal entries
    for r, cnt in dense.items():
        c = rng.choice(np.setdiff1d(np.arange(n), [r], assume_unique=True), cnt, replace=False) if cnt < n - 1 else np.delete(np.arange(n), r)
        v = rng.standard_normal(len(c)) * 1e-3
        rows += [np.full(len(c), r), c]; cols += [c, np.full(len(c), r)]; vals += [v, v]
    M = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    M.sum_duplicates()
    M = M + sp.diags(np.asarray(abs(M).sum(axis=1)).ravel() + 1.0)         # strictly diagonally dominant: SPD
    M = M.tocsr(); M.sort_indices()
    rp, ci, v = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)
    assert len(ci) >= 4_000_000 and int(np.diff(rp).max()) == n
    A = api.CsrMatrix.from_csr(rp, ci, v)
    xh = rng.standard_normal(n)
    x = torch.from_numpy(xh).cuda(); y = torch.empty_like(x)
    A.spmv(x, y); api.synchronize()
    first = (C.c_int * 8)()
    nr = lib.lcg_hip_csr_ranges(A.h, 8, first)
    name = lib.lcg_hip_csr_last_kernel(A.h).decode()
    assert nr == 6, (nr, list(first[:nr]), name)                    # band | dense | band | dense | band | dense
    parts = name.split(" | ")
    assert [("k_lr_partial" in p) for p in parts] == [False, True, False, True, False, True], name
    assert all("k_spmv_lds" in p and "ldsw" not in p for p in parts[0::2]), name
    for r in dense:                                                 # each dense row inside a long-row range of a few 64-row blocks
        i = max(j for j in range(nr) if first[j] <= r)
        assert "k_lr_partial" in parts[i] and (first[i + 1] if i + 1 < nr else n) - first[i] <= 4096, (r, i, list(first[:nr]))
    ref = port.csr_matvec(rp, ci, v, xh)
    bound = port.csr_matvec(rp, ci, np.abs(v), np.abs(xh))
    assert float(np.max(np.abs(y.cpu().numpy() - ref) / bound)) <= 1e-13, name
    y2 = torch.empty_like(x); A.spmv(x, y2); api.synchronize()
    assert torch.equal(y, y2)                                       # the same bits from call to call
    t0 = time.perf_counter()
    for _ in range(20):
        A.spmv(x, y2)
    api.synchronize()
    per = (time.perf_counter() - t0) / 20
    assert per < 1e-3, per                                          # (the window-by-window kernel alone on the dense row: several ms)
    # CG through the split product against the oracle's loop
    xt = rng.standard_normal(n); bh = port.csr_matvec(rp, ci, v, xt)
    bd = torch.from_numpy(bh).cuda()
    last = {}

    def solve_gpu(cap):
        m = torch.zeros(n, dtype=torch.float64, device="cuda")
        info = api.lcg_solver("lcg_hip_csr_ax", None, m, bd, n, api.lcg_default_parameters(epsilon=1e-12, abs_diff=1, max_iterations=cap), A, api.LCG_CG)
        last[cap] = m.cpu().numpy()
        return info.ret, info.iterations, info.residual, last[cap]
    check_converged_run(port, solve_gpu, po.LCG_CG, rp, ci, v, bh, 1e-12, 1, tag=("arrow",), xt=xt)
    api.set_cg_schedule(api.CG_CLASSIC)         # (the reference's own recurrence: a late iterate at the same count too)
    try:
        # (floor 1e-8: a row of 300,000 entries sums to 1e-13 |A||x| at best -- on either side, and differently -- and through A^-1 that
        #  is some 1e-9 in x: the two sides' iterates lie 2-3e-9 apart from the 20th iteration on, as far as either lies from xt)
        check_converged_run(port, solve_gpu, po.LCG_CG, rp, ci, v, bh, 1e-12, 1, tag=("arrow", "classic"), xt=xt, late=True, floor=1e-8)
    finally:
        api.set_cg_schedule(api.CG_AUTO)
    assert float(np.max(np.abs(last[0] - xt))) <= 1e-6
    A.destroy()


def test_small_system_with_a_dense_row(port):
    """Below 4M entries the ranges are not tried -- unless a 64-row block holds many LDS windows of entries: 60,000 rows of a 5-point-like
    band plus one dense row and column.  Only the dense row's block is cut out; the other rows keep one kernel."""
    import scipy.sparse as sp
    from liblcg_amd import _lib, api
    lib = _lib.load()
    rng = np.random.default_rng(2718 + FUZZ_SEED_OFFSET)
    n = 60_000
    B = sp.diags([rng.standard_normal(n - o) for o in (0, 1, 300)], (0, 1, 300), shape=(n, n), format="coo")
    r = 31_007
    c = np.delete(np.arange(n), r); vv = rng.standard_normal(n - 1) * 1e-2
    M = sp.coo_matrix((np.concatenate([B.data, B.data[B.row != B.col], vv, vv]),
                       (np.concatenate([B.row, B.col[B.row != B.col], np.full(n - 1, r), c]),
                        np.concatenate([B.col, B.row[B.row != B.col], c, np.full(n - 1, r)]))), shape=(n, n)).tocsr()
    M.sum_duplicates(); M.sort_indices()
    rp, ci, v = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)
    assert len(ci) < 1_000_000
    A = api.CsrMatrix.from_csr(rp, ci, v)
    xh = rng.standard_normal(n)
    x = torch.from_numpy(xh).cuda(); y = torch.empty_like(x)
    A.spmv(x, y); api.synchronize()
    first = (C.c_int * 8)()
    nr = lib.lcg_hip_csr_ranges(A.h, 8, first)
    name = lib.lcg_hip_csr_last_kernel(A.h).decode()
    assert nr == 3 and first[1] == (r // 64) * 64 and first[2] == first[1] + 64, (nr, list(first[:nr]), name)
    assert "k_lr_partial" in name.split(" | ")[1] and "k_lr_" not in name.split(" | ")[0] + name.split(" | ")[2], name
    ref = port.csr_matvec(rp, ci, v, xh)
    bound = port.csr_matvec(rp, ci, np.abs(v), np.abs(xh))
    assert float(np.max(np.abs(y.cpu().numpy() - ref) / bound)) <= 1e-13, name
    A.destroy()
    # the same band without the dense row: one kernel, no ranges
    M0 = (B + sp.triu(B, 1).T).tocsr(); M0.sort_indices()
    A0 = api.CsrMatrix.from_csr(M0.indptr.astype(np.int32), M0.indices.astype(np.int32), M0.data.astype(np.float64))
    A0.spmv(x, y); api.synchronize()
    assert lib.lcg_hip_csr_ranges(A0.h, 8, first) == 0, lib.lcg_hip_csr_last_kernel(A0.h).decode()
    A0.destroy()
