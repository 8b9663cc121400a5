"""-m gpu: kernel-level parity through the C ABI -- A.x (every variant, ragged/empty/long rows),
BLAS-1, Jacobi, COO ingest, generators, the sharded product, and size-independent properties
at the full 10M-row benchmark size."""
import ctypes as C
import os

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


def _ragged(rng, n, ncols, max_len, long_rows=()):
    lens = rng.integers(0, max_len + 1, n)
    lens[rng.integers(0, n, n // 10)] = 0                  # empty rows
    for r, ln in long_rows:
        lens[r] = ln
    rp = np.zeros(n + 1, np.int32); np.cumsum(lens, out=rp[1:])
    col = rng.integers(0, ncols, rp[-1]).astype(np.int32)
    return rp, col


@pytest.mark.parametrize("cplx", [False, True])
def test_spmv_all_variants_ragged(api, port, cplx):
    rng = np.random.default_rng(11 + FUZZ_SEED_OFFSET)
    n = 3001
    rp, col = _ragged(rng, n, n, 40, long_rows=[(5, 5000), (2999, 2500), (3000, 7)])
    val = rng.standard_normal(rp[-1]) + (1j * rng.standard_normal(rp[-1]) if cplx else 0)
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    ref = port.csr_matvec(rp, col, val, x)
    A = api.CsrMatrix.from_csr(rp, col, val)
    xd = torch.from_numpy(x).cuda(); yd = torch.empty_like(xd)
    scale = np.abs(ref).max()
    for var in (0, -1, -16, -32, -64, -128, -256, 1, 2, 4, 8, 16, 32, 64):
        yd.zero_()
        A.set_kernel(var)
        A.spmv(xd, yd); api.synchronize()
        assert np.abs(yd.cpu().numpy() - ref).max() <= 1e-12 * scale, var


def test_packed_columns_are_bit_identical(api, port):
    """The packed-column form of the one-window kernel (21-bit block-relative columns) against the plain
    one on the same matrix: same rows per block, same summation order => identical bits.  Ragged rows
    (empty rows, a block that is almost one window), generated banded system, and a system whose blocks
    span more than 2^21 columns (not eligible: the plain kernel must answer)."""
    from liblcg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(21)
    n = 20000
    lens = rng.integers(20, 34, n); lens[rng.integers(0, n, n // 15)] = 0; lens[64:128] = 34
    rp = np.zeros(n + 1, np.int32); np.cumsum(lens, out=rp[1:])
    col = np.concatenate([np.sort(rng.choice(min(n, 5000) if ln else 1, ln, replace=False)) + rng.integers(0, n - 5000) for ln in lens if ln]).astype(np.int32)
    val = rng.standard_normal(rp[-1])
    x = rng.standard_normal(n)
    ref = port.csr_matvec(rp, col, val, x)
    A = api.CsrMatrix.from_csr(rp, col, val)
    xd = torch.from_numpy(x).cuda()
    y0 = torch.empty_like(xd); y1 = torch.empty_like(xd)
    A.set_kernel(-64)
    assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0
    A.spmv(xd, y0); api.synchronize()
    assert lib.lcg_hip_csr_set_packed(A.h, 1) == 0
    A.spmv(xd, y1); api.synchronize()
    assert torch.equal(y0, y1)
    assert np.abs(y1.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
    # generated banded system, automatic R
    for band, eligible in ((3000, True), (0, False)):
        nn = 400000 if band else 3_000_000
        B = api.CsrMatrix.generate(nn, 16, band, True, 5, 0.01)
        xb = torch.empty(nn, dtype=torch.float64, device="cuda"); api.gen_xtrue(nn, 3, 0, nn, xb)
        z0 = torch.empty_like(xb); z1 = torch.empty_like(xb)
        assert lib.lcg_hip_csr_set_packed(B.h, 0) == 0
        B.spmv(xb, z0); api.synchronize()
        assert lib.lcg_hip_csr_set_packed(B.h, 1) == 0
        B.spmv(xb, z1); B.spmv(xb, z1); api.synchronize()
        assert torch.equal(z0, z1), band
        B.destroy()


def test_run_blocks_of_the_packed_form(api, port):
    """Run blocks (csr.hip: k_pk_meta / the run path of k_spmv_ldsp): a block of 64 rows whose rows all hold L entries with
    column(row r, slot k) = column(row 0, slot k) + r keeps row 0's columns only.  Against the plain row-block kernel the
    product must not change by a bit -- on pure stencils (L = 1 .. 40: one slot per lane, the 9-slot batch, the tail loop),
    on sizes that leave a partial last block, on matrices where ONE entry of one block breaks the pattern (that block must
    fall back to its own columns, the others stay runs), and on the generated constant-diagonal system, whose interior
    blocks must be found to be runs."""
    from liblcg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(99)
    nblk = C.c_int64()
    for L, n in ((1, 640), (2, 700), (3, 64), (5, 6400), (9, 1000), (33, 20000), (35, 4096 + 7), (36, 5000), (40, 3000)):
        ncols = n + 5000
        offs = np.sort(rng.choice(5000, L, replace=False)).astype(np.int64)
        col = (np.arange(n)[:, None] + offs[None, :]).astype(np.int32)
        rp = (np.arange(n + 1) * L).astype(np.int32)
        for broken in (False, True):
            c = col.copy()
            want_runs = (n + 63) // 64
            if broken and n > 64 and L > 1:
                c[70, L // 2] = c[70, L // 2 - 1] + 1 if c[70, L // 2 - 1] + 1 < c[70, L // 2] else c[70, L // 2] - 1     # still sorted, still in range
                if np.array_equal(c, col):
                    continue
                want_runs -= 1
            val = rng.standard_normal(n * L); x = rng.standard_normal(ncols)
            A = api.CsrMatrix.from_csr(rp, c.reshape(-1), val, n_cols=ncols)
            xd = torch.from_numpy(x).cuda()
            y0 = torch.empty(n, dtype=torch.float64, device="cuda"); y1 = torch.full_like(y0, 5.0)
            A.set_kernel(-64)
            assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0
            A.spmv(xd, y0); api.synchronize()
            assert lib.lcg_hip_csr_set_packed(A.h, 1) == 0
            A.spmv(xd, y1); api.synchronize()
            if 64 * L <= 2240:      # the block fits one LDS window: the packed kernel answers
                assert "run blocks" in lib.lcg_hip_csr_last_kernel(A.h).decode(), (L, n, lib.lcg_hip_csr_last_kernel(A.h))
                assert lib.lcg_hip_csr_packed_runs(A.h, C.byref(nblk)) == want_runs and nblk.value == (n + 63) // 64, (L, n, broken)
            assert torch.equal(y0, y1), (L, n, broken)
            ref = port.csr_matvec(rp, c.reshape(-1), val, x)
            assert np.abs(y1.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
            A.destroy()
    # the generated constant-diagonal system: all blocks away from the first and last `band` rows are runs
    nn, band = 400000, 3000
    B = api.CsrMatrix.generate(nn, 16, band, True, 5, 0.01)
    xb = torch.empty(nn, dtype=torch.float64, device="cuda"); api.gen_xtrue(nn, 3, 0, nn, xb)
    z0 = torch.empty_like(xb); z1 = torch.empty_like(xb)
    assert lib.lcg_hip_csr_set_packed(B.h, 0) == 0
    B.spmv(xb, z0); api.synchronize()
    assert lib.lcg_hip_csr_set_packed(B.h, 1) == 0
    B.spmv(xb, z1); api.synchronize()
    runs = lib.lcg_hip_csr_packed_runs(B.h, C.byref(nblk))
    assert nblk.value - 2 * (band // 64 + 2) <= runs < nblk.value, (runs, nblk.value)
    assert torch.equal(z0, z1)
    # a row-random band has none
    B2 = api.CsrMatrix.generate(nn, 16, band, True, 5, 0.01, pattern=api.GEN_ROW_RANDOM_BAND)
    assert lib.lcg_hip_csr_set_tiled(B2.h, 0) == 0 and lib.lcg_hip_csr_set_packed(B2.h, 1) == 0
    B2.spmv(xb, z1); api.synchronize()
    assert lib.lcg_hip_csr_packed_runs(B2.h, None) == 0 and "run blocks" not in lib.lcg_hip_csr_last_kernel(B2.h).decode()


def _stencil(dims, reach, faces=False):
    """CSR of the full (2 reach + 1)^d-point stencil on a grid of the given dims (last index fastest); faces: the 2d + 1-point
    stencil (neighbours along one axis only)."""
    n = int(np.prod(dims))
    idx = np.arange(n).reshape(dims)
    rows, cols = [], []
    rng = range(-reach, reach + 1)
    import itertools
    for d in itertools.product(*([rng] * len(dims))):
        if faces and sum(1 for k in d if k) > 1:
            continue
        src = idx[tuple(slice(max(0, -k), m - max(0, k)) for k, m in zip(d, dims))].ravel()
        dst = idx[tuple(slice(max(0, k), m - max(0, -k)) for k, m in zip(d, dims))].ravel()
        rows.append(src); cols.append(dst)
    r = np.concatenate(rows); c = np.concatenate(cols)
    order = np.lexsort((c, r))
    r, c = r[order], c[order]
    rp = np.zeros(n + 1, np.int64); np.add.at(rp, r + 1, 1)
    return n, np.cumsum(rp).astype(np.int32), c.astype(np.int32)


def test_template_blocks_of_the_packed_form(api, port):
    """Template blocks (csr.hip: k_pk_meta / the template path of k_spmv_ldsp): a block of 64 rows whose entries all lie on the
    <= 64 diagonals keeps those offsets and a 64-bit mask per row -- the blocks of a stencil that grid boundaries
    pass through.  27-point stencils on grids whose lines are shorter than, equal to, and no multiple of the 64-row block, a
    5 x 5 stencil in 2D: every block must be a run or a template, y must equal the plain row-block kernel's bit for bit, the
    carried dot must agree with numpy, and with one row moved off the diagonals its block must fall back alone (more than 32
    diagonals) or take the new diagonals in (the template is the union of the block's offsets).  A 7-point stencil with three unknowns
    per point (35 diagonals) needs the masks' full width."""
    from liblcg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(27)
    nblk = C.c_int64()
    for dims, reach in (((9, 7, 5), 1), ((20, 33, 64), 1), ((6, 10, 128), 1), ((11, 13, 100), 1), ((70, 90), 2), ((3, 257), 2)):
        n, rp, ci = _stencil(dims, reach)
        val = rng.standard_normal(len(ci)); x = rng.standard_normal(n)
        for broken in (False, True):
            c = ci.copy()
            if broken:      # rows 70 .. 81: every entry 2, 4, .. 24 columns further (still ascending): diagonals the block does not have
                if n <= 82 or c[rp[82] - 1] + 24 >= n:
                    continue
                for j in range(12):
                    c[rp[70 + j]:rp[71 + j]] += 2 * (j + 1)
            A = api.CsrMatrix.from_csr(rp, c, val)
            xd = torch.from_numpy(x).cuda()
            y0 = torch.empty(n, dtype=torch.float64, device="cuda"); y1 = torch.full_like(y0, 5.0)
            A.set_kernel(-64)
            assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0
            A.spmv(xd, y0); api.synchronize()
            assert lib.lcg_hip_csr_set_packed(A.h, 1) == 0
            A.spmv(xd, y1); api.synchronize()
            name = lib.lcg_hip_csr_last_kernel(A.h).decode()
            runs = lib.lcg_hip_csr_packed_runs(A.h, C.byref(nblk)); tpls = lib.lcg_hip_csr_packed_templates(A.h)
            assert "k_spmv_ldsp" in name and "template blocks" in name, (dims, name)
            # every block with at most 64 distinct offsets (column - row in block) is a run or a template; the others -- where the
            # shifted rows bring more diagonals than a mask has bits -- fall back to their packed columns, alone
            lane = np.arange(n) % 64
            offs = c - np.repeat(lane, np.diff(rp))
            blk = np.repeat(np.arange(n) // 64, np.diff(rp))
            distinct = np.array([len(np.unique(offs[blk == b])) for b in range((n + 63) // 64)])
            assert runs + tpls == int((distinct <= 64).sum()) and tpls > 0, (dims, broken, runs, tpls, nblk.value, distinct[:4])
            if broken and dims == (9, 7, 5):       # (here the shifted rows bring the block past 64 diagonals: it falls back, alone)
                assert distinct[1] > 64 and runs + tpls == nblk.value - 1, distinct[:3]
            assert torch.equal(y0, y1), (dims, broken)
            ref = port.csr_matvec(rp, c, val, x)
            bound = port.csr_matvec(rp, c, np.abs(val), np.abs(x))
            assert float(np.max(np.abs(y1.cpu().numpy() - ref) / bound)) <= 1e-13, (dims, broken)
            # the product the solver loops run: the dot carried by template blocks too
            u = torch.from_numpy(rng.standard_normal(n)).cuda()
            sums = (C.c_double * 2)()
            y2 = torch.empty_like(y0)
            A.set_kernel(0)        # (the automatic choice: rows of ~27 entries take the packed kernel with the dot in its epilogue)
            assert lib.lcg_hip_spmv_dot(A.h, xd.data_ptr(), y2.data_ptr(), u.data_ptr(), sums) == 0
            # (another number of rows per block than the forced 64 above where the rows are short: other last bits)
            assert float(np.max(np.abs(y2.cpu().numpy() - ref) / bound)) <= 1e-13, (dims, broken, lib.lcg_hip_csr_last_kernel(A.h))
            yu = float(ref @ u.cpu().numpy()); yy = float(ref @ ref)
            assert abs(sums[0] - yu) <= 1e-11 * float(np.abs(ref) @ np.abs(u.cpu().numpy())) and abs(sums[1] - yy) <= 1e-12 * yy
            A.destroy()


def test_template_blocks_with_several_unknowns_per_point(api, port):
    """7-point stencil x 4 unknowns (28 entries per row on 43 diagonals) and 9-point 2D x 3 (27 entries on 33): template blocks
    need more than 32 mask bits; bit-equal to the plain kernel."""
    import scipy.sparse as sp
    from liblcg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    nblk = C.c_int64()
    for dims, reach, faces, dof in (((12, 11, 20), 1, True, 4), ((40, 70), 1, False, 3)):
        n0, rp0, ci0 = _stencil(dims, reach, faces)
        B = sp.kron(sp.csr_matrix((np.ones(len(ci0)), ci0, rp0), shape=(n0, n0)), np.ones((dof, dof)), format="csr"); B.sort_indices()
        n = n0 * dof
        rp, ci = B.indptr.astype(np.int32), B.indices.astype(np.int32)
        val = rng.standard_normal(len(ci)); x = rng.standard_normal(n)
        A = api.CsrMatrix.from_csr(rp, ci, val)
        xd = torch.from_numpy(x).cuda()
        y0 = torch.empty(n, dtype=torch.float64, device="cuda"); y1 = torch.full_like(y0, 5.0)
        A.set_kernel(-64)
        assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0
        A.spmv(xd, y0); api.synchronize()
        assert lib.lcg_hip_csr_set_packed(A.h, 1) == 0
        A.spmv(xd, y1); api.synchronize()
        runs = lib.lcg_hip_csr_packed_runs(A.h, C.byref(nblk)); tpls = lib.lcg_hip_csr_packed_templates(A.h)
        lane = np.arange(n) % 64
        offs = ci - np.repeat(lane, np.diff(rp)); blk = np.repeat(np.arange(n) // 64, np.diff(rp))
        distinct = np.array([len(np.unique(offs[blk == b])) for b in range((n + 63) // 64)])
        assert distinct.max() > 32 and runs + tpls == int((distinct <= 64).sum()) == nblk.value, (dims, runs, tpls, nblk.value, distinct.max())
        assert torch.equal(y0, y1), dims
        ref = port.csr_matvec(rp, ci, val, x)
        assert float(np.max(np.abs(y1.cpu().numpy() - ref) / port.csr_matvec(rp, ci, np.abs(val), np.abs(x)))) <= 1e-13
        A.destroy()


def test_short_row_runs_one_wavefront_per_block(api, port):
    """k_spmv_run1 (short rows whose 64-row blocks are mostly runs -- stencils): against the LDS-staged kernel the automatic
    choice would otherwise take (packed copy switched off), bit for bit.  L = 1 .. 17 covers one and two partial sums per row
    (T = 1 up to 8.5 entries per row, T = 2 up to 17) and more than one batch of 8; sizes leave partial last blocks; every
    third system has blocks that are not runs (a shortened row per 1000 rows, like the Laplacian's grid-row boundaries), which
    the kernel walks from the CSR arrays; a system whose blocks are mostly not runs must stay with the staged kernel."""
    from liblcg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(123)
    res2 = (C.c_double * 2)()
    case = 0
    for L in (1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17):
        for n in (64, 1000, 6400 + 13, 50000):
            case += 1
            ncols = n + 3000
            offs = np.sort(rng.choice(3000, L, replace=False)).astype(np.int64)
            lens = np.full(n, L)
            if case % 3 == 0 and L > 1 and n >= 1000:
                lens[::1000] = L - 1                      # rows that break their block's run
            rp = np.zeros(n + 1, np.int32); np.cumsum(lens, out=rp[1:])
            col = np.concatenate([(i + offs[:lens[i]]) for i in range(n)]).astype(np.int32) if n <= 6500 else None
            if col is None:
                full = (np.arange(n)[:, None] + offs[None, :])
                mask = np.arange(L)[None, :] < lens[:, None]
                col = full[mask].astype(np.int32)
            val = rng.standard_normal(rp[-1]); x = rng.standard_normal(ncols)
            A = api.CsrMatrix.from_csr(rp, col, val, n_cols=ncols)
            xd = torch.from_numpy(x).cuda()
            y0 = torch.empty(n, dtype=torch.float64, device="cuda"); y1 = torch.full_like(y0, 5.0)
            assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0
            A.spmv(xd, y0); api.synchronize()
            assert "k_spmv_lds1" in lib.lcg_hip_csr_last_kernel(A.h).decode()
            assert lib.lcg_hip_csr_set_packed(A.h, -1) == 0
            A.spmv(xd, y1); api.synchronize()
            assert "k_spmv_run1" in lib.lcg_hip_csr_last_kernel(A.h).decode(), (L, n, lib.lcg_hip_csr_last_kernel(A.h))
            assert torch.equal(y0, y1), (L, n)
            ref = port.csr_matvec(rp, col, val, x)
            assert np.abs(y1.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
            # the same product carrying y.u and y.y: k_spmv_run1d up to 15 entries per row (its eight wavefronts' LDS), the
            # staged kernel's twin or two launches beyond -- y unchanged, sums to rounding
            u = rng.standard_normal(n); ud = torch.from_numpy(u).cuda()
            y1.fill_(7.0)
            assert lib.lcg_hip_spmv_dot(A.h, xd.data_ptr(), y1.data_ptr(), ud.data_ptr(), res2) == 0
            kern = lib.lcg_hip_csr_last_kernel(A.h).decode()
            assert ("k_spmv_run1d" in kern) == (L <= 15), (L, n, kern)
            assert torch.equal(y0, y1), (L, n, kern)
            assert abs(res2[0] - float(ref @ u)) <= 1e-12 * float(np.abs(ref) @ np.abs(u)) and abs(res2[1] - float(ref @ ref)) <= 1e-12 * float(ref @ ref)
            A.destroy()
    # short-row stencils whose EVERY block holds boundary rows (grid lines no longer than the block): template blocks in the
    # one-wavefront-per-block kernel -- 7-point in 3D, 5-point and 9-point in 2D, 13-point (reach 2) in 3D
    for dims, reach, faces in (((6, 9, 16), 1, True), ((5, 40, 64), 1, True), ((7, 5, 128), 1, True), ((40, 50), 1, True), ((33, 64), 1, False),
                               ((9, 8, 30), 2, True)):
        n, rp, col = _stencil(dims, reach, faces)
        val = rng.standard_normal(len(col)); x = rng.standard_normal(n)
        A = api.CsrMatrix.from_csr(rp, col, val)
        xd = torch.from_numpy(x).cuda()
        y0 = torch.empty(n, dtype=torch.float64, device="cuda"); y1 = torch.full_like(y0, 5.0)
        assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0
        A.spmv(xd, y0); api.synchronize()
        assert "k_spmv_lds1" in lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert lib.lcg_hip_csr_set_packed(A.h, -1) == 0
        A.spmv(xd, y1); api.synchronize()
        assert "k_spmv_run1" in lib.lcg_hip_csr_last_kernel(A.h).decode(), (dims, lib.lcg_hip_csr_last_kernel(A.h))
        assert lib.lcg_hip_csr_packed_templates(A.h) > 0, dims
        assert torch.equal(y0, y1), dims
        ref = port.csr_matvec(rp, col, val, x)
        assert float(np.max(np.abs(y1.cpu().numpy() - ref) / port.csr_matvec(rp, col, np.abs(val), np.abs(x)))) <= 1e-13
        u = rng.standard_normal(n); ud = torch.from_numpy(u).cuda()
        y1.fill_(7.0)
        assert lib.lcg_hip_spmv_dot(A.h, xd.data_ptr(), y1.data_ptr(), ud.data_ptr(), res2) == 0
        assert torch.equal(y0, y1), (dims, lib.lcg_hip_csr_last_kernel(A.h))
        assert abs(res2[0] - float(ref @ u)) <= 1e-12 * float(np.abs(ref) @ np.abs(u)) and abs(res2[1] - float(ref @ ref)) <= 1e-12 * float(ref @ ref)
        A.destroy()
    # the Laplacian of configs[1] at a tenth of the size: mostly runs; a ragged matrix of the same density: none -> staged kernel
    A = api.CsrMatrix.laplace2d(400, 250); n = 100000
    xd = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 3, 0, n, xd)
    y0 = torch.empty_like(xd); y1 = torch.empty_like(xd)
    assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0
    A.spmv(xd, y0)
    assert lib.lcg_hip_csr_set_packed(A.h, -1) == 0
    A.spmv(xd, y1); api.synchronize()
    nblk = C.c_int64()
    runs = lib.lcg_hip_csr_packed_runs(A.h, C.byref(nblk))
    assert "k_spmv_run1" in lib.lcg_hip_csr_last_kernel(A.h).decode() and 0.6 * nblk.value < runs < nblk.value and torch.equal(y0, y1)
    rp, col = _ragged(rng, 20000, 20000, 9)
    B = api.CsrMatrix.from_csr(rp, col, rng.standard_normal(rp[-1]))
    xb = torch.from_numpy(rng.standard_normal(20000)).cuda(); yb = torch.empty_like(xb)
    B.spmv(xb, yb); api.synchronize()
    assert "k_spmv_lds1" in lib.lcg_hip_csr_last_kernel(B.h).decode() and lib.lcg_hip_csr_packed_runs(B.h, None) == 0


@pytest.mark.parametrize("cplx", [False, True])
def test_spmv_transpose_and_conjugate(api, port, cplx):
    """op(A).x for the (layout, conjugate) pairs of clcg_axfunc_ptr (clcg.h:40-41): A^T, A^H, conj(A)."""
    from liblcg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(21)
    n = 2500
    rp, col = _ragged(rng, n, n, 30, long_rows=[(7, 2400)])
    val = rng.standard_normal(rp[-1]) + (1j * rng.standard_normal(rp[-1]) if cplx else 0)
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    A = api.CsrMatrix.from_csr(rp, col, val)
    xd = torch.from_numpy(x).cuda(); yd = torch.empty_like(xd)
    import scipy.sparse as sp
    M = sp.csr_matrix((val, col, rp), shape=(n, n))
    for layout in (0, 1):
        for conj in (0, 1):
            ref = (M.T if layout else M)
            ref = (ref.conj() if conj else ref) @ x
            if cplx:
                o = port.lib  # oracle's scatter product agrees with scipy
                yo = np.empty(n, np.complex128)
                o.orc_csr_cmatvec_op(rp.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
                                     np.ascontiguousarray(val).ctypes.data_as(C.c_void_p),
                                     np.ascontiguousarray(x).ctypes.data_as(C.c_void_p), yo.ctypes.data_as(C.c_void_p),
                                     C.c_int(n), C.c_int(layout), C.c_int(conj))
                assert np.abs(yo - ref).max() <= 1e-11 * np.abs(ref).max()
            for rep in range(2):        # second call reuses the cached op(A)
                yd.zero_()
                assert lib.lcg_hip_spmv_op(A.h, xd.data_ptr(), yd.data_ptr(), layout, conj) == 0
                api.synchronize()
                assert np.abs(yd.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max(), (layout, conj)
    # the built transpose is deterministic: two independent builds give identical bits
    B = api.CsrMatrix.from_csr(rp, col, val)
    y2 = torch.empty_like(xd)
    lib.lcg_hip_spmv_op(A.h, xd.data_ptr(), yd.data_ptr(), 1, 1); lib.lcg_hip_spmv_op(B.h, xd.data_ptr(), y2.data_ptr(), 1, 1)
    api.synchronize()
    assert torch.equal(yd, y2)


def test_product_carrying_its_dot(api, port):
    """lcg_hip_spmv_dot: y = A.x with y.u and y.y riding in the product's epilogue (k_spmv_lds1d, what the one-GPU solver loops
    use for the dot that follows every A.x: lcg.cpp:234, 548-552, 735-740).  y must be BIT-identical to the plain product's;
    the sums agree with the oracle's product summed in numpy to rounding; the same bits on a second call.  Row lengths choose
    every rows-per-block shape (256 .. 16); ragged rows, empty rows, sizes around the block edges; systems the fused form does
    not take (rows longer than a window, more row blocks than the threshold) answer through product + reduction."""
    from liblcg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    res = (C.c_double * 2)()
    cases = [(n, ml) for ml in (3, 12, 30, 60, 130) for n in (1, 63, 64, 65, 257, 5000)] + [(40000, 6), (3001, 40)]
    for n, max_len in cases:
        long_rows = [(5, 5000)] if (n, max_len) == (3001, 40) else ()
        rp, col = _ragged(rng, n, n, max_len, long_rows=long_rows)
        if rp[-1] == 0:
            continue
        val = rng.standard_normal(rp[-1]); x = rng.standard_normal(n); u = rng.standard_normal(n)
        A = api.CsrMatrix.from_csr(rp, col, val)
        xd = torch.from_numpy(x).cuda(); ud = torch.from_numpy(u).cuda()
        y0 = torch.empty_like(xd); y1 = torch.full_like(xd, 3.0)
        A.spmv(xd, y0); api.synchronize()
        assert lib.lcg_hip_spmv_dot(A.h, xd.data_ptr(), y1.data_ptr(), ud.data_ptr(), res) == 0
        kern = lib.lcg_hip_csr_last_kernel(A.h).decode()
        fused = "carrying the dot" in kern
        if long_rows:
            assert not fused, kern          # a row longer than an LDS window: the windowed kernel, two launches
        elif max_len <= 30:
            assert fused, (n, max_len, kern)    # (longer rows: some block's slice may exceed one window, then as above)
        assert torch.equal(y0, y1), (n, max_len, kern)
        yr = port.csr_matvec(rp, col, val, x)
        bound = float(np.abs(yr) @ np.abs(u)) + 1e-300, float(yr @ yr) + 1e-300
        assert abs(res[0] - float(yr @ u)) <= 1e-12 * bound[0] and abs(res[1] - float(yr @ yr)) <= 1e-12 * bound[1], (n, max_len)
        first = (res[0], res[1])
        assert lib.lcg_hip_spmv_dot(A.h, xd.data_ptr(), y1.data_ptr(), ud.data_ptr(), res) == 0
        assert (res[0], res[1]) == first
        A.destroy()
    # the packed kernel of the large matrices (run blocks and blocks with their own columns) carries the dot too: one partial
    # per block of 64 rows, folded by k_axp_fold; symmetric (runs in the interior) and row-random (no runs) columns
    for pattern, n in ((api.GEN_DIAGONALS, 300000), (api.GEN_ROW_RANDOM_BAND, 200000)):
        A = api.CsrMatrix.generate(n, 16, 3000, True, 9, 0.01, pattern=pattern)
        assert lib.lcg_hip_csr_set_tiled(A.h, 0) == 0 and lib.lcg_hip_csr_set_packed(A.h, 1) == 0
        xd = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 3, 0, n, xd)
        ud = torch.empty_like(xd); api.gen_xtrue(n, 4, 0, n, ud)
        y0 = torch.empty_like(xd); y1 = torch.full_like(xd, 2.0)
        A.spmv(xd, y0); api.synchronize()
        assert "k_spmv_ldsp" in lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert lib.lcg_hip_spmv_dot(A.h, xd.data_ptr(), y1.data_ptr(), ud.data_ptr(), res) == 0
        assert "carrying the dot" in lib.lcg_hip_csr_last_kernel(A.h).decode() and torch.equal(y0, y1)
        assert abs(res[0] - float(y0 @ ud)) <= 1e-12 * float(y0.abs() @ ud.abs()) and abs(res[1] - float(y0 @ y0)) <= 1e-12 * float(y0 @ y0)
        first = (res[0], res[1])
        assert lib.lcg_hip_spmv_dot(A.h, xd.data_ptr(), y1.data_ptr(), ud.data_ptr(), res) == 0 and (res[0], res[1]) == first
        A.destroy()
    # short-row stencils: the one-wavefront-per-block kernel carries the dot too (k_spmv_run1d, a ticket instead of a barrier)
    A = api.CsrMatrix.laplace2d(800, 800); n = 640000
    xd = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 3, 0, n, xd)
    ud = torch.empty_like(xd); api.gen_xtrue(n, 8, 0, n, ud)
    y0 = torch.empty_like(xd); y1 = torch.empty_like(xd)
    A.spmv(xd, y0)
    assert "k_spmv_run1 " in lib.lcg_hip_csr_last_kernel(A.h).decode()
    for _ in range(2):
        assert lib.lcg_hip_spmv_dot(A.h, xd.data_ptr(), y1.data_ptr(), ud.data_ptr(), res) == 0
        assert "k_spmv_run1d" in lib.lcg_hip_csr_last_kernel(A.h).decode() and torch.equal(y0, y1)
        assert abs(res[0] - float(y0 @ ud)) <= 1e-12 * float(y0.abs() @ ud.abs()) and abs(res[1] - float(y0 @ y0)) <= 1e-12 * float(y0 @ y0)
    A.destroy()
    # more row blocks than the threshold and no runs: two launches, same answers
    n = 700000
    rp, col = _ragged(rng, n, n, 9)
    val = rng.standard_normal(rp[-1])
    A = api.CsrMatrix.from_csr(rp, col, val)
    xd = torch.from_numpy(rng.standard_normal(n)).cuda()
    y0 = torch.empty_like(xd); y1 = torch.empty_like(xd)
    A.spmv(xd, y0)
    assert lib.lcg_hip_spmv_dot(A.h, xd.data_ptr(), y1.data_ptr(), xd.data_ptr(), res) == 0
    assert "carrying the dot" not in lib.lcg_hip_csr_last_kernel(A.h).decode() and torch.equal(y0, y1)
    assert abs(res[0] - float(y0 @ xd)) <= 1e-12 * float(y0.abs() @ xd.abs()) and abs(res[1] - float(y0 @ y0)) <= 1e-12 * float(y0 @ y0)


def test_spmv_edge_shapes(api, port):
    rng = np.random.default_rng(3)
    for n in (1, 2, 63, 64, 65, 255, 256, 257):
        rp, col = _ragged(rng, n, n, 9)
        val = rng.standard_normal(rp[-1])
        if rp[-1] == 0:
            continue
        x = rng.standard_normal(n)
        A = api.CsrMatrix.from_csr(rp, col, val)
        xd = torch.from_numpy(x).cuda(); yd = torch.full_like(xd, 7.0)
        for var in (0, -64, 8):
            A.set_kernel(var); A.spmv(xd, yd); api.synchronize()
            assert np.allclose(yd.cpu().numpy(), port.csr_matvec(rp, col, val, x), rtol=0, atol=1e-13), (n, var)


def test_blas1(api, port):
    rng = np.random.default_rng(5)
    from liblcg_amd import _lib
    lib = _lib.load()
    for n in (1, 2, 3, 1000, 1001, 524289, 3_000_001):
        a = rng.standard_normal(n); b = rng.standard_normal(n)
        ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        exact = float(np.dot(a.astype(np.longdouble), b.astype(np.longdouble)))
        tol = 1e-13 * float(np.dot(np.abs(a), np.abs(b))) + 1e-300
        assert abs(api.dot(ad, bd) - exact) <= tol
        assert abs(api.dot(ad, bd) - port.dot(a, b)) <= 1.2e-16 * n * float(np.dot(np.abs(a), np.abs(b))) + tol   # serial-sum bound
        assert abs(api.nrm2(ad) - np.linalg.norm(a)) <= 1e-13 * np.linalg.norm(a)
        # unaligned (odd offset) views take the scalar path
        if n > 3:
            assert abs(api.dot(ad[1:], bd[1:]) - float(np.dot(a[1:], b[1:]))) <= 100 * tol
        yd = bd.clone()
        lib.lcg_hip_axpy(n, 0.75, ad.data_ptr(), yd.data_ptr()); api.synchronize()
        mag = 4e-16 * (np.abs(b) + np.abs(a)).max()          # FMA vs mul+add: 1 ulp of the operands
        assert np.allclose(yd.cpu().numpy(), b + 0.75 * a, rtol=0, atol=mag)
        lib.lcg_hip_scal(n, -2.0, yd.data_ptr()); api.synchronize()
        assert np.allclose(yd.cpu().numpy(), -2.0 * (b + 0.75 * a), rtol=0, atol=2 * mag)
        cd = torch.empty_like(ad)
        lib.lcg_hip_vecmul(n, ad.data_ptr(), bd.data_ptr(), cd.data_ptr()); api.synchronize()
        assert np.array_equal(cd.cpu().numpy(), a * b)
        lib.lcg_hip_vecdiv(n, ad.data_ptr(), bd.data_ptr(), cd.data_ptr()); api.synchronize()
        assert np.allclose(cd.cpu().numpy(), a / b, rtol=2e-16, atol=0)


def test_complex_blas1(api):
    rng = np.random.default_rng(6)
    from liblcg_amd import _lib
    lib = _lib.load()
    n = 100_003
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n); b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert abs(api.cdot(ad, bd) - np.sum(a * b)) <= 1e-10                  # clcg_dot: no conjugate
    assert abs(api.cdot(ad, bd, conj=True) - np.vdot(a, b)) <= 1e-10       # clcg_inner: conj(a).b
    yd = bd.clone(); alpha = (C.c_double * 2)(0.5, -1.25)
    lib.clcg_hip_axpy(n, alpha, ad.data_ptr(), yd.data_ptr()); api.synchronize()
    assert np.allclose(yd.cpu().numpy(), b + (0.5 - 1.25j) * a, rtol=1e-14, atol=1e-14)
    cd = torch.empty_like(ad)
    lib.clcg_hip_vecdiv(n, ad.data_ptr(), bd.data_ptr(), cd.data_ptr()); api.synchronize()
    assert np.allclose(cd.cpu().numpy(), a / b, rtol=1e-14, atol=0)


def test_jacobi_and_diagonal(api, port, case10k, case1kc):
    n, rp, ci, v, b, _ = case10k
    A = api.CsrMatrix.from_csr(rp, ci, v)
    d = torch.empty(n, dtype=torch.float64, device="cuda")
    A.build_jacobi(d); api.synchronize()
    assert np.array_equal(d.cpu().numpy(), port.csr_diag(rp, ci, v))       # algebra_cuda.cu:40-57
    from liblcg_amd import _lib
    lib = _lib.load()
    r = torch.from_numpy(b).cuda(); z = torch.empty_like(r)
    lib.lcg_hip_jacobi_mx(A.h, r.data_ptr(), z.data_ptr(), n); api.synchronize()
    assert np.allclose(z.cpu().numpy(), b * (1.0 / port.csr_diag(rp, ci, v)), rtol=1e-16, atol=0)
    n, rp, ci, v, b, _ = case1kc
    A = api.CsrMatrix.from_csr(rp, ci, v)
    d = torch.empty(n, dtype=torch.complex128, device="cuda")
    A.build_jacobi(d); api.synchronize()
    assert np.array_equal(d.cpu().numpy(), port.csr_diag(rp, ci, v))       # lcg_complex_cuda.cu:46-63


def test_coo_ingest(api, port, case10k):
    import os
    from conftest import GOLDEN
    from liblcg_amd.coo_io import read_coo_system
    n, row, col, val, b = read_coo_system(os.path.join(GOLDEN, "case_10K_A"))
    rp, perm = port.coo_to_csr(row, col, n)
    A = api.CsrMatrix.from_coo(n, row, col, val)                            # row-sorted file: device path
    rp2, ci2, v2 = A.arrays_to_host()
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, col[perm]) and np.array_equal(v2, val[perm])
    sh = np.random.default_rng(1).permutation(len(row))                     # unsorted: stable host sort
    A = api.CsrMatrix.from_coo(n, row[sh], col[sh], val[sh])
    rp3, ci3, v3 = A.arrays_to_host()
    rp4, perm4 = port.coo_to_csr(row[sh], col[sh], n)
    assert np.array_equal(rp3, rp4) and np.array_equal(ci3, col[sh][perm4]) and np.array_equal(v3, val[sh][perm4])
    x = np.random.default_rng(2).standard_normal(n)
    xd = torch.from_numpy(x).cuda(); yd = torch.empty_like(xd)
    A.spmv(xd, yd); api.synchronize()
    assert np.allclose(yd.cpu().numpy(), port.coo_matvec(row, col, val, x), rtol=0, atol=1e-12)
    # duplicates of one (row, col) keep their input order too (stable), complex values included
    r2 = np.array([3, 0, 3, 3, 1, 0, 3], np.int32); c2 = np.array([1, 2, 1, 0, 1, 2, 1], np.int32)
    v2 = (np.arange(7) + 1) * (1 + 0.5j)
    A2 = api.CsrMatrix.from_coo(4, r2, c2, v2)
    rp5, ci5, v5 = A2.arrays_to_host()
    rp6, perm6 = port.coo_to_csr(r2, c2, 4)
    assert np.array_equal(rp5, rp6) and np.array_equal(ci5, c2[perm6]) and np.array_equal(v5, v2[perm6])
    bad = row.copy(); bad[17] = n
    with pytest.raises(api.LcgHipError):
        api.CsrMatrix.from_coo(n, bad[sh], col[sh], val[sh])


@pytest.mark.parametrize("band,sym", [(64, True), (0, True), (1000, False), (0, False), (3, True)])
def test_generator_twin_is_bit_identical(api, port, band, sym):
    n = 7777
    g = port.gen_init(n, 16, band, sym, 5, 0.01)
    for r0, r1 in ((0, n), (1234, 4321), (n - 100, n)):
        rp, ci, v = port.gen_rows(g, r0, r1)
        A = api.CsrMatrix.generate(n, 16, band, sym, 5, 0.01, r0, r1)
        rp2, ci2, v2 = A.arrays_to_host()
        assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2) and np.array_equal(v, v2)
    xt = torch.empty(n, dtype=torch.float64, device="cuda")
    api.gen_xtrue(n, 5, 0, n, xt); api.synchronize()
    assert np.array_equal(xt.cpu().numpy(), port.gen_xtrue(g))
    if sym:     # symmetric and strictly diagonally dominant => SPD
        import scipy.sparse as sp
        rp, ci, v = port.gen_rows(g)
        M = sp.csr_matrix((v, ci, rp), shape=(n, n))
        assert abs(M - M.T).max() == 0.0
        off = abs(M).sum(axis=1).A1 - M.diagonal()
        assert np.all(M.diagonal() - off > 0.0099)


@pytest.mark.parametrize("band,sym", [(600, True), (64, False), (1, True), (5000, True)])
def test_row_random_band_generator_twin(api, port, band, sym):
    """Pattern 2 (every row draws its own columns inside (i - band, i + band)): device generator == oracle twin bit
    for bit, on whole matrices and on shards; symmetric pattern and values; columns inside the band."""
    n = 9001
    g = port.gen_init(n, 16, band, sym, 7, 0.01, pattern=2)
    for r0, r1 in ((0, n), (2000, 5555), (n - 77, n)):
        rp, ci, v = port.gen_rows(g, r0, r1)
        A = api.CsrMatrix.generate(n, 16, band, sym, 7, 0.01, r0, r1, pattern=api.GEN_ROW_RANDOM_BAND)
        rp2, ci2, v2 = A.arrays_to_host()
        assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2) and np.array_equal(v, v2)
    import scipy.sparse as sp
    rp, ci, v = port.gen_rows(g)
    rows = np.repeat(np.arange(n), np.diff(rp))
    assert np.abs(ci - rows).max() < max(2, band)
    M = sp.csr_matrix((v, ci, rp), shape=(n, n))
    P = sp.csr_matrix((np.ones_like(v), ci, rp), shape=(n, n))
    assert abs(P - P.T).max() == 0.0                               # the pattern is symmetric either way
    if sym:
        assert abs(M - M.T).max() == 0.0
        off = abs(M).sum(axis=1).A1 - M.diagonal()
        assert np.all(M.diagonal() - off > 0.0099)
    if band >= 600:     # rows really differ: the offsets of row i and row i+1 are not the same set
        o0 = set((ci[rp[4000]:rp[4001]] - 4000).tolist()); o1 = set((ci[rp[4001]:rp[4002]] - 4001).tolist())
        assert len(o0 & o1) <= 3


def test_laplace2d_generator(api):
    import scipy.sparse as sp
    nx, ny = 37, 23
    A = api.CsrMatrix.laplace2d(nx, ny)
    rp, ci, v = A.arrays_to_host()
    M = sp.csr_matrix((v, ci, rp), shape=(nx * ny, nx * ny))
    T = sp.kron(sp.eye(ny), sp.diags([-1, 2, -1], [-1, 0, 1], shape=(nx, nx))) + \
        sp.kron(sp.diags([-1, 2, -1], [-1, 0, 1], shape=(ny, ny)), sp.eye(nx))
    assert abs(M - T.tocsr()).max() == 0.0
    assert np.all(np.diff(ci[rp[5]:rp[6]]) > 0)


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_sharded_product_on_one_gpu(api, port, nranks):
    """Row shards split into local-/remote-column parts (comm.hip) reproduce the full A.x."""
    from liblcg_amd import _lib, partition
    lib = _lib.load()
    n = 10007
    for band in (50, 0):
        g = port.gen_init(n, 16, band, True, 9, 0.01)
        rp, ci, v = port.gen_rows(g)
        x = np.random.default_rng(4).standard_normal(n)
        ref = port.csr_matvec(rp, ci, v, x)
        glen = partition.gathered_length(n, nranks)
        xpad = np.zeros(glen); xpad[:n] = x
        for r in range(nranks):
            r0, r1 = partition.shard_range(n, nranks, r)
            A = api.CsrMatrix.generate(n, 16, band, True, 9, 0.01, r0, r1)
            assert lib.lcg_hip_csr_split_for_test(A.h, n, nranks, r) == 0
            xf = lib.lcg_hip_csr_xfull(A.h)
            assert lib.lcg_hip_memcpy(xf, xpad.ctypes.data, xpad.nbytes, 1) == 0
            xl = torch.from_numpy(x[r0:r1].copy()).cuda(); yl = torch.empty_like(xl)
            A.spmv(xl, yl); api.synchronize()
            assert np.abs(yl.cpu().numpy() - ref[r0:r1]).max() <= 1e-12 * np.abs(ref).max()
            if band:
                assert lib.lcg_hip_csr_local_nnz(A.h) > 0.9 * A.nnz


def test_full_size_properties(api):
    """BASELINE.json configs[2] size (10M rows, ~33 nnz/row): linearity, symmetry and a CG solve
    back to x_true -- properties that need no CPU reference."""
    n = 10_000_000
    A = api.CsrMatrix.generate(n, 16, 131072, True, 1, 0.01)
    assert 32.5 * n < A.nnz <= 33 * n
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    y = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    Ax, Ay, Az = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    A.spmv(x, Ax); A.spmv(y, Ay); api.synchronize()
    z = x + 2.0 * y
    A.spmv(z, Az); api.synchronize()
    assert (Az - (Ax + 2.0 * Ay)).abs().max().item() <= 1e-12 * Az.abs().max().item()     # linearity
    assert abs(api.dot(x, Ay) - api.dot(y, Ax)) <= 1e-12 * abs(api.dot(x, Ay))             # A = A^T
    assert api.dot(x, Ax) > 0                                                               # positive
    for var in (8, -32):    # other kernels agree at full size
        A.set_kernel(var); A.spmv(x, Az); api.synchronize()
        assert (Az - Ax).abs().max().item() <= 1e-12 * Ax.abs().max().item()
    A.set_kernel(0)
    xt = torch.empty_like(x); api.gen_xtrue(n, 1, 0, n, xt)
    b = torch.empty_like(x); A.spmv(xt, b); api.synchronize()
    m = torch.zeros_like(x)
    info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=1e-10, abs_diff=1), A, api.LCG_CG)
    assert info.ret == 0 and info.residual <= 1e-10
    assert ((m - xt).norm() / xt.norm()).item() <= 1e-5
    A.spmv(m, Ax); api.synchronize()
    assert ((Ax - b).norm().item() / n) <= 1.01e-10                       # the monitored quantity, recomputed


@pytest.mark.parametrize("pattern", ["constant_diagonals", "row_random_band", "scrambled"])
def test_near_the_int32_limit(api, pattern):
    """The largest system the int32 CSR of the reference's interface can hold at this density: 60M rows, 1.98e9 entries
    (92 % of 2^31), 24 GB of matrix -- entry offsets, the packed / tiled copies and the per-block tables all near their
    limits.  Size-independent properties: linearity, A = A^T, agreement of the chosen kernel with a lanes-per-row kernel
    down to the last rows, and a CG solve whose monitored residual is the residual of its answer."""
    from liblcg_amd import _lib
    lib = _lib.load()
    n = 60_000_000
    pat = {"constant_diagonals": api.GEN_DIAGONALS, "row_random_band": api.GEN_ROW_RANDOM_BAND, "scrambled": api.GEN_SCRAMBLED}[pattern]
    A = api.CsrMatrix.generate(n, 16, 131072 if pattern != "scrambled" else 0, True, 1, 0.01, pattern=pat)
    assert 0.9 * 2**31 < A.nnz < 2**31
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    y = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    Ax, Ay, Az = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    A.spmv(x, Ax); A.spmv(y, Ay); api.synchronize()
    # (scrambled: 2.1e8 (chunk, tile) groups of 9 entries -- the binned plan's granule offsets near THEIR limit too)
    want = {"constant_diagonals": "run blocks", "row_random_band": "k_tile_spmv", "scrambled": "k_bin_expand + k_bin_reduce"}[pattern]
    assert want in lib.lcg_hip_csr_last_kernel(A.h).decode(), (lib.lcg_hip_csr_last_kernel(A.h).decode(), lib.lcg_hip_csr_binned_status(A.h))
    z = x + 2.0 * y
    A.spmv(z, Az); api.synchronize()
    assert (Az - (Ax + 2.0 * Ay)).abs().max().item() <= 1e-12 * Az.abs().max().item()
    assert abs(api.dot(x, Ay) - api.dot(y, Ax)) <= 1e-12 * abs(api.dot(x, Ay))
    del y, Ay, z
    A.set_kernel(8); A.spmv(x, Az); api.synchronize(); A.set_kernel(0)
    assert (Az - Ax).abs().max().item() <= 1e-12 * Ax.abs().max().item()
    assert (Az[-4096:] - Ax[-4096:]).abs().max().item() <= 1e-12 * Ax.abs().max().item() and Ax[-4096:].abs().min().item() > 0
    xt = torch.empty_like(x); api.gen_xtrue(n, 1, 0, n, xt)
    b = torch.empty_like(x); A.spmv(xt, b); api.synchronize()
    m = torch.zeros_like(x)
    info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=1e-10, abs_diff=1), A, api.LCG_CG)
    # (the scrambled member is the best conditioned of the family at this size: 19 iterations)
    assert info.ret == 0 and info.residual <= 1e-10 and (10 if pattern == "scrambled" else 50) <= info.iterations <= 400
    assert ((m - xt).norm() / xt.norm()).item() <= 5.5e-5
    A.spmv(m, Ax); api.synchronize()
    true_res = (Ax - b).norm().item() / n
    assert true_res <= 1.05e-10 and abs(true_res - info.residual) <= 0.05 * info.residual
    A.destroy()


def test_block_structured_matrices_stay_with_the_row_block_kernels(api, port):
    """A stencil with several unknowns per grid point: consecutive rows of one point share all their columns, so the columns do
    NOT advance by one per row (diag-like share ~ 0) -- yet the rows of a 64-row block touch few different cache lines of x
    (k_line_ratio), and that is what the row-block kernels need.  The automatic choice must not take the tiled (nor the binned)
    product for it (1289 vs 645 us on the 27-point x 3 stencil of 3M rows), must say why, and the product must meet the oracle's."""
    import scipy.sparse as sp
    from liblcg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    for dims, reach, faces, dof in (((60, 60, 60), 1, True, 3), ((52, 52, 52), 1, False, 2)):
        n0, rp0, ci0 = _stencil(dims, reach, faces)
        S = sp.csr_matrix((np.ones(len(ci0)), ci0, rp0), shape=(n0, n0))
        B = sp.kron(S, np.ones((dof, dof)), format="csr")
        B.sort_indices()
        n = n0 * dof
        rp, ci = B.indptr.astype(np.int32), B.indices.astype(np.int32)
        assert len(ci) >= (1 << 22)                                 # the automatic choice considers its big-matrix families
        val = rng.standard_normal(len(ci)); x = rng.standard_normal(n)
        A = api.CsrMatrix.from_csr(rp, ci, val)
        xd = torch.from_numpy(x).cuda(); y = torch.empty_like(xd)
        A.spmv(xd, y); api.synchronize()
        name = lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert name.startswith(("k_spmv_lds", "k_spmv_run1")) and "k_tile" not in name and "k_bin" not in name, (dims, dof, name)
        assert "share their cache lines" in lib.lcg_hip_csr_tiled_status(A.h).decode(), lib.lcg_hip_csr_tiled_status(A.h)
        ref = port.csr_matvec(rp, ci, val, x)
        bound = port.csr_matvec(rp, ci, np.abs(val), np.abs(x))
        assert float(np.max(np.abs(y.cpu().numpy() - ref) / bound)) <= 1e-13, (dims, dof, name)
        A.destroy()


def test_lds_window_does_not_change_a_bit():
    """The row-block kernels size their LDS window to the matrix's largest block (1696 / 1872 / 2208 / 2240 entries: eight / seven / six /
    five workgroups per CU).  The window is read once per process, so the same menu of matrices is multiplied in three processes --
    automatic, the full window forced, the 2208 window as the least -- and y and the carried sums must agree to the last bit."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for win in ("", "2240", "2208"):
        env = dict(os.environ)
        env.pop("LCG_HIP_PACKED_WINDOW", None)
        if win:
            env["LCG_HIP_PACKED_WINDOW"] = win
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "_window_worker.py")], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l and l.split()[0] in ("stencil27", "stencil7x3", "stencil27x2", "stencil27x2p", "diagonals33", "band")]
        assert len(lines) == 6, r.stdout
        outs.append(lines)
    assert outs[0] == outs[1] == outs[2], outs
    kernels = {l.split()[0]: l.split()[1] for l in outs[0]}
    assert kernels["stencil27"] == kernels["diagonals33"] == kernels["band"] == kernels["stencil27x2p"] == "k_spmv_ldsp", kernels
    assert kernels["stencil27x2"] == "k_spmv_lds1", kernels
    # (the two forms of the long-row product: the same bits)
    assert [l.split()[2:] for l in outs[0] if l.startswith("stencil27x2 ")] == [l.split()[2:] for l in outs[0] if l.startswith("stencil27x2p ")]


def test_long_rows_take_packed_columns(api, port):
    """Rows too long for 64 of them to fit the LDS window (27-point stencils with 2 / 3 unknowns per point: 54 / 81 entries; irregular
    rows of 40 .. 180 entries) go through k_spmv_ldsp in blocks of 32 / 16 rows with packed columns: the same entry order per lane and the
    same order of additions as k_spmv_lds1 with as many rows per block -- y bit for bit, the carried sums against numpy, the oracle's bound."""
    import scipy.sparse as sp
    from liblcg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    cases = []
    for dims, dof in (((14, 15, 16), 2), ((12, 13, 14), 3)):
        n0, rp0, ci0 = _stencil(dims, 1, False)
        B = sp.kron(sp.csr_matrix((np.ones(len(ci0)), ci0, rp0), shape=(n0, n0)), np.ones((dof, dof)), format="csr"); B.sort_indices()
        cases.append((f"27-point x {dof}", B.indptr.astype(np.int32), B.indices.astype(np.int32)))
        if dof == 3:    # the same with ONE row that breaks the groups of three consecutive columns (its last entry moved one column on)
            rpb, cib = B.indptr.astype(np.int32), B.indices.astype(np.int32).copy()
            r = len(rpb) // 2
            if cib[rpb[r + 1] - 1] + 1 < len(rpb) - 1:
                cib[rpb[r + 1] - 1] += 1
            cases.append(("27-point x 3, one row out of step", rpb, cib))
    # ragged rows, columns drawn inside a band, some rows empty; the last system has a few rows beyond the one predicated batch of a lane
    # (9 x 16 = 144 entries at 16 rows per block): the batched loop and the serial tail behind it
    for lo, hi, nrow in ((40, 56, 9000), (40, 68, 9001), (70, 96, 7001), (70, 130, 7002), (60, 80, 5003)):
        lens = rng.integers(lo, hi + 1, nrow); lens[rng.integers(0, nrow, 20)] = 0
        if nrow == 5003:
            lens[rng.integers(0, nrow, 30)] = rng.integers(150, 181, 30)
        rp = np.zeros(nrow + 1, np.int64); rp[1:] = np.cumsum(lens)
        ci = np.empty(rp[-1], np.int64)
        for i in range(nrow):
            w = np.arange(max(0, i - 3000), min(nrow, i + 3000))
            ci[rp[i]:rp[i + 1]] = np.sort(rng.choice(w, lens[i], replace=False))
        cases.append((f"ragged {lo}..{hi}", rp.astype(np.int32), ci.astype(np.int32)))
    for name, rp, ci in cases:
        n = len(rp) - 1
        val = rng.standard_normal(len(ci)); x = rng.standard_normal(n); u = rng.standard_normal(n)
        A = api.CsrMatrix.from_csr(rp, ci, val)
        xd = torch.from_numpy(x).cuda(); ud = torch.from_numpy(u).cuda()
        y0 = torch.empty(n, dtype=torch.float64, device="cuda"); y1 = torch.full_like(y0, 5.0); y2 = torch.full_like(y0, 6.0)
        assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0
        A.spmv(xd, y0); api.synchronize()
        k0 = lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert k0.startswith("k_spmv_lds1"), (name, k0)
        assert lib.lcg_hip_csr_set_packed(A.h, 1) == 0
        A.spmv(xd, y1); api.synchronize()
        k1 = lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert "long rows" in k1, (name, k1)
        # several unknowns per point: one packed column per group of consecutive columns -- and only where EVERY row is made of such groups
        grouped = name.startswith("27-point x") and not name.endswith("out of step")
        assert ("per group of consecutive columns" in k1) == grouped, (name, k1)
        assert torch.equal(y0, y1), name
        sums = (C.c_double * 2)()
        assert lib.lcg_hip_spmv_dot(A.h, xd.data_ptr(), y2.data_ptr(), ud.data_ptr(), sums) == 0
        k2 = lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert "long rows" in k2, (name, k2)          # (the dot stays a pass of its own behind this product: DESIGN 9)
        assert torch.equal(y0, y2), name
        yh = y0.cpu().numpy()
        assert abs(sums[0] - float(yh @ u)) <= 1e-12 * float(np.abs(yh) @ np.abs(u)) and abs(sums[1] - float(yh @ yh)) <= 1e-12 * float(yh @ yh), name
        ref = port.csr_matvec(rp, ci, val, x)
        scale = port.csr_matvec(rp, ci, np.abs(val), np.abs(x))
        assert float(np.max(np.abs(yh - ref) / np.maximum(scale, 1e-300))) <= 1e-13, name
        assert 8 * len(ci) < lib.lcg_hip_csr_last_traffic_model(A.h) < 12 * len(ci) + 40 * n
        A.destroy()


def test_odd_shapes_against_the_oracle(api, port):
    """Shapes no benchmark has (scripts/odd_shapes.py at test size): rows of power-law lengths (most rows a few entries, some thousands),
    dense diagonal blocks, 200 diagonals, tridiagonal, diagonal only, a diagonal with one dense column -- whatever kernel the automatic
    choice takes, the product meets the oracle's row-wise bound, and repeats bit for bit."""
    import scipy.sparse as sp
    from liblcg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77 + FUZZ_SEED_OFFSET)
    shapes = []
    n = 120_000
    lens = np.minimum((rng.pareto(1.3, n) * 4 + 1).astype(np.int64), 3000)
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum(lens)
    ci = (np.repeat(np.arange(n), lens) + rng.integers(-20000, 20000, rp[-1])) % n
    shapes.append(("power-law rows", sp.csr_matrix((rng.standard_normal(rp[-1]), ci, rp), shape=(n, n))))
    shapes.append(("dense blocks", sp.block_diag([sp.csr_matrix(rng.standard_normal((96, 96))) for _ in range(300)], format="csr")))
    n = 30_000
    shapes.append(("200 diagonals", sp.diags([rng.standard_normal(n - o) for o in range(0, 2000, 10)], list(range(0, 2000, 10)), shape=(n, n))))
    n = 1_000_003
    shapes.append(("tridiagonal", sp.diags([rng.standard_normal(n - 1), rng.standard_normal(n), rng.standard_normal(n - 1)], [-1, 0, 1], shape=(n, n))))
    shapes.append(("diagonal", sp.diags([rng.standard_normal(n)], [0], shape=(n, n))))
    n = 400_000
    shapes.append(("dense column", sp.diags([np.ones(n)], [0], shape=(n, n), format="csr")
                   + sp.csr_matrix((rng.standard_normal(n), (np.arange(n), np.full(n, 7))), shape=(n, n))))
    for name, M in shapes:
        M = M.tocsr(); M.sum_duplicates(); M.sort_indices()
        n = M.shape[0]
        rp, ci, v = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)
        A = api.CsrMatrix.from_csr(rp, ci, v)
        xh = rng.standard_normal(n)
        x = torch.from_numpy(xh).cuda(); y = torch.full((n,), 3.0, dtype=torch.float64, device="cuda"); y2 = torch.full_like(y, 4.0)
        A.spmv(x, y); A.spmv(x, y2); api.synchronize()
        k = lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert torch.equal(y, y2), (name, k)
        ref = port.csr_matvec(rp, ci, v, xh)
        scale = port.csr_matvec(rp, ci, np.abs(v), np.abs(xh))
        assert float(np.max(np.abs(y.cpu().numpy() - ref) / np.maximum(scale, 1e-300))) <= 1e-13, (name, k)
        A.destroy()


def test_the_dot_takes_u_from_the_gathers_where_u_is_x(api):
    """k_spmv_ldsp<DOT> with u == x (CG / PCG: d.Ad): a run block that holds its own diagonal takes u of row r from its gathers --
    x[column(0, kd) + r], kd left behind row 0's columns by k_pk_pack -- instead of reading it.  The sums must be the SAME BITS as with u
    = a copy of x at another address (which is read), on generated systems with the diagonal (symmetric family), on one whose rows do NOT
    hold their diagonal (the diagonals shifted by one column: kd = -1 everywhere), and at the matrix edges (no run blocks there)."""
    from liblcg_amd import _lib
    lib = _lib.load()
    import scipy.sparse as sp
    n = 300_000
    cases = []
    A0 = api.CsrMatrix.generate(n, 16, 4096, True, 4, 0.01, pattern=api.GEN_DIAGONALS)
    cases.append(("with its diagonal", A0, True))
    offs = np.array([-3000, -517, -64, -2, 1, 2, 77, 900, 2500, 4000, 4001, 4002, 5000, 6000, 7000, 8000, 9000, 9100, 9200, 9300, 9400, 9500], np.int64)   # no 0
    i = np.arange(n, dtype=np.int64)[:, None] + offs[None, :]
    ok = (i >= 0) & (i < n)
    rp = np.zeros(n + 1, np.int64); rp[1:] = np.cumsum(ok.sum(1))
    ci = i[ok]
    rng = np.random.default_rng(5)
    A1 = api.CsrMatrix.from_csr(rp.astype(np.int32), ci.astype(np.int32), rng.standard_normal(len(ci)))
    cases.append(("without a diagonal", A1, False))
    # a 27-point stencil: template blocks (every block holds boundary rows at this grid), the diagonal among their diagonals
    ns, rps, cis = _stencil((24, 40, 64), 1, False)
    A2 = api.CsrMatrix.from_csr(rps, cis, rng.standard_normal(len(cis)))
    cases.append(("template blocks", A2, True))
    res = (C.c_double * 2)(); res2 = (C.c_double * 2)()
    for name, A, has_diag in cases:
        n = A.n
        assert lib.lcg_hip_csr_set_packed(A.h, 1) == 0
        x = torch.from_numpy(rng.standard_normal(n)).cuda(); xc = x.clone()
        y = torch.empty_like(x); y2 = torch.empty_like(x)
        assert lib.lcg_hip_spmv_dot(A.h, x.data_ptr(), y.data_ptr(), x.data_ptr(), res) == 0           # u IS x
        kern = lib.lcg_hip_csr_last_kernel(A.h).decode()
        assert "k_spmv_ldsp" in kern and ("template blocks" if name == "template blocks" else "run blocks") in kern and "carrying the dot" in kern, (name, kern)
        assert lib.lcg_hip_spmv_dot(A.h, x.data_ptr(), y2.data_ptr(), xc.data_ptr(), res2) == 0       # u = the same numbers elsewhere
        assert torch.equal(y, y2) and res[0] == res2[0] and res[1] == res2[1], (name, res[0], res2[0])
        yh = y.cpu().numpy(); xh = x.cpu().numpy()
        assert abs(res[0] - float(yh @ xh)) <= 1e-12 * float(np.abs(yh) @ np.abs(xh)), name
        A.destroy()
