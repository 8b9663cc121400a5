"""-m gpu: fuzz of the device COO -> CSR ingest (lcg_hip_csr_from_coo) and of the op(A) materialisation against
numpy: random unsorted and row-sorted triplets, duplicates, empty rows, one-row and one-entry systems, real and
complex; the CSR must equal a STABLE sort by row (the reference's coo2csr keeps the input order within a row), and
A^T.x / A^H.x must equal the dense-free numpy products."""
import numpy as np
import pytest

from conftest import FUZZ_SEED_OFFSET

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_coo_ingest_and_transposes_fuzz():
    from liblcg_amd import api
    rng = np.random.default_rng(314 + FUZZ_SEED_OFFSET)
    for case in range(60):
        n = int(rng.choice([1, 2, 7, 64, 65, 1000, 20011]))
        nnz = int(rng.integers(1, 40 * n + 2))
        cplx = bool(case % 3 == 2)
        row = rng.integers(0, n, nnz).astype(np.int32)
        if case % 4 == 0:
            row = np.sort(row)                             # the row-sorted fast path
        if case % 5 == 0 and n > 2:
            row[row == 1] = 0                              # an empty row 1
        col = rng.integers(0, n, nnz).astype(np.int32)
        val = rng.standard_normal(nnz) + (1j * rng.standard_normal(nnz) if cplx else 0)
        A = api.CsrMatrix.from_coo(n, row, col, val)
        rp, ci, v = A.arrays_to_host()
        order = np.argsort(row, kind="stable")
        want_rp = np.zeros(n + 1, np.int64); np.add.at(want_rp, row.astype(np.int64) + 1, 1); want_rp = np.cumsum(want_rp)
        assert np.array_equal(rp, want_rp.astype(np.int32)), case
        assert np.array_equal(ci, col[order]), case
        assert np.array_equal(v, val[order]), case
        x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
        xd = torch.from_numpy(x).cuda(); yd = torch.empty_like(xd)
        A.spmv(xd, yd); api.synchronize()
        want = np.zeros(n, dtype=x.dtype); np.add.at(want, row, val * x[col])
        scale = max(1e-300, np.abs(want).max())
        assert np.abs(yd.cpu().numpy() - want).max() <= 1e-12 * scale + 1e-13, case
        if cplx:
            from liblcg_amd import _lib
            lib = _lib.load()
            for layout, conj, f in ((1, 0, lambda a: a), (1, 1, np.conj), (0, 1, np.conj)):
                assert lib.lcg_hip_spmv_op(A.h, xd.data_ptr(), yd.data_ptr(), layout, conj) == 0
                api.synchronize()
                w = np.zeros(n, dtype=x.dtype)
                if layout == 1:
                    np.add.at(w, col, f(val) * x[row])
                else:
                    np.add.at(w, row, f(val) * x[col])
                assert np.abs(yd.cpu().numpy() - w).max() <= 1e-12 * max(1e-300, np.abs(w).max()) + 1e-13, (case, layout, conj)
        A.destroy()
