"""-m gpu: fuzz of the packed-column A.x (csr.hip: k_spmv_ldsp) against the plain kernel (bit-exact) and the
oracle (1e-12): random ragged matrices whose 64-row blocks stay within one LDS window (<= 2240 entries), odd and
even slice starts, empty rows, empty blocks, a full window, spans below and above the 2^21 limit (above it the
plain kernel must answer), sizes that are not multiples of 64, more columns than rows."""
import numpy as np
import pytest

from conftest import FUZZ_SEED_OFFSET

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_packed_columns_fuzz(port):
    from liblcg_amd import _lib, api
    lib = _lib.load()
    rng = np.random.default_rng(2026 + FUZZ_SEED_OFFSET)
    done = 0
    for case in range(40):
        n = int(rng.integers(1, 40000))
        maxlen = int(rng.integers(1, 36))
        lens = rng.integers(0, maxlen + 1, n)
        if rng.random() < 0.5:
            lens[rng.integers(0, n, max(1, n // 8))] = 0
        if rng.random() < 0.3 and n > 200:
            lens[64:192] = 0                                # whole empty blocks
        if rng.random() < 0.3 and n > 64:
            lens[:64] = 35                                  # a full window: 64 x 35 = 2240
        rp = np.zeros(n + 1, np.int64); np.cumsum(lens, out=rp[1:])
        nnz = int(rp[-1])
        if nnz == 0:
            continue
        ncols = int(rng.choice([n, n + 17, 3_000_000]))
        span = int(rng.choice([50, 5000, (1 << 21) - 1, 1 << 22]))
        base = rng.integers(0, max(1, ncols - min(span, ncols) + 1), n)
        col = np.empty(nnz, np.int32)
        for i in range(n):
            if lens[i]:
                col[rp[i]:rp[i + 1]] = base[i] + rng.integers(0, min(span, ncols - base[i]), lens[i])
        val = rng.standard_normal(nnz)
        x = rng.standard_normal(ncols)
        A = api.CsrMatrix.from_csr(rp.astype(np.int32), col, val, n_cols=ncols)
        xd = torch.from_numpy(x).cuda()
        y0 = torch.empty(n, dtype=torch.float64, device="cuda"); y1 = torch.empty_like(y0)
        A.set_kernel(-64)
        assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0
        A.spmv(xd, y0); api.synchronize()
        assert lib.lcg_hip_csr_set_packed(A.h, 1) == 0
        A.spmv(xd, y1); api.synchronize()
        assert torch.equal(y0, y1), (case, n, maxlen, span)
        ref = port.csr_matvec(rp.astype(np.int32), col, val, x)
        err = np.abs(y1.cpu().numpy() - ref).max() / max(1e-300, np.abs(ref).max())
        assert err < 1e-12, (case, err)
        A.destroy()
        done += 1
    assert done >= 30


def test_run_blocks_fuzz(port):
    """Mixtures of run blocks, near-runs and ragged blocks (csr.hip: k_pk_meta, the run paths of k_spmv_ldsp and k_spmv_run1):
    every 64-row block of a matrix is drawn as (a) a run with its own L and offsets -- unsorted, with repeated columns now and
    then --, (b) the same with one entry moved, (c) rows of equal length but unrelated columns, or (d) ragged rows; L short
    (the one-wavefront-per-block kernel) or around 33 (the packed kernel).  The automatic choice must reproduce the plain
    row-block kernel (packed copy switched off) bit for bit, and the oracle to 1e-12."""
    from liblcg_amd import _lib, api
    lib = _lib.load()
    rng = np.random.default_rng(4242 + FUZZ_SEED_OFFSET)
    kernels = set()
    for case in range(24):
        short = case % 2 == 0
        nblocks = int(rng.integers(1, 40))
        n = 64 * nblocks - int(rng.integers(0, 64))
        ncols = n + 6000
        Lmain = int(rng.integers(1, 9)) if short else int(rng.integers(28, 36))
        rows_cols = []
        for b in range(nblocks):
            r0, r1 = 64 * b, min(n, 64 * b + 64)
            kind = rng.choice(["run", "run", "run", "near", "equal", "ragged"])
            L = Lmain if rng.random() < 0.8 else max(1, Lmain + int(rng.integers(-2, 3)))
            offs = rng.integers(0, 5000, L)                      # unsorted, repeats possible
            for i in range(r0, r1):
                if kind in ("run", "near"):
                    c = i + offs
                elif kind == "equal":
                    c = rng.integers(0, ncols, L)
                else:
                    c = rng.integers(0, ncols, int(rng.integers(0, L + 3)))
                rows_cols.append(np.asarray(c, np.int64))
            if kind == "near" and r1 - r0 > 2 and L > 0:
                i = int(rng.integers(r0 + 1, r1)); rows_cols[i] = rows_cols[i].copy(); rows_cols[i][int(rng.integers(0, L))] += 1
        lens = np.array([len(c) for c in rows_cols])
        rp = np.zeros(n + 1, np.int32); np.cumsum(lens, out=rp[1:])
        if rp[-1] == 0:
            continue
        col = np.concatenate(rows_cols).astype(np.int32)
        val = rng.standard_normal(rp[-1]); x = rng.standard_normal(ncols)
        A = api.CsrMatrix.from_csr(rp, col, val, n_cols=ncols)
        xd = torch.from_numpy(x).cuda()
        y0 = torch.empty(n, dtype=torch.float64, device="cuda"); y1 = torch.full_like(y0, -3.0)
        assert lib.lcg_hip_csr_set_packed(A.h, 0) == 0
        A.spmv(xd, y0); api.synchronize()
        assert lib.lcg_hip_csr_set_packed(A.h, 1) == 0           # short rows: runs only when most blocks are runs; else the packed copy
        A.spmv(xd, y1); api.synchronize()
        kernels.add(lib.lcg_hip_csr_last_kernel(A.h).decode().split(" ")[0])
        assert torch.equal(y0, y1), (case, n, Lmain, lib.lcg_hip_csr_last_kernel(A.h))
        ref = port.csr_matvec(rp, col, val, x)
        assert np.abs(y1.cpu().numpy() - ref).max() <= 1e-12 * max(1e-300, np.abs(ref).max()), case
        A.destroy()
    assert {"k_spmv_run1", "k_spmv_ldsp"} <= kernels, kernels
