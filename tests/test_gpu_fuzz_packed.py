"""-m gpu: fuzz of the packed-column A.x (csr.hip: k_spmv_ldsp) against the plain kernel (bit-exact) and the
oracle (1e-12): random ragged matrices whose 64-row blocks stay within one LDS window (<= 2240 entries), odd and
even slice starts, empty rows, empty blocks, a full window, spans below and above the 2^21 limit (above it the
plain kernel must answer), sizes that are not multiples of 64, more columns than rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_packed_columns_fuzz(port):
    from liblcg_amd import _lib, api
    lib = _lib.load()
    rng = np.random.default_rng(2026)
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
