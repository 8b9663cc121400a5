"""-m gpu: fuzz of the shard split and of the three one-GPU forms of a rank's A.x (unsplit | local + remote parts with
the gather buffer filled by hand | direct exchange with the rank standing in for its neighbours:
lcg_hip_csr_direct_selfloop_for_test) -- random sizes, bands from 1 to wider than a shard, scrambled columns, 2-8 ranks,
every rank position, real generated systems and complex random ones (16-byte elements through pushes, landing zone and
the remote-column product)."""
import ctypes as C

import numpy as np
import pytest

from conftest import FUZZ_SEED_OFFSET

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _forms(api, lib, A, n, P, r, r0, r1, x, cplx):
    """y of the unsplit shard, of the split shard, and of the direct self-loop, for the same x."""
    w = 2 if cplx else 1
    xd = torch.from_numpy(x).cuda()
    ys = []
    y = torch.empty(r1 - r0, dtype=xd.dtype, device="cuda")
    A.spmv(xd, y); api.synchronize()
    ys.append(y.cpu().numpy())
    assert lib.lcg_hip_csr_split_for_test(A.h, n, P, r) == 0, lib.lcg_hip_last_error()
    xf = lib.lcg_hip_csr_xfull(A.h)
    assert lib.lcg_hip_memcpy(xf, xd.data_ptr(), 8 * w * n, 3) == 0
    xl = xd[r0:r1].contiguous()
    A.spmv(xl, y); api.synchronize()
    ys.append(y.cpu().numpy().copy())
    rc = lib.lcg_hip_csr_direct_selfloop_for_test(A.h, P, r)
    assert rc == 0, lib.lcg_hip_last_error()
    for _ in range(3):          # both halves of the landing zone and the ticket reset get used
        y.zero_()
        A.spmv(xl, y); api.synchronize()
    ys.append(y.cpu().numpy().copy())
    return ys


def test_split_and_direct_forms_fuzz():
    from liblcg_amd import _lib, api, partition
    lib = _lib.load()
    rng = np.random.default_rng(777 + FUZZ_SEED_OFFSET)
    for case in range(24):
        P = int(rng.integers(2, 9))
        r = int(rng.integers(0, P))
        cplx = case % 3 == 2
        if not cplx:
            n = int(rng.integers(8 * P, 80000))
            band = 0 if case % 4 == 1 else int(rng.integers(1, max(2, n // 2)))
            r0, r1 = partition.shard_range(n, P, r)
            if r1 <= r0:
                continue
            A = api.CsrMatrix.generate(n, 16, band, bool(case % 2), 9, 0.01, r0, r1)
            x = rng.standard_normal(n)
        else:
            n = int(rng.integers(8 * P, 20000))
            r0, r1 = partition.shard_range(n, P, r)
            if r1 <= r0:
                continue
            lens = rng.integers(0, 20, r1 - r0)
            rp = np.zeros(r1 - r0 + 1, np.int32); np.cumsum(lens, out=rp[1:])
            width = int(rng.integers(1, n))
            lo = np.clip(np.arange(r0, r1) - width, 0, n - 1); hi = np.clip(np.arange(r0, r1) + width, 1, n)
            col = np.concatenate([rng.integers(lo[i], hi[i], lens[i]) for i in range(r1 - r0)] or [np.zeros(0)]).astype(np.int32)
            if len(col) == 0:
                continue
            val = rng.standard_normal(len(col)) + 1j * rng.standard_normal(len(col))
            A = api.CsrMatrix.from_csr(rp, col, val, n_cols=n)
            x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        y_unsplit, y_split, y_direct = _forms(api, lib, A, n, P, r, r0, r1, x, cplx)
        scale = max(1e-300, np.abs(y_unsplit).max())
        assert np.abs(y_split - y_unsplit).max() <= 1e-13 * scale, (case, P, r, n)
        assert np.abs(y_direct - y_unsplit).max() <= 1e-13 * scale, (case, P, r, n)
        A.destroy()
