"""Row-random bands of growing width at N = 1e7: automatic choice / tiled / packed row blocks, with the statistics the choice used
(LCG_HIP_DEBUG_BINNED=1 prints diag_like and line_ratio).  gpurun -- 'LCG_HIP_DEBUG_BINNED=1 python3 scripts/band_sweep.py'"""
import sys, time; sys.path.insert(0, ".")
import torch
from liblcg_amd import _lib, api
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
n = 10_000_000
def t(A, x, y, reps=10):
    A.spmv(x, y); A.spmv(x, y); api.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): A.spmv(x, y)
    api.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for band in (2048, 4096, 6144, 8192, 12288, 16384, 24576):
    A = api.CsrMatrix.generate(n, 16, band, True, 1, 0.01, pattern=api.GEN_ROW_RANDOM_BAND)
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    ta = t(A, x, y); ka = lib.lcg_hip_csr_last_kernel(A.h).decode()[:14]
    assert lib.lcg_hip_csr_set_tiled(A.h, 1) == 0
    tt = t(A, x, y); kt = lib.lcg_hip_csr_last_kernel(A.h).decode()[:14]
    assert lib.lcg_hip_csr_set_tiled(A.h, 0) == 0
    tp = t(A, x, y); kp = lib.lcg_hip_csr_last_kernel(A.h).decode()[:14]
    print(f"band {band}: auto {ta:.0f} ({ka}) tiled {tt:.0f} ({kt}) plain {tp:.0f} ({kp})", flush=True)
    A.destroy()
