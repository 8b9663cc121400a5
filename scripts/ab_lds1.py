import sys, time
ROOT = sys.argv[1]; sys.path.insert(0, ROOT)
import torch
from liblcg_amd import _lib, api
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
out = []
for name, gen in (("rrb plain", lambda: api.CsrMatrix.generate(10_000_000, 16, 131072, True, 1, 0.01, pattern=api.GEN_ROW_RANDOM_BAND)),
                  ("scr 2M", lambda: api.CsrMatrix.generate(2_000_000, 16, 0, True, 1, 0.01, pattern=api.GEN_SCRAMBLED)),
                  ("diag plain", lambda: api.CsrMatrix.generate(10_000_000, 16, 131072, True, 1, 0.01))):
    A = gen(); n = A.n
    lib.lcg_hip_csr_set_tiled(A.h, 0); lib.lcg_hip_csr_set_binned(A.h, 0)
    if name == "diag plain": lib.lcg_hip_csr_set_packed(A.h, 0)
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for _ in range(5): A.spmv(x, y)
    api.synchronize(); t0 = time.perf_counter()
    for _ in range(30): A.spmv(x, y)
    api.synchronize(); t = (time.perf_counter() - t0) / 30 * 1e6
    out.append(f"{name}: {t:.1f} us {lib.lcg_hip_csr_last_kernel(A.h).decode()[:40]}")
    A.destroy()
print(ROOT[-8:], " | ".join(out))
