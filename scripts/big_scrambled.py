import sys, time
sys.path.insert(0, ".")
import torch
from liblcg_amd import _lib, api
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
n = 60_000_000
t0=time.time()
A = api.CsrMatrix.generate(n, 16, 0, True, 1, 0.01, pattern=api.GEN_SCRAMBLED)
api.synchronize(); print("generated", A.nnz, time.time()-t0, flush=True)
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x); y2=torch.empty_like(x)
t0=time.time(); A.spmv(x, y); api.synchronize(); print("first", time.time()-t0, lib.lcg_hip_csr_last_kernel(A.h).decode(), lib.lcg_hip_csr_binned_status(A.h), flush=True)
for mode in ("auto","plain"):
    if mode=="plain": lib.lcg_hip_csr_set_binned(A.h, 0)
    A.spmv(x, y2 if mode=="plain" else y); api.synchronize()
    t0=time.perf_counter()
    for _ in range(5): A.spmv(x, y2 if mode=="plain" else y)
    api.synchronize(); t=(time.perf_counter()-t0)/5
    print(mode, lib.lcg_hip_csr_last_kernel(A.h).decode(), f"{t*1e3:.2f} ms  alg {(12*A.nnz+20*n)/t/1e12:.3f} TB/s  frac {(12*A.nnz+20*n)/t/8e12:.3f}  model {lib.lcg_hip_csr_last_traffic_model(A.h)/1e9:.1f} GB", flush=True)
print("max rel diff", ((y-y2).abs().max()/y.abs().max()).item())
