import os, sys, time
sys.path.insert(0, ".")
import torch
from liblcg_amd import _lib, api
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
# CG_SIZES="1000,1414,...": grid sides instead of the default four
for nx in [int(v) for v in os.environ["CG_SIZES"].split(",")] if os.environ.get("CG_SIZES") else (316, 1000, 2000, 3162):
    n = nx * nx
    A = api.CsrMatrix.laplace2d(nx, nx); A.build_jacobi()
    xt = torch.rand(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    for name, sched in (("auto", api.CG_AUTO), ("classic", api.CG_CLASSIC), ("one-red", api.CG_ONE_REDUCTION)):
        api.set_cg_schedule(sched)
        best = 1e9
        for rep in range(4):
            m = torch.zeros_like(xt); torch.cuda.synchronize(); api.synchronize()
            t0 = time.perf_counter()
            info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=1e-300, max_iterations=500), A, api.LCG_CG)
            api.synchronize(); best = min(best, (time.perf_counter() - t0) / info.iterations * 1e6)
        print(f"n={n:9d} CG {name:8s} {best:7.2f} us/it", flush=True)
    api.set_cg_schedule(api.CG_AUTO)
    best = 1e9
    for rep in range(4):
        m = torch.zeros_like(xt); torch.cuda.synchronize(); api.synchronize()
        t0 = time.perf_counter()
        info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, api.lcg_default_parameters(epsilon=1e-300, max_iterations=500), A)
        api.synchronize(); best = min(best, (time.perf_counter() - t0) / info.iterations * 1e6)
    print(f"n={n:9d} PCG+Jacobi     {best:7.2f} us/it", flush=True)
    A.destroy()
