"""A/B of the launch-bound cases between two checkouts on the SAME box (boxes of the pool differ by 10-25 %
on these): best-of-N us/iteration of CG / CGS / BiCGStab on case_10K_A and PCG+Jacobi on the 1000x1000 Laplacian.

  git worktree add .ab_old <commit> && make -C .ab_old/liblcg_amd/csrc
  gpurun -- 'python scripts/ab_small.py $GRAFT_REPO_ROOT/.ab_old; python scripts/ab_small.py $GRAFT_REPO_ROOT'
"""
import json, os, sys, time
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
import numpy as np, torch
from liblcg_amd import _lib, api
from liblcg_amd.coo_io import read_coo_system, coo_to_csr_host
lib = _lib.load()
G = os.path.join(ROOT, "tests", "golden")
n, row, col, val, b = read_coo_system(os.path.join(G, "case_10K_A"))
rp, ci, v = coo_to_csr_host(n, row, col, val)
A = api.CsrMatrix.from_csr(rp, ci, v); A.build_jacobi()
bd = torch.from_numpy(b).cuda()
para = api.lcg_default_parameters(epsilon=1e-10, abs_diff=1)
out = {}
for name, sid in (("cg", api.LCG_CG), ("cgs", api.LCG_CGS), ("bicgstab", api.LCG_BICGSTAB)):
    best = 1e9
    for rep in range(6):
        m = torch.zeros(n, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize(); api.synchronize()
        t0 = time.perf_counter()
        info = api.lcg_solver("lcg_hip_csr_ax", None, m, bd, n, para, A, sid)
        api.synchronize()
        dt = time.perf_counter() - t0
        best = min(best, dt / info.iterations * 1e6)
    out[name] = round(best, 2)
# 1M Laplacian PCG
A2 = api.CsrMatrix.laplace2d(1000, 1000); A2.build_jacobi()
N = 1000000
xt = torch.empty(N, dtype=torch.float64, device="cuda"); api.gen_xtrue(N, 1, 0, N, xt)
b2 = torch.empty_like(xt); A2.spmv(xt, b2); api.synchronize()
p = api.lcg_default_parameters(epsilon=1e-300, max_iterations=500)
best = 1e9
for rep in range(4):
    m = torch.zeros_like(xt); torch.cuda.synchronize(); api.synchronize()
    t0 = time.perf_counter()
    info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b2, N, p, A2)
    api.synchronize()
    best = min(best, (time.perf_counter() - t0) / 500 * 1e6)
out["lap_pcg_us"] = round(best, 2)
print(os.path.basename(ROOT) or ROOT, json.dumps(out))
