"""Same-box A/B of the solvers' vector passes between two checkouts (boxes of the pool differ by more than the effect):
CG (automatic schedule) and PCG + Jacobi on 5-point Laplacians of 1M / 2M / 4M rows and plain CG on the 10M-row headline
system, best of N capped solves, us per iteration.

  git worktree add .ab_old <commit> && make -C .ab_old/liblcg_amd/csrc
  gpurun -- 'for i in 1 2 3; do python scripts/ab_vec.py $GRAFT_REPO_ROOT/.ab_old; python scripts/ab_vec.py $GRAFT_REPO_ROOT; done'
"""
import json, os, sys, time
ROOT = os.path.abspath(sys.argv[1])
BIG = "--no-big" not in sys.argv
sys.path.insert(0, ROOT)
import torch
from liblcg_amd import _lib, api
lib = _lib.load()
assert os.path.abspath(_lib.SO_PATH).startswith(ROOT), _lib.SO_PATH
out = {}


def timed(solve, its, reps):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); api.synchronize()
        t0 = time.perf_counter()
        solve()
        api.synchronize()
        best = min(best, (time.perf_counter() - t0) / its * 1e6)
    return round(best, 2)


for side in (1000, 1414, 2000):
    A = api.CsrMatrix.laplace2d(side, side); A.build_jacobi()
    N = side * side
    xt = torch.empty(N, dtype=torch.float64, device="cuda"); api.gen_xtrue(N, 1, 0, N, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    p = api.lcg_default_parameters(epsilon=1e-300, max_iterations=400)
    m = torch.zeros_like(xt)
    out[f"lap{side}_cg"] = timed(lambda: (m.zero_(), api.lcg_solver("lcg_hip_csr_ax", None, m, b, N, p, A, api.LCG_CG)), 400, 5)
    out[f"lap{side}_pcg"] = timed(lambda: (m.zero_(), api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, N, p, A)), 400, 5)
    del A, xt, b, m
if BIG:
    N = 10000000
    A = api.CsrMatrix.generate(N, 16, 131072, True, 1, 0.01, pattern=api.GEN_DIAGONALS)
    xt = torch.empty(N, dtype=torch.float64, device="cuda"); api.gen_xtrue(N, 1, 0, N, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    m = torch.zeros_like(xt)
    p = api.lcg_default_parameters(epsilon=1e-300, max_iterations=100)
    for sched, name in ((api.CG_CLASSIC, "classic"), (api.CG_ONE_REDUCTION, "one_red")):
        api.set_cg_schedule(sched)
        out[f"big_cg_{name}"] = timed(lambda: (m.zero_(), api.lcg("lcg_hip_csr_ax", None, m, b, N, p, A)), 100, 6)
print(os.path.basename(ROOT) or ROOT, json.dumps(out), flush=True)
