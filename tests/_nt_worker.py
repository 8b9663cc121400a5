"""Worker of tests/test_gpu_solvers.py::test_cache_policy_of_the_vector_passes_changes_no_bit: runs CG, PCG + Jacobi, CGS and BiCGStab for a
fixed number of iterations on generated systems and prints the SHA-256 of every iterate.  The parent runs it with LCG_HIP_NT_VECTORS=0 and
=1 (the policy is read once per process): non-temporal loads and stores are a cache hint, the arithmetic is the same."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from liblcg_amd import _lib, api

lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
n = 300_007
for sym, runs in ((True, (("cg", api.LCG_CG, False), ("pcg", api.LCG_PCG, True), ("cgs", api.LCG_CGS, False))),
                  (False, (("bicgstab", api.LCG_BICGSTAB, False), ("cgs_nonsym", api.LCG_CGS, False)))):
    A = api.CsrMatrix.generate(n, 16, 5000, sym, 7, 0.01, pattern=api.GEN_DIAGONALS)
    A.build_jacobi()
    xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 7, 0, n, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    for name, sid, jac in runs:
        for sched in ((api.CG_CLASSIC, api.CG_ONE_REDUCTION) if name in ("cg", "pcg") else (api.CG_AUTO,)):
            api.set_cg_schedule(sched)
            m = torch.zeros_like(xt)
            para = api.lcg_default_parameters(epsilon=1e-300, max_iterations=25)
            if jac:
                info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, para, A)
            else:
                info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, sid)
            api.synchronize()
            print("iterate", name, sched, info.iterations, hashlib.sha256(m.cpu().numpy().tobytes()).hexdigest(), flush=True)
    api.set_cg_schedule(api.CG_AUTO)
    A.destroy()
