#!/usr/bin/env python3
"""A.x carrying its dot (k_spmv_lds1d) against A.x + a dot pass, back to back on one box: mean time per call over `reps`
enqueued calls, for a few systems.  LCG_HIP_AX_DOT_DBG=1/2/3 knocks parts of the epilogue out (lab only: wrong sums)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from liblcg_amd import _lib, api
lib = _lib.load()
reps = int(os.environ.get("REPS", "100"))

def run(name, A, n):
    x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, x)
    y = torch.empty_like(x)
    res = (_lib.C.c_double * 2)()
    out = {"system": name, "rows": n}
    def t(fn):
        for _ in range(10): fn()
        api.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        api.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
    out["ax_us"] = round(t(lambda: lib.lcg_hip_spmv(A.h, x.data_ptr(), y.data_ptr())), 2)
    out["ax_dot_us"] = round(t(lambda: lib.lcg_hip_spmv_dot(A.h, x.data_ptr(), y.data_ptr(), x.data_ptr(), None)), 2)
    out["kernel"] = lib.lcg_hip_csr_last_kernel(A.h).decode()[:40]
    print(json.dumps(out), flush=True)

A = api.CsrMatrix.laplace2d(1000, 1000); run("laplace 1000^2", A, 1000000); A.destroy()
A = api.CsrMatrix.laplace2d(300, 300); run("laplace 300^2", A, 90000); A.destroy()
A = api.CsrMatrix.generate(200000, 16, 3000, True, 1, 0.01); run("diagonals 200K", A, 200000); A.destroy()
A = api.CsrMatrix.generate(1000000, 16, 30000, True, 1, 0.01); run("diagonals 1M", A, 1000000); A.destroy()
A = api.CsrMatrix.generate(10000000, 16, 131072, True, 1, 0.01); run("diagonals 10M (the headline)", A, 10000000); A.destroy()
