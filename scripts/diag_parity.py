#!/usr/bin/env python3
"""Print GPU-vs-oracle parity numbers for every solver on the bundled systems (diagnostic)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from liblcg_amd import api
from liblcg_amd.coo_io import coo_to_csr_host, read_coo_system, read_solution
from oracle import pyoracle as po

G = os.path.join(ROOT, "tests", "golden")
port = po.Oracle("port")


def main():
    n, row, col, val, b = read_coo_system(os.path.join(G, "case_10K_A"))
    xs = read_solution(os.path.join(G, "case_10K_B"))
    rp, ci, v = coo_to_csr_host(n, row, col, val)
    A = api.CsrMatrix.from_csr(rp, ci, v); A.build_jacobi()
    bd = torch.from_numpy(b).cuda()
    for eps, ad in ((1e-12, 1), (1e-6, 0)):
        for sid, name in ((0, "cg"), (1, "pcg"), (2, "cgs"), (3, "bicgstab")):
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            p = api.lcg_default_parameters(epsilon=eps, abs_diff=ad)
            if sid == 1:
                info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, bd, n, p, A)
            else:
                info = api.lcg_solver("lcg_hip_csr_ax", None, m, bd, n, p, A, sid)
            ref = port.solve(sid, rp, ci, v, b, para=po.default_para(epsilon=eps, abs_diff=ad), jacobi=(sid == 1))
            x = m.cpu().numpy()
            print(f"real {name:9s} eps={eps:g}: gpu ret={info.ret} it={info.iterations} | oracle it={ref['iters']} | "
                  f"rel(x-oracle)={np.linalg.norm(x - ref['x']) / np.linalg.norm(ref['x']):.2e} |x-x*|={np.linalg.norm(x - xs):.2e} "
                  f"oracle |x-x*|={np.linalg.norm(ref['x'] - xs):.2e}")
    for case in ("1K", "10K"):
        n, row, col, val, b = read_coo_system(os.path.join(G, f"case_{case}_cA"), True)
        xs = read_solution(os.path.join(G, f"case_{case}_cB"), True)
        rp, ci, v = coo_to_csr_host(n, row, col, val)
        A = api.CsrMatrix.from_csr(rp, ci, v)
        bd = torch.from_numpy(b).cuda()
        rb = port.vecrnd(n, 42)
        for sid, name in ((1, "bicgsym"), (2, "cgs"), (3, "bicgstab"), (4, "tfqmr")):
            for maxit in (12, 0):
                mi = maxit if maxit else (300 if sid == 3 else 0)
                m = torch.zeros(n, dtype=torch.complex128, device="cuda")
                info = api.clcg_solver("clcg_hip_csr_ax", None, m, bd, n,
                                       api.clcg_default_parameters(epsilon=1e-10, abs_diff=1, max_iterations=mi), A, sid, shadow=rb)
                ref = port.csolve(sid, rp, ci, v, b, para=po.default_cpara(epsilon=1e-10, abs_diff=1, max_iterations=mi), rbar0=rb)
                x = m.cpu().numpy()
                print(f"cplx {case} {name:8s} max={mi}: gpu ret={info.ret} it={info.iterations} res={info.residual:.3e} | "
                      f"oracle ret={ref['ret']} it={ref['iters']} res={ref['residual']:.3e} | "
                      f"rel(x-oracle)={np.linalg.norm(x - ref['x']) / np.linalg.norm(ref['x']):.2e} |x-x*|={np.linalg.norm(x - xs):.2e} "
                      f"oracle |x-x*|={np.linalg.norm(ref['x'] - xs):.2e}")
    # blas1 detail
    rng = np.random.default_rng(5)
    for nn in (1000, 524289, 3_000_001):
        a = rng.standard_normal(nn); bb = rng.standard_normal(nn)
        ad, bdv = torch.from_numpy(a).cuda(), torch.from_numpy(bb).cuda()
        exact = float(np.dot(a.astype(np.longdouble), bb.astype(np.longdouble)))
        s = float(np.dot(np.abs(a), np.abs(bb)))
        print(f"dot n={nn}: gpu-exact={(api.dot(ad, bdv) - exact) / s:.2e} serial-exact={(port.dot(a, bb) - exact) / s:.2e} "
              f"unaligned gpu-exact={(api.dot(ad[1:], bdv[1:]) - float(np.dot(a[1:].astype(np.longdouble), bb[1:].astype(np.longdouble)))) / s:.2e} "
              f"nrm2 rel={(api.nrm2(ad) - np.linalg.norm(a)) / np.linalg.norm(a):.2e}")


if __name__ == "__main__":
    main()
