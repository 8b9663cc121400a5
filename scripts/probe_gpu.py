#!/usr/bin/env python3
"""First-contact GPU probe: smoke, generator parity, A.x variants, CG timing.
Writes a log to gpurun_out/probe.log (also stdout)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from liblcg_amd import api
from oracle import pyoracle as po

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
LOG = open(os.path.join(ROOT, "gpurun_out", "probe.log"), "a")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + "\n"); LOG.flush()


def spmv_bytes(n, nnz):
    return 12 * nnz + 4 * (n + 1) + 16 * n


def time_spmv(A, x, y, reps=20):
    torch.cuda.synchronize()
    for _ in range(3):
        A.spmv(x, y)
    api.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        A.spmv(x, y)
    api.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    what = sys.argv[1:] or ["smoke", "gen", "spmv", "cg"]
    log("== probe", what, torch.cuda.get_device_name(0))
    port = po.Oracle("port")
    if "smoke" in what:
        import __graft_entry__ as g
        g.smoke()
    if "gen" in what:
        for band in (64, 0):
            n = 5000
            g = port.gen_init(n, 16, band, True, 7, 0.01)
            rp, ci, v = port.gen_rows(g)
            A = api.CsrMatrix.generate(n, 16, band, True, 7, 0.01)
            rp2, ci2, v2 = A.arrays_to_host()
            log("gen band", band, "rowptr eq", np.array_equal(rp, rp2), "col eq", np.array_equal(ci, ci2),
                "val eq", np.array_equal(v, v2), "nnz", len(ci))
            x = torch.rand(n, dtype=torch.float64, device="cuda")
            y = torch.empty_like(x)
            for var in (0, -1, 4, 8, 16, 64):
                A.set_kernel(var)
                A.spmv(x, y); api.synchronize()
                yo = port.csr_matvec(rp, ci, v, x.cpu().numpy())
                log("  spmv variant", var, "max rel err", np.abs(y.cpu().numpy() - yo).max() / np.abs(yo).max())
    if "spmv" in what:
        n = int(os.environ.get("PROBE_N", 10_000_000))
        for band, tag in ((131072, "banded W=131072"), (0, "scrambled")):
            t0 = time.time()
            A = api.CsrMatrix.generate(n, 16, band, True, 1, 0.01)
            api.synchronize()
            nnz = A.nnz
            log(f"{tag}: n={n} nnz={nnz} gen {time.time() - t0:.2f}s, bytes/spmv {spmv_bytes(n, nnz) / 1e9:.3f} GB")
            x = torch.rand(n, dtype=torch.float64, device="cuda")
            y = torch.empty_like(x)
            ref = None
            for var in (-1, -32, -128, 4, 8, 16, 32, 64):
                A.set_kernel(var)
                t = time_spmv(A, x, y)
                if ref is None:
                    ref = y.clone()
                err = (y - ref).abs().max().item()
                log(f"  variant {var:4d}: {t * 1e3:8.3f} ms  {spmv_bytes(n, nnz) / t / 1e9:8.1f} GB/s  maxdiff {err:.1e}")
            A.destroy()
            del x, y
    if "cg" in what:
        n = int(os.environ.get("PROBE_N", 10_000_000))
        A = api.CsrMatrix.generate(n, 16, 131072, True, 1, 0.01)
        xt = torch.empty(n, dtype=torch.float64, device="cuda")
        api.gen_xtrue(n, 1, 0, n, xt)
        b = torch.empty_like(xt)
        A.spmv(xt, b); api.synchronize()
        L = __import__("liblcg_amd._lib", fromlist=["x"]).load()
        L.lcg_hip_set_profiling(1)
        for iters in (20, 200):
            m = torch.zeros_like(xt)
            p = api.lcg_default_parameters(epsilon=1e-300, max_iterations=iters)
            torch.cuda.synchronize(); t0 = time.time()
            info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, p, A, api.LCG_CG)
            torch.cuda.synchronize(); dt = time.time() - t0
            log(f"CG {iters} its: ret={info.ret} it={info.iterations} resid={info.residual:.3e} {dt * 1e3:.1f} ms "
                f"-> {iters / dt:.1f} it/s; A.x mean {L.lcg_hip_last_ax_mean_us():.1f} us over {L.lcg_hip_last_ax_calls()} calls; "
                f"err vs x_true {(m - xt).norm().item() / xt.norm().item():.3e}")
        # converge
        m = torch.zeros_like(xt)
        p = api.lcg_default_parameters(epsilon=1e-10, abs_diff=1)
        t0 = time.time()
        info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, p, A, api.LCG_CG)
        torch.cuda.synchronize(); dt = time.time() - t0
        log(f"CG to 1e-10: ret={info.ret} it={info.iterations} resid={info.residual:.3e} {dt * 1e3:.1f} ms "
            f"relerr {(m - xt).norm().item() / xt.norm().item():.3e}")


if __name__ == "__main__":
    main()
