"""Helper process of tests/test_gpu_comm.py (not collected by pytest): solves the bundled systems
to CONVERGENCE and writes the outcomes to an .npz.  With LCG_HIP_FORCE_COMM=1 the matrices are
distributed over a one-rank RCCL communicator, so every solver runs its sharded loop: split A.x
with the x exchange, reduce / all-reduce / finish scalar steps, and the batched lock-step enqueue
(driver.hpp: run_lockstep) that keeps the ranks' collective counts equal.

usage: python tests/_sharded_case.py OUT.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def main(out_path):
    import torch
    from liblcg_amd import _lib, api, partition
    from liblcg_amd.coo_io import coo_to_csr_host, read_coo_system

    lib = _lib.load()
    assert lib.lcg_hip_init(0) == 0
    sharded = bool(os.environ.get("LCG_HIP_FORCE_COMM"))
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        partition.init_comm_from_torch(lib)
        if os.environ.get("LCG_HIP_P2P", "0") == "1":
            ok, why = partition.init_p2p_from_torch(lib)
            assert ok, why
    mode = int(os.environ.get("LCG_HIP_DIST_MODE", "1"))
    res = {}

    n, row, col, val, b = read_coo_system(os.path.join(GOLDEN, "case_10K_A"))
    rp, ci, v = coo_to_csr_host(n, row, col, val)
    A = api.CsrMatrix.from_csr(rp, ci, v)
    A.build_jacobi()
    if sharded:
        A.distribute(n, mode)
    bd = torch.from_numpy(b).cuda()
    para = api.lcg_default_parameters(epsilon=1e-12, abs_diff=1)
    for name, sid in (("cg", api.LCG_CG), ("pcg", api.LCG_PCG), ("cgs", api.LCG_CGS), ("bicgstab", api.LCG_BICGSTAB),
                      ("bicgstab2", api.LCG_BICGSTAB2)):
        m = torch.zeros(n, dtype=torch.float64, device="cuda")
        if sid == api.LCG_PCG:
            info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, bd, n, para, A)
        else:
            info = api.lcg_solver("lcg_hip_csr_ax", None, m, bd, n, para, A, sid)
        res[f"{name}/meta"] = np.array([info.ret, info.iterations, info.residual])
        res[f"{name}/x"] = m.cpu().numpy()
    # a short capped run (the count must be exact) and one with a progress callback
    m = torch.zeros(n, dtype=torch.float64, device="cuda")
    info = api.lcg_solver("lcg_hip_csr_ax", None, m, bd, n, api.lcg_default_parameters(epsilon=1e-12, abs_diff=1, max_iterations=13), A, api.LCG_CG)
    res["cg13/meta"] = np.array([info.ret, info.iterations, info.residual]); res["cg13/x"] = m.cpu().numpy()
    seen = []
    m = torch.zeros(n, dtype=torch.float64, device="cuda")
    info = api.lcg_solver("lcg_hip_csr_ax", lambda i, mp, c, p, nn, k: seen.append(k) or 0, m, bd, n,
                          api.lcg_default_parameters(epsilon=1e-6), A, api.LCG_CG)
    res["cgpfp/meta"] = np.array([info.ret, info.iterations, info.residual, len(seen)]); res["cgpfp/x"] = m.cpu().numpy()

    # real, non-symmetric, generated: A^T.x sharded against unsharded
    G = api.CsrMatrix.generate(30000, 16, 500, False, 4, 0.01)
    if sharded:
        G.distribute(30000, mode)
    xr = torch.empty(30000, dtype=torch.float64, device="cuda"); api.gen_xtrue(30000, 5, 0, 30000, xr)
    yr = torch.empty_like(xr)
    assert lib.lcg_hip_spmv_op(G.h, xr.data_ptr(), yr.data_ptr(), 1, 0) == 0, lib.lcg_hip_last_error()
    api.synchronize()
    res["opT_real/y"] = yr.cpu().numpy()
    G.spmv(xr, yr); api.synchronize()
    res["op_real/y"] = yr.cpu().numpy()
    G.destroy()

    nc, row, col, val, bc = read_coo_system(os.path.join(GOLDEN, "case_1K_cA"), True)
    rp, ci, v = coo_to_csr_host(nc, row, col, val)
    Ac = api.CsrMatrix.from_csr(rp, ci, v)
    if sharded:
        Ac.distribute(nc, mode)
    bcd = torch.from_numpy(bc).cuda()
    cpara = api.clcg_default_parameters(epsilon=1e-10, abs_diff=1)
    # A^H.x and A^T.x on the sharded matrix (every rank multiplies the transpose of its rows, ncclReduceScatter sums the
    # contributions): the product itself and clbicg (clcg.cpp:77-226), whose second product per iteration is A^H.d
    rng = np.random.default_rng(11)
    xc = torch.from_numpy(rng.standard_normal(nc) + 1j * rng.standard_normal(nc)).cuda()
    yc = torch.empty_like(xc)
    for tag, layout, conj in (("opH", 1, 1), ("opT", 1, 0)):
        assert lib.lcg_hip_spmv_op(Ac.h, xc.data_ptr(), yc.data_ptr(), layout, conj) == 0, lib.lcg_hip_last_error()
        api.synchronize()
        res[f"{tag}/y"] = yc.cpu().numpy()
    res["op_conj_only_rc"] = np.array(lib.lcg_hip_spmv_op(Ac.h, xc.data_ptr(), yc.data_ptr(), 0, 1))
    for name, sid in (("c_bicg", api.CLCG_BICG), ("c_bicg_sym", api.CLCG_BICG_SYM), ("c_cgs", api.CLCG_CGS), ("c_tfqmr", api.CLCG_TFQMR)):
        m = torch.zeros(nc, dtype=torch.complex128, device="cuda")
        info = api.clcg_solver("clcg_hip_csr_ax", None, m, bcd, nc, cpara, Ac, sid, shadow_seed=7)
        res[f"{name}/meta"] = np.array([info.ret, info.iterations, info.residual])
        res[f"{name}/x"] = m.cpu().numpy()
    api.synchronize()
    res["p2p_status"] = np.array(lib.lcg_hip_p2p_status())
    np.savez(out_path, **res)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
