"""Helper process of tests/test_gpu_direct.py (not collected by pytest): rank RANK of WORLD_SIZE
processes on GPU 0 running the REAL multi-rank sharded path with no RCCL at all -- sums over the
peer mailboxes, x over the direct neighbour exchange (dist mode 2).  Every rank owns its row block
of a generated system, multiplies and solves, and compares its slice with the single-process
results the parent test computed (REF.npz).

usage: RANK=r WORLD_SIZE=p MASTER_PORT=... python tests/_direct_worker.py REF.npz OUT.json
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(ref_path, out_path):
    import torch
    import torch.distributed as dist
    from liblcg_amd import _lib, api, partition

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    lib = _lib.load()
    assert lib.lcg_hip_init(0) == 0
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok, why = partition.init_p2p_from_torch(lib, rounds=16)
    assert ok, why
    assert lib.lcg_hip_comm_size() == 1            # no RCCL communicator anywhere in this test
    ref = np.load(ref_path)
    res = {"rank": rank}

    verbose = bool(os.environ.get("LCG_DIRECT_VERBOSE"))

    def say(*a):
        if verbose:
            print(f"[rank {rank}]", *a, file=sys.stderr, flush=True)

    for tag in ("band", "scr", "nsym", "rrb"):
        if f"{tag}/n" not in ref.files:
            continue
        n = int(ref[f"{tag}/n"]); band = int(ref[f"{tag}/band"]); sym = bool(ref[f"{tag}/sym"])
        r0, r1 = partition.shard_range(n, world, rank)
        A = api.CsrMatrix.generate(n, 16, band, sym, 3, 0.01, r0, r1, pattern=api.GEN_ROW_RANDOM_BAND if tag == "rrb" else None)
        A.distribute(n, 2)
        if tag == "rrb":        # rows that draw their own columns: the tiled product with pushing blocks (auto only from 4M entries up)
            assert lib.lcg_hip_csr_set_tiled(A.h, 1) == 0
        elif tag != "scr":      # also drive the packed-column product with pushing blocks (auto only from 4M entries up)
            assert lib.lcg_hip_csr_set_packed(A.h, 1) == 0
        res[f"{tag}/recv"] = int(lib.lcg_hip_csr_exchange_volume(A.h))
        x1 = torch.from_numpy(ref[f"{tag}/x1"][r0:r1]).cuda()
        x2 = torch.from_numpy(ref[f"{tag}/x2"][r0:r1]).cuda()
        y = torch.empty_like(x1)
        errs = []
        # alternate two inputs back to back, no reduction in between: a stale or overtaken receive
        # buffer (the neighbours run ahead or behind by one call) would show here
        for it in range(12):
            xin, want = (x1, ref[f"{tag}/y1"]) if it % 2 == 0 else (x2, ref[f"{tag}/y2"])
            t0 = time.perf_counter()
            A.spmv(xin, y)
            api.synchronize()
            if time.perf_counter() - t0 > 0.5:
                say(tag, f"call {it} took {time.perf_counter() - t0:.2f} s")
            w = want[r0:r1]
            errs.append(float(np.abs(y.cpu().numpy() - w).max() / np.abs(w).max()))
        res[f"{tag}/spmv_err"] = max(errs)
        res[f"{tag}/kernel"] = lib.lcg_hip_csr_last_kernel(A.h).decode()
        say(tag, "products: max err", max(errs), "per call", ["%.1e" % e for e in errs], "p2p status", lib.lcg_hip_p2p_status())
        b = torch.from_numpy(ref[f"{tag}/b"][r0:r1]).cuda()
        para = api.lcg_default_parameters(epsilon=1e-10, abs_diff=1)
        for name, sid in (("cg", api.LCG_CG), ("bicgstab", api.LCG_BICGSTAB), ("cgs", api.LCG_CGS)):
            if name == "cg" and not sym:
                continue
            m = torch.zeros(r1 - r0, dtype=torch.float64, device="cuda")
            say(tag, name, "...")
            info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, r1 - r0, para, A, sid)
            xt = ref[f"{tag}/x1"][r0:r1]
            res[f"{tag}/{name}"] = [int(info.ret), int(info.iterations), float(np.abs(m.cpu().numpy() - xt).max())]
            say(tag, name, res[f"{tag}/{name}"])
        if sym:
            A.build_jacobi()
            m = torch.zeros(r1 - r0, dtype=torch.float64, device="cuda")
            info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, r1 - r0, para, A)
            res[f"{tag}/pcg"] = [int(info.ret), int(info.iterations), float(np.abs(m.cpu().numpy() - ref[f"{tag}/x1"][r0:r1]).max())]
        assert lib.lcg_hip_barrier() == 0
        dist.barrier()
        A.destroy()
    if "cplx/x1" not in ref.files:      # the full-size run of tests/test_gpu_config3.py carries no complex case
        dist.barrier()
        lib.lcg_hip_p2p_disconnect()
        dist.destroy_process_group()
        json.dump(res, open(out_path, "w"))
        return
    # complex system (bundled case_10K_cA, complex symmetric): row slice with GLOBAL columns, 16-byte
    # elements through the pushes, the landing zone and the remote-column product
    from liblcg_amd.coo_io import coo_to_csr_host, read_coo_system
    nc, row, col, val, bc = read_coo_system(os.path.join(ROOT, "tests", "golden", "case_10K_cA"), True)
    rp, ci, v = coo_to_csr_host(nc, row, col, val)
    r0, r1 = partition.shard_range(nc, world, rank)
    lo, hi = int(rp[r0]), int(rp[r1])
    Ac = api.CsrMatrix.from_csr((rp[r0:r1 + 1] - rp[r0]).astype(np.int32), ci[lo:hi], v[lo:hi], n_cols=nc)
    Ac.distribute(nc, 2)
    xc = torch.from_numpy(ref["cplx/x1"][r0:r1]).cuda()
    yc = torch.empty_like(xc)
    Ac.spmv(xc, yc); api.synchronize()
    w = ref["cplx/y1"][r0:r1]
    res["cplx/spmv_err"] = float(np.abs(yc.cpu().numpy() - w).max() / np.abs(w).max())
    bcd = torch.from_numpy(bc[r0:r1]).cuda()
    cpara = api.clcg_default_parameters(epsilon=1e-10, abs_diff=1)
    for name, sid in (("bicg_sym", api.CLCG_BICG_SYM), ("tfqmr", api.CLCG_TFQMR)):
        m = torch.zeros(r1 - r0, dtype=torch.complex128, device="cuda")
        info = api.clcg_solver("clcg_hip_csr_ax", None, m, bcd, r1 - r0, cpara, Ac, sid, shadow_seed=7)
        xs = ref["cplx/xsol"][r0:r1]
        res[f"cplx/{name}"] = [int(info.ret), int(info.iterations), float(np.abs(m.cpu().numpy() - xs).max())]
    assert lib.lcg_hip_barrier() == 0
    dist.barrier()
    Ac.destroy()
    dist.barrier()
    lib.lcg_hip_p2p_disconnect()
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
