"""Helper process of tests/test_gpu_rccl_ranks.py (not collected by pytest): rank RANK of WORLD_SIZE processes on GPU 0
running the NORTH-STAR sharded path -- lcg_hip_csr_distribute modes 0 and 1: ncclAllGather / grouped ncclSend+ncclRecv of x on
the second stream beside the local-column product, ncclAllReduce of the dots inside the lock-step loops, ncclReduceScatter of
op(A).x -- with MORE THAN ONE RANK.  The collectives come from tests/fake_rccl (LCG_HIP_RCCL_LIB; the real RCCL refuses two ranks
on one device), everything else is the product's own code.  Every rank compares ITS rows with the oracle's / the single-process
results the parent test put into REFDIR (one .npy per array, memory-mapped).

usage: RANK=r WORLD_SIZE=p MASTER_PORT=... LCG_HIP_RCCL_LIB=.../librccl_fake.so python tests/_rccl_worker.py REFDIR OUT.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Ref:
    def __init__(self, d):
        self.d = d
        self.meta = json.load(open(os.path.join(d, "meta.json")))

    def has(self, key):
        return os.path.exists(os.path.join(self.d, key.replace("/", "__") + ".npy"))

    def __getitem__(self, key):
        return np.load(os.path.join(self.d, key.replace("/", "__") + ".npy"), mmap_mode="r")


def main(ref_dir, out_path):
    import torch
    import torch.distributed as dist
    from liblcg_amd import _lib, api, partition

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("LCG_RCCL_WATCHDOG_S"):       # a rank that is still here then says where it stands (all threads) before the test ends it
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["LCG_RCCL_WATCHDOG_S"]), exit=False)
    lib = _lib.load()
    assert lib.lcg_hip_init(0) == 0
    dist.init_process_group("gloo", rank=rank, world_size=world)
    partition.init_comm_from_torch(lib)
    assert lib.lcg_hip_comm_size() == world and lib.lcg_hip_comm_rank() == rank
    assert b"librccl_fake" in lib.lcg_hip_comm_library(), lib.lcg_hip_comm_library()
    ref = Ref(ref_dir)
    res = {"rank": rank, "library": lib.lcg_hip_comm_library().decode()}
    verbose = bool(os.environ.get("LCG_RCCL_VERBOSE"))

    import time
    t_start = time.time()

    def say(*a):
        if verbose:
            print(f"[rank {rank} +{time.time() - t_start:7.2f}s]", *a, file=sys.stderr, flush=True)

    def worst_rowwise(y, want, bound):
        return float(np.max(np.abs(y - want) / bound)) if len(want) else 0.0

    def rel_max(x, want):
        return float(np.max(np.abs(x - want)) / max(np.max(np.abs(want)), 1e-300)) if len(want) else 0.0

    solvers = (("cg", api.LCG_CG, api.CG_AUTO), ("cg_classic", api.LCG_CG, api.CG_CLASSIC), ("pcg", api.LCG_PCG, api.CG_AUTO),
               ("bicgstab", api.LCG_BICGSTAB, api.CG_AUTO), ("cgs", api.LCG_CGS, api.CG_AUTO))

    def solve(A, sid, sched, m, b, nloc, para):
        api.set_cg_schedule(sched)
        try:
            if sid == api.LCG_PCG:
                return api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, nloc, para, A)
            return api.lcg_solver("lcg_hip_csr_ax", None, m, b, nloc, para, A, sid)
        finally:
            api.set_cg_schedule(api.CG_AUTO)

    for case in ref.meta["cases"]:
        tag, n, band, sym, pattern = case["tag"], case["n"], case["band"], case["sym"], case["pattern"]
        r0, r1 = partition.shard_range(n, world, rank)
        nloc = r1 - r0
        say(tag, "generating rows", r0, r1)
        A = api.CsrMatrix.generate(n, 16, band, sym, case["seed"], 0.01, r0, r1, pattern=pattern)
        if sym:
            A.build_jacobi()
        api.synchronize()
        say(tag, "generated")
        x1 = torch.from_numpy(np.array(ref[f"{tag}/x1"][r0:r1])).cuda()
        x2 = torch.from_numpy(np.array(ref[f"{tag}/x2"][r0:r1])).cuda()
        y = torch.empty_like(x1)
        b = torch.from_numpy(np.array(ref[f"{tag}/y1"][r0:r1])).cuda()
        products = {}
        for mode in (0, 1):
            A.distribute(n, mode)
            say(tag, "distributed, mode", mode)
            for which, setter in (("packed", lib.lcg_hip_csr_set_packed), ("tiled", lib.lcg_hip_csr_set_tiled)):
                if case.get("force") == which:      # small systems: drive the kernel family the full-size system selects by itself
                    assert setter(A.h, 1) == 0
            res[f"{tag}/m{mode}/recv"] = int(lib.lcg_hip_csr_exchange_volume(A.h))
            worst = 0.0
            # two inputs alternating back to back: a stale gather buffer or a receive overtaken by the next call would show
            for it in range(12):
                xin, key = (x1, "y1") if it % 2 == 0 else (x2, "y2")
                A.spmv(xin, y)
                api.synchronize()
                yh = y.cpu().numpy()
                worst = max(worst, worst_rowwise(yh, ref[f"{tag}/{key}"][r0:r1], ref[f"{tag}/bound{key[1]}"][r0:r1]))
                if it < 2:
                    products[(mode, key)] = yh
            res[f"{tag}/m{mode}/spmv_worst"] = worst
            res[f"{tag}/m{mode}/kernel"] = lib.lcg_hip_csr_last_kernel(A.h).decode()
            say(tag, "mode", mode, "products: worst row-wise", worst, res[f"{tag}/m{mode}/kernel"][:60])
            # four capped iterations of every real solver: this rank's rows of the iterate against the oracle's (liblcg's own loop,
            # restated) and against the single-process run of the product
            for name, sid, sched in solvers:
                if (not sym and sid in (api.LCG_CG, api.LCG_PCG)) or not ref.has(f"{tag}/{name.split('_')[0]}4_oracle"):
                    continue
                for ad in (1, 0):
                    m = torch.zeros(nloc, dtype=torch.float64, device="cuda")
                    info = solve(A, sid, sched, m, b, nloc, api.lcg_default_parameters(epsilon=1e-300, abs_diff=ad, max_iterations=4))
                    base = name.split("_")[0]
                    xo = ref[f"{tag}/{base}4_oracle"][r0:r1]
                    xs = ref[f"{tag}/{base}4_single"][r0:r1]
                    say(tag, mode, name, ad, "done")
                    res[f"{tag}/m{mode}/{name}4/ad{ad}"] = [int(info.ret), int(info.iterations), rel_max(m.cpu().numpy(), xo), rel_max(m.cpu().numpy(), xs),
                                                            float(info.residual)]
                    if mode == 1 or case.get("big"):
                        break
            if mode == 0 and not case.get("big"):
                # to convergence, in lock-step: counts against the single-process run, error against x_true (= x1)
                para = api.lcg_default_parameters(epsilon=1e-10, abs_diff=1)
                for name, sid, sched in solvers:
                    if not sym and sid in (api.LCG_CG, api.LCG_PCG):
                        continue
                    m = torch.zeros(nloc, dtype=torch.float64, device="cuda")
                    info = solve(A, sid, sched, m, b, nloc, para)
                    res[f"{tag}/{name}"] = [int(info.ret), int(info.iterations), float(np.abs(m.cpu().numpy() - np.array(ref[f"{tag}/x1"][r0:r1])).max())]
                    say(tag, name, res[f"{tag}/{name}"])
        # the neighbour-range product must be the all-gather product bit for bit (bench.py admits it only then)
        res[f"{tag}/m1_equals_m0"] = bool(all(np.array_equal(products[(0, k)], products[(1, k)]) for k in ("y1", "y2")))
        if ref.has(f"{tag}/yT"):
            # A^T.x: every rank multiplies the transpose of its rows, ncclReduceScatter sums the contributions to each row block
            assert lib.lcg_hip_spmv_op(A.h, x1.data_ptr(), y.data_ptr(), 1, 0) == 0, lib.lcg_hip_last_error()
            api.synchronize()
            res[f"{tag}/opT_worst"] = worst_rowwise(y.cpu().numpy(), ref[f"{tag}/yT"][r0:r1], ref[f"{tag}/boundT"][r0:r1])
        assert lib.lcg_hip_barrier() == 0
        dist.barrier()
        A.destroy()

    if ref.meta.get("complex"):
        # complex system (bundled case_10K_cA, complex symmetric): 16-byte elements through the all-gather, the grouped send/recv and
        # the reduce-scatter; clbicg's second product per iteration is A^H.d (clcg.cpp:187)
        from liblcg_amd.coo_io import coo_to_csr_host, read_coo_system
        nc, row, col, val, bc = read_coo_system(os.path.join(ROOT, "tests", "golden", "case_10K_cA"), True)
        rp, ci, v = coo_to_csr_host(nc, row, col, val)
        r0, r1 = partition.shard_range(nc, world, rank)
        lo, hi = int(rp[r0]), int(rp[r1])
        Ac = api.CsrMatrix.from_csr((rp[r0:r1 + 1] - rp[r0]).astype(np.int32), ci[lo:hi], v[lo:hi], n_cols=nc)
        xc = torch.from_numpy(np.array(ref["cplx/x1"][r0:r1])).cuda()
        yc = torch.empty_like(xc)
        bcd = torch.from_numpy(bc[r0:r1]).cuda()
        cpara = api.clcg_default_parameters(epsilon=1e-10, abs_diff=1)
        for mode in (0, 1):
            Ac.distribute(nc, mode)
            Ac.spmv(xc, yc); api.synchronize()
            w = ref["cplx/y1"][r0:r1]
            res[f"cplx/m{mode}/spmv_err"] = float(np.abs(yc.cpu().numpy() - w).max() / np.abs(w).max())
            for key, layout, conj in (("yH", 1, 1), ("yT", 1, 0)):
                assert lib.lcg_hip_spmv_op(Ac.h, xc.data_ptr(), yc.data_ptr(), layout, conj) == 0, lib.lcg_hip_last_error()
                api.synchronize()
                w = ref[f"cplx/{key}"][r0:r1]
                res[f"cplx/m{mode}/{key}_err"] = float(np.abs(yc.cpu().numpy() - w).max() / np.abs(w).max())
            for name, sid in (("bicg", api.CLCG_BICG), ("bicg_sym", api.CLCG_BICG_SYM), ("tfqmr", api.CLCG_TFQMR)):
                if mode == 1 and name != "bicg_sym":
                    continue
                m = torch.zeros(r1 - r0, dtype=torch.complex128, device="cuda")
                info = api.clcg_solver("clcg_hip_csr_ax", None, m, bcd, r1 - r0, cpara, Ac, sid, shadow_seed=7)
                xs = ref["cplx/xsol"][r0:r1]
                res[f"cplx/m{mode}/{name}"] = [int(info.ret), int(info.iterations), float(np.abs(m.cpu().numpy() - xs).max())]
                say("cplx", mode, name, res[f"cplx/m{mode}/{name}"])
        assert lib.lcg_hip_barrier() == 0
        dist.barrier()
        Ac.destroy()

    if ref.meta.get("mailbox_case"):
        # RCCL moves x, the peer-mapped mailboxes sum the dots (bench.py's "all-gather / neighbour ranges + direct all-reduce")
        case = next(c for c in ref.meta["cases"] if c["tag"] == ref.meta["mailbox_case"])
        tag, n = case["tag"], case["n"]
        ok, why = partition.init_p2p_from_torch(lib, rounds=8)
        assert ok, why
        r0, r1 = partition.shard_range(n, world, rank)
        A = api.CsrMatrix.generate(n, 16, case["band"], case["sym"], case["seed"], 0.01, r0, r1, pattern=case["pattern"])
        b = torch.from_numpy(np.array(ref[f"{tag}/y1"][r0:r1])).cuda()
        for mode in (0, 1):
            A.distribute(n, mode)
            m = torch.zeros(r1 - r0, dtype=torch.float64, device="cuda")
            info = solve(A, api.LCG_CG, api.CG_AUTO, m, b, r1 - r0, api.lcg_default_parameters(epsilon=1e-300, abs_diff=1, max_iterations=4))
            res[f"{tag}/mailbox/m{mode}/cg4"] = [int(info.ret), int(info.iterations), rel_max(m.cpu().numpy(), ref[f"{tag}/cg4_oracle"][r0:r1])]
        res["p2p_status"] = int(lib.lcg_hip_p2p_status())
        assert lib.lcg_hip_barrier() == 0
        dist.barrier()
        A.destroy()
        dist.barrier()
        lib.lcg_hip_p2p_disconnect()

    api.synchronize()
    dist.barrier()
    assert lib.lcg_hip_comm_destroy() == 0
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
