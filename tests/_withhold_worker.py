"""Helper process of tests/test_gpu_direct.py::test_a_rank_that_withholds_its_pushes (not collected by pytest).

Rank RANK of WORLD_SIZE processes on GPU 0, sums over the peer mailboxes, x over the direct neighbour exchange (mode 2).
The rank named by WITHHOLD runs with LCG_HIP_TEST_WITHHOLD_PUSH: it multiplies but never writes its boundary entries of x
to its neighbours nor raises its flag -- a dead link.  Expected on EVERY rank: the solve ends with LCG_HIP_E_COMM after
the time-out (nothing hangs), the ranks agree to tear the direct paths down, and after connecting anew (without the
fault) the same exchange works again.

usage: RANK=r WORLD_SIZE=p WITHHOLD=q MASTER_PORT=... python tests/_withhold_worker.py OUT.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    faulty = int(os.environ["WITHHOLD"]) == rank
    phase2 = os.environ.get("PHASE") == "2"
    if faulty and not phase2:
        os.environ["LCG_HIP_TEST_WITHHOLD_PUSH"] = "1"
    import torch
    import torch.distributed as dist
    from liblcg_amd import _lib, api, partition

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    lib = _lib.load()
    assert lib.lcg_hip_init(0) == 0
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok, why = partition.init_p2p_from_torch(lib, rounds=8)
    assert ok, why
    assert lib.lcg_hip_p2p_set_timeout_ms(1500) == 0
    n, band = 30000, 400
    r0, r1 = partition.shard_range(n, world, rank)
    A = api.CsrMatrix.generate(n, 16, band, True, 3, 0.01, r0, r1)
    A.distribute(n, 2)
    xt = torch.empty(r1 - r0, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 3, r0, r1, xt)
    b = torch.empty_like(xt)
    res = {"rank": rank, "faulty": faulty, "phase": 2 if phase2 else 1}
    t0 = time.perf_counter()
    rc_spmv = lib.lcg_hip_spmv(A.h, xt.data_ptr(), b.data_ptr())
    rc_sync = lib.lcg_hip_synchronize()
    res["product"] = [rc_spmv, rc_sync, lib.lcg_hip_p2p_status()]
    m = torch.zeros_like(xt)
    p = api.lcg_default_parameters(epsilon=1e-10, abs_diff=1, max_iterations=50)
    import ctypes as C
    rc = lib.lcg_hip_lcg(_lib.fnptr(lib, "lcg_hip_csr_ax"), None, m.data_ptr(), b.data_ptr(), r1 - r0, C.byref(p), A.h, None, None, None, 1)
    res["solve_rc"] = rc
    res["seconds"] = time.perf_counter() - t0
    res["status_after"] = lib.lcg_hip_p2p_status()
    # the vote bench.py takes: anybody's failure switches the direct paths off everywhere
    t = torch.tensor([1 if lib.lcg_hip_p2p_status() < 0 else 0], dtype=torch.int32)
    dist.all_reduce(t)
    res["votes"] = int(t.item())
    A.destroy()
    lib.lcg_hip_p2p_disconnect()
    dist.barrier()
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main(sys.argv[1])
