"""Helper process of tests/test_gpu_p2p.py (not collected by pytest): rank RANK of WORLD_SIZE
processes that all use GPU 0.  The ranks rendezvous over gloo (CPU tensors), map each other's
mailboxes through HIP IPC and run the direct all-reduce; every result is checked against the
sum, in rank order, of the inputs all ranks contributed (gathered over gloo).

usage: RANK=r WORLD_SIZE=p MASTER_PORT=... python tests/_p2p_worker.py OUT.json
       python tests/_p2p_worker.py --silent     export a mailbox, print the handle, never write to anybody
       python tests/_p2p_worker.py --lonely     rank 0 of 2 whose peer is a --silent process: must time out
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    import torch
    import torch.distributed as dist
    from liblcg_amd import _lib, partition

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    lib = _lib.load()
    assert lib.lcg_hip_init(0) == 0
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok, why = partition.init_p2p_from_torch(lib, rounds=48)
    res = {"rank": rank, "enabled": ok, "why": why, "status": lib.lcg_hip_p2p_status(), "mismatch": 0, "calls": 0}
    if ok:
        rng = np.random.default_rng(100 + rank)
        for it in range(60):
            cnt = 1 + it % 8
            mine = rng.standard_normal(cnt) * 10.0 ** rng.integers(-8, 8)
            t = torch.from_numpy(mine).cuda()
            assert lib.lcg_hip_allreduce_sum(t.data_ptr(), cnt) == 0
            assert lib.lcg_hip_synchronize() == 0
            got = t.cpu().numpy()
            everyone = [torch.empty(cnt, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(everyone, torch.from_numpy(mine))
            want = np.zeros(cnt)
            for q in range(world):          # rank order, like the kernel
                want = want + everyone[q].numpy()
            res["mismatch"] += int(not np.array_equal(got, want))
            res["calls"] += 1
        assert lib.lcg_hip_barrier() == 0
        dist.barrier()
        lib.lcg_hip_p2p_disconnect()
    dist.destroy_process_group()
    json.dump(res, open(out_path, "w"))


def silent():
    from liblcg_amd import _lib
    lib = _lib.load()
    assert lib.lcg_hip_init(0) == 0
    h = (C.c_ubyte * 64)()
    assert lib.lcg_hip_p2p_export(h) == 0
    sys.stdout.write(bytes(h).hex() + "\n")
    sys.stdout.flush()
    sys.stdin.readline()        # keep the allocation alive until told to leave


def lonely():
    import subprocess
    import time
    os.environ["LCG_HIP_P2P_TIMEOUT_MS"] = "300"
    from liblcg_amd import _lib
    lib = _lib.load()
    assert lib.lcg_hip_init(0) == 0
    h = (C.c_ubyte * 64)()
    assert lib.lcg_hip_p2p_export(h) == 0
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--silent"], stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, text=True)
    try:
        peer = bytes.fromhex(child.stdout.readline().strip())
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(h) + peer)
        assert lib.lcg_hip_p2p_connect(2, 0, buf) == 0, lib.lcg_hip_last_error()
        t0 = time.time()
        rc = lib.lcg_hip_p2p_selftest(8)
        dt = time.time() - t0
        print("RC", rc, "DT", round(dt, 2), lib.lcg_hip_last_error().decode())
        assert rc == -2002 and dt < 5.0
        assert lib.lcg_hip_p2p_status() == -1       # the path reports itself dead
        assert lib.lcg_hip_synchronize() == -2002   # ... and that is what a caller's synchronisation returns
        lib.lcg_hip_p2p_disconnect()
    finally:
        child.stdin.write("bye\n"); child.stdin.flush(); child.wait(timeout=30)


if __name__ == "__main__":
    if sys.argv[1] == "--silent":
        silent()
    elif sys.argv[1] == "--lonely":
        lonely()
    else:
        main(sys.argv[1])
