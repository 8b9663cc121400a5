"""Row partition used by the sharded solver (host logic, no GPU needed).

Rank r of P owns the contiguous rows [r*rpr, min(N, (r+1)*rpr)) with rpr = ceil(N/P) -- the
same rule ``lcg_hip_csr_distribute`` applies on the device side (comm.hip: dist_split).  Only
the last rank can be short, so the all-gathered vector of P*rpr entries is addressed by
GLOBAL column indices without remapping.
"""
from __future__ import annotations


def rows_per_rank(n_global: int, nranks: int) -> int:
    return (n_global + nranks - 1) // nranks


def shard_range(n_global: int, nranks: int, rank: int):
    rpr = rows_per_rank(n_global, nranks)
    r0 = min(n_global, rank * rpr)
    r1 = min(n_global, r0 + rpr)
    return r0, r1


def gathered_length(n_global: int, nranks: int) -> int:
    return rows_per_rank(n_global, nranks) * nranks


def init_comm_from_torch(lib=None):
    """Create the library's RCCL communicator using torch.distributed only as the courier of
    the 128-byte unique id (rank 0 creates it, everybody receives it)."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    from . import _lib as L
    lib = lib or L.load()
    rank, world = dist.get_rank(), dist.get_world_size()
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        rc = lib.lcg_hip_comm_unique_id(buf)
        if rc:
            raise RuntimeError(f"lcg_hip_comm_unique_id rc={rc}: {lib.lcg_hip_last_error().decode()}")
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(list(buf), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=0)
    raw = bytes(t.cpu().tolist())
    idb = (C.c_ubyte * 128).from_buffer_copy(raw)
    rc = lib.lcg_hip_comm_init(world, rank, idb)
    if rc:
        raise RuntimeError(f"lcg_hip_comm_init rc={rc}: {lib.lcg_hip_last_error().decode()}")
    return rank, world


P2P_HANDLE_BYTES = 64


def init_p2p_from_torch(lib=None, rounds=32):
    """Connect the direct all-reduce (lcg_hip.h: lcg_hip_p2p_*): every rank exports the IPC handle
    of its mailbox, torch.distributed carries the handles, every rank maps its peers' mailboxes
    and all run the self-test together.  The path is enabled only if EVERY rank succeeded at every
    step (the ranks agree through a MIN all-reduce after each), otherwise it is torn down on all
    of them and RCCL keeps doing the all-reduces.  Returns (enabled, reason)."""
    import ctypes as C

    import torch
    import torch.distributed as dist

    from . import _lib as L
    lib = lib or L.load()
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"

    def all_ok(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def why(step):
        return f"{step}: {lib.lcg_hip_last_error().decode()}"

    if world > 64:
        return False, "more than 64 ranks"
    h = (C.c_ubyte * P2P_HANDLE_BYTES)()
    rc = lib.lcg_hip_p2p_export(h)
    reason = why("export") if rc else ""
    mine = torch.tensor(list(h), dtype=torch.uint8, device=dev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if not all_ok(rc == 0):
        lib.lcg_hip_p2p_disconnect()
        return False, reason or "export failed on a peer"
    raw = b"".join(bytes(g.cpu().tolist()) for g in gathered)
    buf = (C.c_ubyte * len(raw)).from_buffer_copy(raw)
    rc = lib.lcg_hip_p2p_connect(world, rank, buf)
    reason = why("connect") if rc else ""
    if not all_ok(rc == 0):
        lib.lcg_hip_p2p_disconnect()
        return False, reason or "connect failed on a peer"
    rc = lib.lcg_hip_p2p_selftest(rounds)
    reason = why("self-test") if rc else ""
    if not all_ok(rc == 0):
        lib.lcg_hip_p2p_disconnect()
        return False, reason or "self-test failed on a peer"
    lib.lcg_hip_p2p_enable(1)
    return True, "ok"
