"""world_size-2 (and 3) gloo rehearsal of the sharded path on the CPU.

What runs on the GPUs at N > 1 is: row-block partition (liblcg_amd.partition), all-gather of the
x slices into a padded global vector addressed by GLOBAL column indices, all-reduce of every inner
product.  Here the same partition helper and the same collective placement drive the ORACLE's
kernels, and the result must agree with the single-process oracle CG -- the data path is right by
construction before the 8-GPU bench ever runs.  The RCCL unique-id courier is exercised with a
stub library.
"""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _StubLib:
    """Records what init_comm_from_torch hands to the C ABI."""
    def __init__(self, rank):
        self.rank, self.got = rank, None

    def lcg_hip_comm_unique_id(self, buf):
        for i in range(128):
            buf[i] = (i * 7 + 3) % 256
        return 0

    def lcg_hip_comm_init(self, world, rank, idb):
        self.got = (world, rank, bytes(idb))
        return 0

    def lcg_hip_last_error(self):
        return b""


def _worker(rank, world, port_no, n, band, q, schedule="classic"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from liblcg_amd import partition
    from oracle import pyoracle as po
    orc = po.Oracle("port")

    # 1. the unique-id courier
    stub = _StubLib(rank)
    assert partition.init_comm_from_torch(stub) == (rank, world)
    assert stub.got == (world, rank, bytes((i * 7 + 3) % 256 for i in range(128)))

    # 2. sharded CG: same algorithm/collectives as the device path (lcg.cpp:143-274 semantics)
    g = orc.gen_init(n, 16, band, True, 3, 0.01)
    r0, r1 = partition.shard_range(n, world, rank)
    rpr = partition.rows_per_rank(n, world)
    rp, ci, v = orc.gen_rows(g, r0, r1)                     # local rows, GLOBAL columns
    xt = orc.gen_xtrue(g, r0, r1)
    glen = partition.gathered_length(n, world)

    def gather(xl):
        pad = np.zeros(rpr); pad[:r1 - r0] = xl
        parts = [torch.zeros(rpr, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(pad))
        return torch.cat(parts).numpy()

    def ax(xl):
        xf = gather(xl)
        assert len(xf) == glen
        # a rectangular shard: local rows x padded global columns
        y = np.zeros(r1 - r0)
        for i in range(r1 - r0):
            s, e = rp[i], rp[i + 1]
            y[i] = np.dot(v[s:e], xf[ci[s:e]])
        return y

    def gdot(a, b):
        t = torch.tensor([float(np.dot(a, b))], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item())

    reductions = [0]

    def gsum(*pairs):           # ONE all-reduce for several inner products
        t = torch.tensor([float(np.dot(a, b)) for a, b in pairs], dtype=torch.float64)
        dist.all_reduce(t)
        reductions[0] += 1
        return t.tolist()

    b = ax(xt)
    eps, m = 1e-10, np.zeros(r1 - r0)
    Ad = ax(m); gk = Ad - b; d = -gk
    g2 = gdot(gk, gk); t = 0
    if schedule == "classic":
        while True:
            if np.sqrt(g2) / n <= eps:
                break
            t += 1
            Ad = ax(d); ak = g2 / gdot(d, Ad)
            m += ak * d; gk += ak * Ad
            g2n = gdot(gk, gk); bk = g2n / g2; g2 = g2n
            d = bk * d - gk
    else:
        # the one-reduction schedule of solvers_real.hip (OpCg1Update / OpCg1Dots / FinCg1Close)
        w = ax(gk)
        (gw,) = gsum((gk, w))
        ak, bk = g2 / gw, 0.0
        d = np.zeros_like(gk); Ad = np.zeros_like(gk)
        reductions[0] = 0
        while True:
            if np.sqrt(g2) / n <= eps:
                break
            t += 1
            d = bk * d - gk; Ad = bk * Ad - w
            m += ak * d; gk += ak * Ad
            w = ax(gk)
            g2n, gw = gsum((gk, gk), (gk, w))
            bk = g2n / g2
            ak = g2n / (gw - bk * g2n / ak)
            g2 = g2n
        assert reductions[0] == t
    err = gdot(m - xt, m - xt) ** 0.5
    mall = gather(m)[:n]
    # 3. op(A).x = A^T.x on the sharded matrix as comm.hip: dist_spmv_op places it: every rank multiplies the transpose of ITS
    #    rows with ITS slice of x over the padded global height, the sum over ranks is scattered back in rows-per-rank blocks
    #    (gloo has no reduce-scatter: all-reduce + own block is the same sum).  Non-symmetric values so that A^T != A.
    g2_ = orc.gen_init(n, 16, band, False, 5, 0.01)
    rp2, ci2, v2 = orc.gen_rows(g2_, r0, r1)
    xs = orc.gen_xtrue(g2_, r0, r1)
    z = np.zeros(glen)
    for i in range(r1 - r0):
        s, e = rp2[i], rp2[i + 1]
        np.add.at(z, ci2[s:e], v2[s:e] * xs[i])
    zt = torch.from_numpy(z); dist.all_reduce(zt)
    yT = zt.numpy()[rank * rpr: rank * rpr + (r1 - r0)]
    yT_all = gather(yT)[:n]
    if rank == 0:
        q.put((t, err, mall, yT_all))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,band,schedule", [(2, 40, "classic"), (2, 0, "classic"), (3, 40, "classic"),
                                                 (2, 40, "one-reduction"), (3, 0, "one-reduction")])
def test_sharded_cg_matches_single_process(world, band, schedule, port):
    from oracle import pyoracle as po
    n = 1501
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port_no = _free_port()          # all ranks share it
    procs = [ctx.Process(target=_worker, args=(r, world, port_no, n, band, q, schedule)) for r in range(world)]
    for p in procs:
        p.start()
    t, err, mall, yT_all = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g = port.gen_init(n, 16, band, True, 3, 0.01)
    rp, ci, v = port.gen_rows(g)
    xt = port.gen_xtrue(g)
    b = port.csr_matvec(rp, ci, v, xt)
    ref = port.solve(po.LCG_CG, rp, ci, v, b, para=po.default_para(epsilon=1e-10, abs_diff=1))
    assert ref["ret"] == 0
    assert abs(t - ref["iters"]) <= 1
    assert np.linalg.norm(mall - ref["x"]) / np.linalg.norm(ref["x"]) <= 1e-9
    assert err <= 1e-5 * np.linalg.norm(xt)
    # the sharded transposed product against scipy's on the whole matrix
    import scipy.sparse as sp
    g2 = port.gen_init(n, 16, band, False, 5, 0.01)
    rp2, ci2, v2 = port.gen_rows(g2)
    A2 = sp.csr_matrix((v2, ci2, rp2), shape=(n, n))
    want = A2.T @ port.gen_xtrue(g2)
    assert np.abs(yT_all - want).max() <= 1e-12 * np.abs(want).max()
    assert np.abs(want - A2 @ port.gen_xtrue(g2)).max() > 1e-3 * np.abs(want).max()      # A^T != A


def test_partition_rules():
    from liblcg_amd import partition as P
    for n, w in ((10, 1), (10, 3), (10_000_000, 8), (7, 8), (1501, 2)):
        rpr = P.rows_per_rank(n, w)
        assert rpr * w >= n and P.gathered_length(n, w) == rpr * w
        cover = []
        for r in range(w):
            r0, r1 = P.shard_range(n, w, r)
            assert 0 <= r0 <= r1 <= n and r1 - r0 <= rpr
            assert r0 == min(n, r * rpr)                      # global index == padded index
            cover += list(range(r0, r1)) if n < 100 else []
        if n < 100:
            assert cover == list(range(n))
    assert P.shard_range(10_000_000, 8, 7) == (8_750_000, 10_000_000)
