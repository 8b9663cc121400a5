"""One GPU, no communicator: what does A.x of ONE rank's shard of the headline system cost once it
is split into the locally-owned and the remote columns (the form the sharded solver multiplies)?

  python scripts/shard_ax_lab.py [--ranks 8] [--rank 3] [--rows 10000000] [--band 131072]

Prints the time of (a) the unsplit shard against the full x and (b) local + remote parts (the x
exchange replaced by a one-time fill of the gather buffer), and checks (b) against (a).  Run it
under `rocprofv3 --kernel-trace --stats` for the per-kernel split.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--band", type=int, default=131072)
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    import torch
    from liblcg_amd import _lib, api, partition
    lib = _lib.load()
    assert lib.lcg_hip_init(0) == 0
    n = a.rows
    r0, r1 = partition.shard_range(n, a.ranks, a.rank)
    nloc = r1 - r0
    A = api.CsrMatrix.generate(n, 16, a.band, True, 1, 0.01, r0, r1)
    x = torch.rand(n, dtype=torch.float64, device="cuda")
    y1 = torch.empty(nloc, dtype=torch.float64, device="cuda")
    y2 = torch.empty_like(y1)

    def timed(fn):
        for _ in range(10):
            fn()
        api.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        api.synchronize()
        return (time.perf_counter() - t0) / a.reps * 1e6

    t_whole = timed(lambda: A.spmv(x, y1))
    nnz = A.nnz
    rc = lib.lcg_hip_csr_split_for_test(A.h, n, a.ranks, a.rank)
    assert rc == 0, rc
    xf = lib.lcg_hip_csr_xfull(A.h)
    assert lib.lcg_hip_memcpy(xf, x.data_ptr(), 8 * n, 3) == 0          # device -> device
    xl = x[r0:r1].contiguous()
    t_split = timed(lambda: A.spmv(xl, y2))
    err = (y1 - y2).abs().max().item() / y1.abs().max().item()
    loc = lib.lcg_hip_csr_local_nnz(A.h)
    bytes_ = 12 * nnz + 4 * (nloc + 1) + 16 * nloc
    print(f"rank {a.rank}/{a.ranks}: rows {nloc}, nnz {nnz} (local columns {loc}, remote {nnz - loc})")
    print(f"unsplit shard, full x      : {t_whole:8.1f} us  ({bytes_ / t_whole / 1e3:7.0f} GB/s algorithmic)")
    print(f"local + copy + remote parts: {t_split:8.1f} us  max rel diff {err:.2e}")
    assert err < 1e-13
    # the direct exchange with this rank standing in for its neighbours: pushing blocks in the local
    # product's grid, flags, waiting remote-column product -- one stream, two kernels
    rc = lib.lcg_hip_csr_direct_selfloop_for_test(A.h, a.ranks, a.rank)
    assert rc == 0, lib.lcg_hip_last_error()
    y3 = torch.empty_like(y1)
    t_direct = timed(lambda: A.spmv(xl, y3))
    err3 = (y1 - y3).abs().max().item() / y1.abs().max().item()
    print(f"direct (self-loop)         : {t_direct:8.1f} us  max rel diff {err3:.2e}  "
          f"({lib.lcg_hip_csr_exchange_volume(A.h)} doubles pushed per call)")
    assert err3 < 1e-13


if __name__ == "__main__":
    main()
