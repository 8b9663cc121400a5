"""-m gpu: BASELINE.json configs[3] -- the 10M-row system row-partitioned over 8 ranks -- at its REAL shard size on one GPU.

What one GPU can show of it (the workload is the A.x callback of sample8.cu:96-103 split by rows):
 * every one of the 8 shards (1.25M rows x ~33, local- and remote-column parts, the packed / tiled copy the shard's
   automatic choice builds, the 262K-row remote part) multiplies to the rows the UNSHARDED 10M-row product has;
 * five REAL ranks (processes) on the one GPU, each owning 1.25M rows of a 6.25M-row system of the same family -- the
   per-rank geometry of an interior rank of the 8-way split: same shard height, same band, a neighbour on both sides --
   run the direct exchange, the rank-ordered sums and the lock-step loops against the single-process run.
Not shown here: the links (one GPU per box)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("pattern,P", [("constant_diagonals", 8), ("row_random_band", 8), ("constant_diagonals", 2), ("constant_diagonals", 4),
                                       ("row_random_band", 4)])
def test_config3_shards_full_size(pattern, P):
    """(P = 2 and 4: the shard sizes of the other points of the scaling curve BASELINE.json's metric is quoted on)"""
    from liblcg_amd import _lib, api, partition
    lib = _lib.load()
    n, band = 10_000_000, 131072
    pat = api.GEN_DIAGONALS if pattern == "constant_diagonals" else api.GEN_ROW_RANDOM_BAND
    A = api.CsrMatrix.generate(n, 16, band, True, 1, 0.01, pattern=pat)
    x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, x)
    x = 2.0 * x - 0.5                                        # both signs: cancellation inside the rows
    y = torch.empty_like(x)
    A.spmv(x, y); api.synchronize()
    full_kernel = lib.lcg_hip_csr_last_kernel(A.h).decode()
    diag = torch.empty_like(x)
    A.build_jacobi(diag)
    api.synchronize()
    A.destroy()
    # |A||x| <= 2 diag (strict diagonal dominance, |x| <= 1.5): the row-wise rounding bound of the other parity tests
    bound = 1e-13 * 3.0 * diag.abs()
    glen = partition.gathered_length(n, P)
    xpad = torch.zeros(glen, dtype=torch.float64, device="cuda"); xpad[:n] = x
    want_kernel = "run blocks" if pattern == "constant_diagonals" else "k_tile_spmv"
    assert want_kernel in full_kernel, full_kernel
    for r in range(P):
        r0, r1 = partition.shard_range(n, P, r)
        S = api.CsrMatrix.generate(n, 16, band, True, 1, 0.01, r0, r1, pattern=pat)
        assert lib.lcg_hip_csr_split_for_test(S.h, n, P, r) == 0
        xf = lib.lcg_hip_csr_xfull(S.h)
        assert lib.lcg_hip_memcpy(xf, xpad.data_ptr(), xpad.numel() * 8, 3) == 0        # device to device
        ys = torch.full((r1 - r0,), 7.0, dtype=torch.float64, device="cuda")
        S.spmv(x[r0:r1].clone(), ys); api.synchronize()
        kern = lib.lcg_hip_csr_last_kernel(S.h).decode()
        assert want_kernel in kern, (r, kern)          # the shard's local part takes the kernel family the whole matrix takes
        d = (ys - y[r0:r1]).abs()
        assert bool((d <= bound[r0:r1]).all()), (r, float(d.max()))
        # rows whose columns all lie inside the shard never see the remote part: the local product alone -- bit for bit
        # the unsharded kernel's arithmetic for the run-block kernel (same 64-row blocks: the shard starts at a multiple of 64)
        lo = r0 + (band if r > 0 else 0) + 64
        hi = r1 - (band if r < P - 1 else 0) - 64
        if pattern == "constant_diagonals" and r0 % 64 == 0:
            assert torch.equal(ys[lo - r0:hi - r0], y[lo:hi]), r
        nloc = int(lib.lcg_hip_csr_local_nnz(S.h))
        assert 0.9 * S.nnz < nloc <= S.nnz and (nloc < S.nnz) == (P > 1)
        S.destroy()


def test_config3_five_real_ranks_at_the_shard_size(tmp_path):
    """Five processes on GPU 0, 1.25M rows each (the 8-way shard height of the 10M-row system), band 131072, constant
    diagonals and the row-random band: slices of A.x on 12 alternating inputs to rounding, CG / PCG / CGS / BiCGStab to
    convergence in lock-step with the single-process iteration counts."""
    from test_gpu_direct import _reference
    world, n, band = 5, 6_250_000, 131072
    cases = (("band", n, band, True), ("rrb", n, band, True))
    ref_path = str(tmp_path / "ref.npz")
    ref = _reference(ref_path, cases, with_complex=False)
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"c3_{r}.json")
        outs.append(out)
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", LCG_HIP_P2P_TIMEOUT_MS="20000")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_direct_worker.py"), ref_path, out],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    logs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(so[-1500:] + se[-3000:])
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    res = [json.load(open(o)) for o in outs]
    for r in res:
        assert 0 < r["band/recv"] <= 2 * band and 0 < r["rrb/recv"] <= 2 * band, r
        assert "k_spmv_ldsp" in r["band/kernel"] and "run blocks" in r["band/kernel"], r["band/kernel"]
        assert "k_tile_spmv" in r["rrb/kernel"] and "pushing blocks" in r["rrb/kernel"], r["rrb/kernel"]
        for tag in ("band", "rrb"):
            assert r[f"{tag}/spmv_err"] < 1e-13, (tag, r)
            for name in ("cg", "pcg", "bicgstab", "cgs"):
                ret, its, err = r[f"{tag}/{name}"]
                assert ret == 0 and err < 1e-4, (tag, name, r[f"{tag}/{name}"])        # stop rule: sqrt(g.g)/N <= 1e-10
                if name in ("cg", "cgs"):
                    assert abs(its - int(ref[f"{tag}/{name}_its"])) <= 3, (tag, name, its, int(ref[f"{tag}/{name}_its"]))
    for key in res[0]:
        if "/" in key and isinstance(res[0][key], list):
            assert len({tuple(r[key][:2]) for r in res}) == 1, key
