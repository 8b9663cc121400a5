"""-m gpu: what can be verified of the sharded path on ONE GPU.

 * the neighbour-exchange plan: the per-owner column ranges computed on the device equal the
   ranges computed from the oracle's rows;
 * the RCCL plumbing end to end with a one-rank communicator (LCG_HIP_FORCE_COMM): unique id
   courier over torch.distributed, ncclCommInitRank, all-gather / grouped send-recv on the second
   stream, all-reduce inside the scalar steps -- bench.py must produce the same iterates as the
   unsharded run.
The multi-rank data path itself is rehearsed on the CPU in tests/test_dist_cpu.py.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_need_ranges_match_oracle_rows(port):
    from liblcg_amd import _lib, api, partition
    lib = _lib.load()
    n, nranks = 20011, 4
    for band in (300, 0):
        g = port.gen_init(n, 16, band, True, 2, 0.01)
        rpr = partition.rows_per_rank(n, nranks)
        for r in range(nranks):
            r0, r1 = partition.shard_range(n, nranks, r)
            rp, ci, v = port.gen_rows(g, r0, r1)
            A = api.CsrMatrix.generate(n, 16, band, True, 2, 0.01, r0, r1)
            assert lib.lcg_hip_csr_split_for_test(A.h, n, nranks, r) == 0
            out = (C.c_int64 * (2 * nranks))()
            assert lib.lcg_hip_csr_need_ranges_for_test(A.h, nranks, out) == 0
            for q in range(nranks):
                cols = ci[(ci // rpr) == q]
                lo, hi = out[2 * q], out[2 * q + 1]
                if q == r or len(cols) == 0:
                    assert hi <= lo or q == r
                else:
                    assert (lo, hi) == (cols.min(), cols.max() + 1)
            if band:    # a banded shard only talks to its neighbours, and only about 2*band entries
                far = [q for q in range(nranks) if abs(q - r) > 1]
                assert all(out[2 * q + 1] <= out[2 * q] for q in far)
                vol = sum(max(0, out[2 * q + 1] - out[2 * q]) for q in range(nranks) if q != r)
                assert vol <= 2 * band


def _run_bench(extra_env, *args):
    env = dict(os.environ, **extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-live-pmc", "--rows", "400000",
                        "--band", "5000", "--steps", "30", "--warmup", "3", *args],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("mode", ["0", "1"])
def test_one_rank_communicator_reproduces_unsharded_run(mode):
    plain = _run_bench({})
    forced = _run_bench({"LCG_HIP_FORCE_COMM": "1", "LCG_HIP_DIST_MODE": mode, "MASTER_PORT": "29541"})
    assert forced["config"]["partition"].startswith("row-block x1")
    assert ("neighbour" in forced["config"]["partition"]) == (mode == "1")
    assert forced["config"]["nnz"] == plain["config"]["nnz"]
    assert forced["steps"] == plain["steps"] == 30
    # same matrix, same arithmetic: the error after 30 iterations agrees to rounding
    fc, pc = forced["solution_check"], plain["solution_check"]
    assert abs(fc["rel_err_vs_x_true_after_100_iterations"] - pc["rel_err_vs_x_true_after_100_iterations"]) <= 1e-4 * pc["rel_err_vs_x_true_after_100_iterations"]
    # the monitored residual after 25 iterations (the classic schedule unsharded, one reduction sharded: same iterates)
    assert abs(fc["residual_monitored_after_25"] - pc["residual_monitored_after_25"]) <= 1e-6 * pc["residual_monitored_after_25"]
    assert forced["value_rccl_allgather"] > 0 and "all-gather + rccl all-reduce" in forced["comm_probe"]["configurations_it_per_s"]
    assert forced["value"] > 0 and forced["timed_repetitions"] == 5 and forced["value_min"] <= forced["value"] <= forced["value_max"]
    # the unsharded line names its workload for what it is and carries the three column patterns
    assert "constant diagonals" in plain["config"]["workload"] and plain["roofline"]["traffic_source"]
    assert set(plain["variants"]) == {"constant_diagonals", "row_random_band", "scrambled", "mixed_rows", "stencil27"}
    assert all(v["it_per_s"] > 0 and 0 < v["frac"] < 1 for v in plain["variants"].values())


def test_sharded_products_use_the_tiled_kernel_too():
    """A row-random band shard under the all-gather exchange: the local-column part (all of it with one rank) is large
    enough for the automatic choice to take the tiled product; the bench's guard (residual recomputed with a second A.x,
    100 iterations towards x_true) passes on it."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-live-pmc", "--no-variants", "--rows", "1500000", "--band", "40000",
                        "--pattern", "row_random_band", "--steps", "20", "--warmup", "3", "--reps", "2"], capture_output=True, text=True,
                       env=dict(os.environ, LCG_HIP_FORCE_COMM="1", LCG_HIP_DIST_MODE="0", MASTER_PORT="29543"), timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert "k_tile_spmv" in out["roofline"]["kernel"], out["roofline"]["kernel"]
    assert out["solution_check"]["rel_err_vs_x_true_after_100_iterations"] < 1e-3 and "row-random band" in out["config"]["workload"]


def _solve_cases(tmp_path, tag, extra_env):
    out = str(tmp_path / f"{tag}.npz")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_sharded_case.py"), out],
                       capture_output=True, text=True, env=dict(os.environ, **extra_env), timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    return np.load(out)


@pytest.mark.parametrize("mode", ["0", "1"])
def test_sharded_loops_converge_like_the_unsharded_ones(tmp_path, mode):
    """Every solver family to CONVERGENCE over a one-rank communicator (reduce -> all-reduce -> finish
    scalar steps, split A.x, lock-step batched enqueue) against the plain single-GPU run of the same
    process image.  Bands = those of tests/test_gpu_solvers.py: CG / PCG / CGS insensitive (1e-9, +-3
    iterations; sharded CG runs the one-reduction schedule), BiCGStab(2) 1e-7 and +-15 %, complex
    solvers 5e-5 and +-12 %.  Capped and progress-callback runs must count exactly."""
    plain = _solve_cases(tmp_path, "plain", {})
    forced = _solve_cases(tmp_path, "forced", {"LCG_HIP_FORCE_COMM": "1", "LCG_HIP_DIST_MODE": mode, "MASTER_PORT": "29548"})
    bands = {"cg": (1e-9, 3, 0.0), "pcg": (1e-9, 3, 0.0), "cgs": (1e-9, 3, 0.0), "bicgstab": (1e-7, 3, 0.15),
             "bicgstab2": (1e-7, 3, 0.15), "c_bicg": (5e-5, 3, 0.12), "c_bicg_sym": (5e-5, 3, 0.12), "c_cgs": (5e-5, 3, 0.12), "c_tfqmr": (5e-5, 3, 0.12)}
    for name, (xtol, it_abs, it_rel) in bands.items():
        rp, ip, _ = plain[f"{name}/meta"]; rf, itf, resf = forced[f"{name}/meta"]
        assert rp == rf == 0, name
        assert abs(itf - ip) <= max(it_abs, it_rel * ip), (name, ip, itf)
        xp, xf = plain[f"{name}/x"], forced[f"{name}/x"]
        assert np.linalg.norm(xf - xp) / np.linalg.norm(xp) <= xtol, name
    # op(A).x on the sharded matrix (A^H, A^T; complex and real): the one-rank reduce-scatter hands back what the unsharded
    # transpose computes, bit for bit (same materialised transpose, same kernels); conj(A) alone is refused when sharded
    for tag in ("opH", "opT", "opT_real", "op_real"):
        assert np.array_equal(forced[f"{tag}/y"], plain[f"{tag}/y"]), tag
    assert np.abs(plain["opT_real/y"] - plain["op_real/y"]).max() > 1e-3          # A != A^T: the transpose really was taken
    assert int(plain["op_conj_only_rc"]) == 0 and int(forced["op_conj_only_rc"]) == -2000
    assert tuple(forced["cg13/meta"][:2]) == tuple(plain["cg13/meta"][:2]) == (-1019, 13)
    assert np.linalg.norm(forced["cg13/x"] - plain["cg13/x"]) / np.linalg.norm(plain["cg13/x"]) <= 1e-10
    assert forced["cgpfp/meta"][0] == 0 and abs(forced["cgpfp/meta"][1] - plain["cgpfp/meta"][1]) <= 2
    assert forced["cgpfp/meta"][3] == forced["cgpfp/meta"][1] + 1          # Pfp called for k = 0..t


def test_a_failing_product_ends_the_solve_with_its_code():
    """liblcg's callback types return void (lcg.h:37-38).  A built-in callback that cannot make its product (here: a
    matrix whose direct exchange could not be set up -- no mailboxes in this process) parks the failure and the solver
    loop returns it (LCG_HIP_E_COMM = -2002) instead of iterating over a stale product and reporting an ordinary
    liblcg code; the handle refuses products until it is distributed again under a mode that works."""
    from liblcg_amd import _lib, api
    lib = _lib.load()
    n = 6000
    A = api.CsrMatrix.generate(n, 16, 50, True, 4, 0.01)
    A.build_jacobi()
    xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 4, 0, n, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    assert lib.lcg_hip_csr_distribute(A.h, n, 2) == -2002                  # needs the mailboxes: none connected
    assert lib.lcg_hip_spmv(A.h, xt.data_ptr(), b.data_ptr()) == -2002     # the handle refuses, it does not guess
    ax = _lib.fnptr(lib, "lcg_hip_csr_ax"); mx = _lib.fnptr(lib, "lcg_hip_jacobi_mx")
    p = api.lcg_default_parameters(epsilon=1e-10, abs_diff=1)
    m = torch.zeros_like(xt)
    for sid in (api.LCG_CG, api.LCG_CGS, api.LCG_BICGSTAB, api.LCG_BICGSTAB2):
        assert lib.lcg_hip_solver(ax, None, m.data_ptr(), b.data_ptr(), n, C.byref(p), A.h, sid, 1) == -2002, sid
    assert lib.lcg_hip_solver_preconditioned(ax, mx, None, m.data_ptr(), b.data_ptr(), n, C.byref(p), A.h, 1, 1) == -2002
    lo = torch.full_like(xt, -1.0); hi = torch.full_like(xt, 2.0)
    assert lib.lcg_hip_solver_constrained(ax, None, m.data_ptr(), b.data_ptr(), lo.data_ptr(), hi.data_ptr(), n, C.byref(p), A.h, 5, 1) == -2002
    assert b"distribute" in lib.lcg_hip_last_error()
    # distributed again under a mode that works here (one rank, all-gather = a copy): products and solves are back
    assert lib.lcg_hip_csr_distribute(A.h, n, 0) == 0
    A.spmv(xt, b); api.synchronize()
    m.zero_()
    assert lib.lcg_hip_solver(ax, None, m.data_ptr(), b.data_ptr(), n, C.byref(p), A.h, api.LCG_CG, 1) == 0
    assert ((m - xt).norm() / xt.norm()).item() < 1e-8


def test_bench_collects_its_traffic_counters_live():
    """bench.py at N = 1 collects roofline.traffic on the box it times on: two rocprofv3 --pmc child runs of the product alone.
    The figure must sit between what the kernel must move by construction and twice that (x re-fetched per XCD, metadata)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-variants", "--rows", "2000000",
                        "--steps", "10", "--warmup", "2", "--reps", "2"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    rf = json.loads(p.stdout.strip().splitlines()[-1])["roofline"]
    assert rf["traffic_source"].startswith("live on this box"), rf["traffic_source"]
    assert rf["must_move_bytes"] <= rf["traffic"] <= 2.0 * rf["must_move_bytes"], (rf["traffic"], rf["must_move_bytes"])
    assert 0.0 < rf["frac_traffic"] < 1.0
