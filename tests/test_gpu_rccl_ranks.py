"""-m gpu: the NORTH-STAR exchange with more than one rank -- on one GPU.

SURVEY.md 8e: rows partitioned over the ranks, `ncclAllGather` of the x slices before each A.x, `ncclAllReduce` of the packed
dots at each sync point.  The real RCCL refuses two ranks on one device, so on a one-GPU box those calls (comm.hip: dist_spmv_impl,
halo_exchange, dist_spmv_op, comm_allreduce), the two-stream fork/join around them and the lock-step loops on their sums ran
either with a one-rank communicator (tests/test_gpu_comm.py) or not at all.  Here P processes share GPU 0 and the library binds its
eleven nccl* symbols from tests/fake_rccl (LCG_HIP_RCCL_LIB): stream-ordered copies through HIP-IPC staging buffers, the ranks
meeting in host shared memory, the call sequence checked across ranks.  Everything between the collective calls is the product.

Checked per rank, on its own rows: A.x (modes 0 and 1, twelve alternating inputs) against the oracle's product row-wise at
1e-13 |A||x|; the neighbour-range product bit-equal to the all-gather product; four capped iterations of CG (both schedules), PCG,
BiCGStab, CGS against the oracle's loop (lcg.cpp:206-264 &c. restated) AND the single-process run at 1e-10; the same solvers to
convergence in lock-step; A^T.x / A^H.x through the reduce-scatter; the complex solvers on the bundled case_10K_cA; RCCL for x with
the peer-mapped mailboxes for the sums.  P = 2 .. 5 (a box admits six processes on its GPU, the test process is one of them);
P = 2 and 4 once more at BASELINE configs[3]'s real shard sizes (5M and 2.5M rows of the 10M-row system), P = 5 at the 8-way shard
height (1.25M rows each)."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")
FAKE_SO = os.path.join(FAKE_DIR, "librccl_fake.so")


def fake_rccl():
    if not os.path.exists(FAKE_SO):
        subprocess.check_call(["make", "-s", "-C", FAKE_DIR])
    return FAKE_SO


SMALL = ({"tag": "band", "n": 60000, "band": 700, "sym": True, "pattern": 1, "seed": 3, "force": "packed"},
         {"tag": "scr", "n": 30011, "band": 0, "sym": True, "pattern": 0, "seed": 3},
         {"tag": "nsym", "n": 45000, "band": 1200, "sym": False, "pattern": 1, "seed": 3, "force": "packed"},
         {"tag": "rrb", "n": 52000, "band": 900, "sym": True, "pattern": 2, "seed": 3, "force": "tiled"})


def make_reference(d, cases, port, with_complex, mailbox_case=None, threads=8):
    """One .npy per array under d: inputs, the oracle's products and row-wise bounds, the oracle's and the single-process
    product's iterates after four iterations, single-process iteration counts."""
    from liblcg_amd import api
    from oracle import pyoracle as po
    os.makedirs(d, exist_ok=True)
    meta = {"cases": list(cases), "complex": bool(with_complex), "mailbox_case": mailbox_case}

    def put(key, arr):
        np.save(os.path.join(d, key.replace("/", "__") + ".npy"), np.ascontiguousarray(arr))

    for case in cases:
        tag, n, sym = case["tag"], case["n"], case["sym"]
        A = api.CsrMatrix.generate(n, 16, case["band"], sym, case["seed"], 0.01, pattern=case["pattern"])
        if sym:
            A.build_jacobi()
        rp, ci, v = A.arrays_to_host()
        x1 = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, x1)
        x2 = 2.0 * x1 - 0.5         # both signs: cancellation inside the rows
        x1h, x2h = x1.cpu().numpy(), x2.cpu().numpy()
        y1 = port.csr_matvec(rp, ci, v, x1h, threads=threads)
        put(f"{tag}/x1", x1h); put(f"{tag}/x2", x2h)
        put(f"{tag}/y1", y1); put(f"{tag}/y2", port.csr_matvec(rp, ci, v, x2h, threads=threads))
        put(f"{tag}/bound1", port.csr_matvec(rp, ci, np.abs(v), np.abs(x1h), threads=threads))
        put(f"{tag}/bound2", port.csr_matvec(rp, ci, np.abs(v), np.abs(x2h), threads=threads))
        if not sym:
            import scipy.sparse as sp
            M = sp.csr_matrix((v, ci, rp), shape=(n, n))
            put(f"{tag}/yT", M.T @ x1h); put(f"{tag}/boundT", abs(M).T @ np.abs(x1h))
        b = torch.from_numpy(y1).cuda()
        for name, sid, osid in (("cg", api.LCG_CG, po.LCG_CG), ("pcg", api.LCG_PCG, po.LCG_PCG), ("bicgstab", api.LCG_BICGSTAB, po.LCG_BICGSTAB),
                                ("cgs", api.LCG_CGS, po.LCG_CGS)):
            if not sym and name in ("cg", "pcg"):
                continue
            for ad in (1, 0):
                r = port.solve(osid, rp, ci, v, y1, para=po.default_para(epsilon=1e-300, abs_diff=ad, max_iterations=4), jacobi=name == "pcg", threads=threads)
                assert r["ret"] == -1019 and r["iters"] == 4
                meta[f"{tag}/{name}4_res_ad{ad}"] = float(r["residual"])
            put(f"{tag}/{name}4_oracle", r["x"])
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            p4 = api.lcg_default_parameters(epsilon=1e-300, abs_diff=1, max_iterations=4)
            if name == "pcg":
                api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, p4, A)
            else:
                api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, p4, A, sid)
            put(f"{tag}/{name}4_single", m.cpu().numpy())
            if not case.get("big"):
                m.zero_()
                pc = api.lcg_default_parameters(epsilon=1e-10, abs_diff=1)
                if name == "pcg":
                    info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, pc, A)
                else:
                    info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, pc, A, sid)
                meta[f"{tag}/{name}_its"] = int(info.iterations)
        A.destroy()
        del rp, ci, v
    if with_complex:
        from liblcg_amd.coo_io import coo_to_csr_host, read_coo_system, read_solution
        import scipy.sparse as sp
        G = os.path.join(ROOT, "tests", "golden")
        nc, row, col, val, bc = read_coo_system(os.path.join(G, "case_10K_cA"), True)
        rp, ci, v = coo_to_csr_host(nc, row, col, val)
        rng = np.random.default_rng(5)
        x1 = rng.standard_normal(nc) + 1j * rng.standard_normal(nc)
        M = sp.csr_matrix((v, ci, rp), shape=(nc, nc))
        put("cplx/x1", x1); put("cplx/y1", M @ x1); put("cplx/yH", M.conj().T @ x1); put("cplx/yT", M.T @ x1)
        put("cplx/xsol", read_solution(os.path.join(G, "case_10K_cB"), True))
        Ac = api.CsrMatrix.from_csr(rp, ci, v)
        for name, sid in (("bicg", api.CLCG_BICG), ("bicg_sym", api.CLCG_BICG_SYM), ("tfqmr", api.CLCG_TFQMR)):
            m = torch.zeros(nc, dtype=torch.complex128, device="cuda")
            info = api.clcg_solver("clcg_hip_csr_ax", None, m, torch.from_numpy(bc).cuda(), nc, api.clcg_default_parameters(epsilon=1e-10, abs_diff=1),
                                   Ac, sid, shadow_seed=7)
            meta[f"cplx/{name}_its"] = int(info.iterations)
            meta[f"cplx/{name}_err"] = float(np.abs(m.cpu().numpy() - read_solution(os.path.join(G, "case_10K_cB"), True)).max())
        Ac.destroy()
    json.dump(meta, open(os.path.join(d, "meta.json"), "w"))
    torch.cuda.empty_cache()
    return meta


def run_ranks(tmp_path, ref_dir, world, port_no, timeout=240, extra=None):
    """The ranks as child processes, their output in files (so that a rank that hangs leaves its story behind); a rank that is still
    running after `timeout` seconds ends the test with everybody's last lines."""
    procs, outs, errs = [], [], []
    logdir = os.environ.get("LCG_RCCL_LOGDIR") or str(tmp_path)
    os.makedirs(logdir, exist_ok=True)
    for r in range(world):
        out = str(tmp_path / f"rccl_{world}_{r}.json")
        outs.append(out)
        errs.append(os.path.join(logdir, f"world{world}_rank{r}_{port_no}.err"))
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port_no),
                   LCG_HIP_RCCL_LIB=fake_rccl(), FAKE_RCCL_TIMEOUT_S="60", FAKE_RCCL_STATS="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   LCG_RCCL_VERBOSE="1", LCG_RCCL_WATCHDOG_S=str(max(30, timeout - 30)), **(extra or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py"), ref_dir, out],
                                      stdout=subprocess.DEVNULL, stderr=open(errs[-1], "w"), env=env))
    t_end = time.time() + timeout
    hung = False
    for p in procs:
        try:
            p.wait(timeout=max(1.0, t_end - time.time()))
        except subprocess.TimeoutExpired:
            hung = True
    if hung:
        for q in procs:
            if q.poll() is None:
                q.kill()
    logs = [open(e).read()[-4000:] for e in errs]
    assert not hung, "a rank was still running after %d s:\n" % timeout + "\n-----\n".join(logs)
    assert all(p.returncode == 0 for p in procs), "\n-----\n".join(logs)
    res = [json.load(open(o)) for o in outs]
    # what the stand-in library itself counted (its closing line on stderr): the collectives really went through it
    import re
    for r, log in zip(res, logs):
        m = re.search(r"calls: all-gather (\d+), all-reduce (\d+), reduce-scatter (\d+), send (\d+), recv (\d+), groups (\d+)", log)
        assert m, log
        r["fake_calls"] = dict(zip(("all_gather", "all_reduce", "reduce_scatter", "send", "recv", "groups"), map(int, m.groups())))
    return res


def check_real(res, meta, world):
    for case in meta["cases"]:
        tag, sym, big = case["tag"], case["sym"], case.get("big")
        for r in res:
            for mode in (0, 1):
                assert r[f"{tag}/m{mode}/spmv_worst"] <= 1e-13, (tag, mode, r["rank"], r[f"{tag}/m{mode}/spmv_worst"])
                for name in ("cg", "cg_classic", "pcg", "bicgstab", "cgs"):
                    for ad in (1, 0):
                        key = f"{tag}/m{mode}/{name}4/ad{ad}"
                        if key not in r:
                            continue
                        ret, its, vs_oracle, vs_single, resid = r[key]
                        assert (ret, its) == (-1019, 4), (key, r[key])
                        assert vs_oracle <= 1e-10 and vs_single <= 1e-10, (key, r[key])
                        want = meta[f"{tag}/{name.split('_')[0]}4_res_ad{ad}"]      # the monitored residual is the oracle's (all-reduced sums)
                        assert abs(resid - want) <= 1e-8 * want, (key, resid, want)
            assert r[f"{tag}/m1_equals_m0"], (tag, r["rank"])
            if not sym:
                assert r[f"{tag}/opT_worst"] <= 1e-13, (tag, r[f"{tag}/opT_worst"])
            if case["pattern"] == 1 and case["band"]:       # banded: the neighbour ranges are band-wide, the all-gather moves everything
                assert 0 < r[f"{tag}/m1/recv"] <= 2 * case["band"] < r[f"{tag}/m0/recv"], (tag, r[f"{tag}/m1/recv"], r[f"{tag}/m0/recv"])
            if not big:
                for name in ("cg", "cg_classic", "pcg", "bicgstab", "cgs"):
                    if f"{tag}/{name}" not in r:
                        continue
                    ret, its, err = r[f"{tag}/{name}"]
                    assert ret == 0 and err < 1e-5, (tag, name, r[f"{tag}/{name}"])
                    if name in ("cg", "cg_classic", "pcg", "cgs"):       # insensitive recurrences: the single-process count
                        assert abs(its - meta[f"{tag}/{name.split('_')[0]}_its"]) <= 3, (tag, name, its)
        # lock-step: every rank reports the same return codes and counts
        for key in res[0]:
            if key.startswith(tag + "/") and isinstance(res[0][key], list):
                assert len({tuple(r[key][:2]) for r in res}) == 1, key


@pytest.fixture(scope="module")
def small_ref(tmp_path_factory, port):
    d = str(tmp_path_factory.mktemp("rccl_ref_small"))
    return d, make_reference(d, SMALL, port, with_complex=True, mailbox_case="band")


@pytest.mark.parametrize("world", [2, 3, 4, 5])
def test_north_star_exchange_with_real_ranks(tmp_path, small_ref, world):
    ref_dir, meta = small_ref
    res = run_ranks(tmp_path, ref_dir, world, 29600 + world)
    assert all("librccl_fake" in r["library"] for r in res)
    for r in res:       # every kind of collective the library makes went through the stand-in, on every rank
        c = r["fake_calls"]
        assert c["all_gather"] > 100 and c["all_reduce"] > 100 and c["reduce_scatter"] >= 5 and c["send"] > 50 and c["recv"] > 50 and c["groups"] > 50, c
    check_real(res, meta, world)
    for r in res:
        assert "run blocks" in r["band/m0/kernel"] or "packed" in r["band/m0/kernel"], r["band/m0/kernel"]
        assert "k_tile_spmv" in r["rrb/m0/kernel"], r["rrb/m0/kernel"]
        for mode in (0, 1):
            assert r[f"cplx/m{mode}/spmv_err"] < 1e-13 and r[f"cplx/m{mode}/yH_err"] < 1e-13 and r[f"cplx/m{mode}/yT_err"] < 1e-13, r
            for name in ("bicg", "bicg_sym", "tfqmr"):
                if f"cplx/m{mode}/{name}" not in r:
                    continue
                ret, its, err = r[f"cplx/m{mode}/{name}"]
                assert ret == 0, (name, r[f"cplx/m{mode}/{name}"])
                assert err < max(5e-3, 10 * meta[f"cplx/{name}_err"]), (name, err)
                assert abs(its - meta[f"cplx/{name}_its"]) <= 0.12 * meta[f"cplx/{name}_its"] + 3, (name, its)
        for mode in (0, 1):
            ret, its, vs_oracle = r[f"band/mailbox/m{mode}/cg4"]
            assert (ret, its) == (-1019, 4) and vs_oracle <= 1e-10, r[f"band/mailbox/m{mode}/cg4"]
        assert r["p2p_status"] == 2


BIG = ({"tag": "diag10m", "n": 10_000_000, "band": 131072, "sym": True, "pattern": 1, "seed": 1, "big": True},
       {"tag": "rrb10m", "n": 10_000_000, "band": 131072, "sym": True, "pattern": 2, "seed": 1, "big": True})


@pytest.fixture(scope="module")
def big_ref(tmp_path_factory, port):
    d = str(tmp_path_factory.mktemp("rccl_ref_10m"))
    return d, make_reference(d, BIG, port, with_complex=False)


@pytest.mark.parametrize("world", [2, 4])
def test_config3_shard_sizes_over_the_collectives(tmp_path, big_ref, world):
    """BASELINE configs[3] at its real size: the 10M-row system (constant diagonals = the headline; row-random band) split over 2
    and 4 processes, all-gather and neighbour ranges, against the oracle at 10M rows."""
    ref_dir, meta = big_ref
    res = run_ranks(tmp_path, ref_dir, world, 29610 + world)
    check_real(res, meta, world)
    for r in res:
        assert "run blocks" in r["diag10m/m0/kernel"], r["diag10m/m0/kernel"]
        assert "k_tile_spmv" in r["rrb10m/m0/kernel"], r["rrb10m/m0/kernel"]


def test_eight_way_shard_height_five_ranks(tmp_path, tmp_path_factory, port):
    """Five ranks of 1.25M rows each (the shard height of the 8-way split of the 10M-row system; five is what one GPU admits)."""
    d = str(tmp_path_factory.mktemp("rccl_ref_8way"))
    cases = ({"tag": "diag8w", "n": 6_250_000, "band": 131072, "sym": True, "pattern": 1, "seed": 1, "big": True},)
    meta = make_reference(d, cases, port, with_complex=False)
    res = run_ranks(tmp_path, d, 5, 29620)
    check_real(res, meta, 5)
    assert all("run blocks" in r["diag8w/m0/kernel"] for r in res)
