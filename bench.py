#!/usr/bin/env python3
"""Headline benchmark: CG iterations/s on the synthetic 10M-row, ~33 nnz/row fp64 SPD CSR.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" is one CG iteration (lcg.cpp:206-264: A.d, three inner products, three vector
updates, stop test) executed by lcg_hip_lcg() through the C ABI with inputs resident in HBM.
N > 1: the same 10M-row system is row-partitioned over the ranks (strong scaling), x is
all-gathered and the inner products all-reduced over RCCL inside the library.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for every field.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def spmv_bytes(n, nnz):     # 12*nnz + 4*(N+1) + 8*N (x) + 8*N (y)
    return 12 * nnz + 4 * (n + 1) + 16 * n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--band", type=int, default=131072, help="0 = scrambled (random-column) variant")
    ap.add_argument("--npairs", type=int, default=16)
    ap.add_argument("--solver", default="cg", choices=["cg", "pcg", "cgs", "bicgstab"])
    ap.add_argument("--cg-schedule", default="auto", choices=["auto", "classic", "one-reduction"],
                    help="lcg_hip_set_cg_schedule: auto = classic on one GPU, one all-reduce per iteration when sharded")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import numpy as np
    import torch

    from liblcg_amd import _lib, api, partition

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if world > 1:
        # one rank per GPU here (LOCAL_RANK picks the device): the remote-column product may wait for its
        # neighbours with every block and gather straight from the landing zone (one launch less per A.x).
        # Left off by default in the library because ranks SHARING a GPU starve each other that way.
        os.environ.setdefault("LCG_HIP_DIRECT_LAND", "1")
    torch.cuda.set_device(local_rank)
    lib = _lib.load()
    rc = lib.lcg_hip_init(local_rank)
    if rc:
        raise SystemExit(f"lcg_hip_init failed: {lib.lcg_hip_last_error().decode()}")

    dist = None
    exchange_probe = None
    p2p, p2p_why = False, "disabled"
    sharded = world > 1 or bool(os.environ.get("LCG_HIP_FORCE_COMM"))    # the env var rehearses the RCCL path on one GPU
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        partition.init_comm_from_torch(lib)
        # the <= 8 sums of a sync point go straight into the peers' mailboxes over xGMI when every
        # rank could map them and the self-test passed everywhere; otherwise RCCL all-reduces them
        if os.environ.get("LCG_HIP_P2P", "1") != "0":
            p2p, p2p_why = partition.init_p2p_from_torch(lib)
            if not p2p and rank == 0:
                print(f"[bench] direct all-reduce not used: {p2p_why}", file=sys.stderr)

    n = args.rows
    r0, r1 = partition.shard_range(n, world, rank)
    nloc = r1 - r0
    symmetric = args.solver in ("cg", "pcg")
    A = api.CsrMatrix.generate(n, args.npairs, args.band, symmetric, 1, 0.01, r0, r1)
    nnz_local = A.nnz
    if args.solver == "pcg":
        A.build_jacobi()
    xt = torch.empty(nloc, dtype=torch.float64, device="cuda")
    api.gen_xtrue(n, 1, r0, r1, xt)
    b = torch.empty_like(xt)
    exchange = "none"
    if sharded:
        # all-gather of x is the reference exchange; the neighbour (range) exchange moves only the
        # column ranges the shard touches and is used when it reproduces the all-gather product
        A.distribute(n, 0)
        A.spmv(xt, b); api.synchronize()
        exchange = "all-gather"
        # Cheaper exchanges are used only when they reproduce the all-gather product on this node:
        #   2 direct  -- owners write their boundary entries of x into the neighbours' buffers over the
        #                peer mappings from inside the A.x kernel (no collective, no second stream)
        #   1 ranges  -- grouped ncclSend/ncclRecv of the column ranges the shard touches
        # Every step is agreed by all ranks (all-reduce of a failure flag): all switch, or none does.
        want = int(os.environ.get("LCG_HIP_DIST_MODE", "2"))
        x2 = 2.0 * xt + 1.0
        b2 = torch.empty_like(b)
        A.spmv(x2, b2); api.synchronize()
        bad = torch.zeros(1, dtype=torch.float64, device="cuda")

        def anybody(failed):
            bad[0] = 1.0 if failed else 0.0
            dist.all_reduce(bad)
            return bad.item() != 0.0

        def close(u, v):    # the direct path adds a row's remote part in another order: rounding only
            return bool(((u - v).abs().max() <= 1e-12 * v.abs().max()).item())

        def ax_time(reps=30):   # A.x incl. its exchange, slowest rank (dependent back-to-back calls)
            t1 = torch.empty_like(b)
            for _ in range(5):
                A.spmv(xt, t1)
            api.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                A.spmv(xt, t1)
            api.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / reps * 1e6], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        labels = {0: "all-gather", 1: "neighbour ranges", 2: "direct peer writes"}
        timings = {0: ax_time()}
        for mode in (2, 1):
            if mode > want or (mode == 2 and not p2p):
                continue
            failed = False
            try:                            # phase 1: every rank builds its plan (collective inside)
                A.distribute(n, mode)
            except Exception as exc:
                print(f"[rank {rank}] exchange mode {mode} unavailable: {exc}", file=sys.stderr)
                failed = True
            if not anybody(failed):         # phase 2: all ranks exchange, or none does
                t1 = torch.empty_like(b); t2 = torch.empty_like(b); t3 = torch.empty_like(b)
                try:
                    A.spmv(xt, t1); A.spmv(x2, t2); A.spmv(xt, t3); api.synchronize()    # alternating inputs expose stale buffers
                    if mode == 1:
                        failed = not (torch.equal(t1, b) and torch.equal(t2, b2) and torch.equal(t3, b))
                    else:
                        failed = not (close(t1, b) and close(t2, b2) and close(t3, b))
                except Exception as exc:        # a timed-out exchange surfaces in synchronize()
                    print(f"[rank {rank}] exchange mode {mode} failed its check: {exc}", file=sys.stderr)
                    failed = True
                if not anybody(failed):
                    try:
                        timings[mode] = ax_time()
                    except Exception as exc:
                        print(f"[rank {rank}] exchange mode {mode} failed while timed: {exc}", file=sys.stderr)
                        raise
            A.distribute(n, 0)
            if p2p and anybody(lib.lcg_hip_p2p_status() < 0):
                # an exchange over the peer mappings timed out somewhere: tear the direct paths down on every
                # rank (no matrix uses them at this point) and let RCCL do everything
                p2p = False
                lib.lcg_hip_p2p_disconnect()
                timings.pop(2, None)
                if rank == 0:
                    print("[bench] direct paths switched off: an exchange timed out", file=sys.stderr)
        # the fastest validated exchange on THIS node (same decision everywhere: the timings are all-reduced)
        best = min(timings, key=lambda k: timings[k]) if want > 0 else 0
        if "LCG_HIP_DIST_MODE" in os.environ and want in timings:
            best = want                     # an explicit request wins when it validated
        if best != 0:
            A.distribute(n, best)
        exchange = labels[best]
        exchange_probe = {labels[k]: round(v, 1) for k, v in timings.items()}
        del x2, b2
    else:
        A.spmv(xt, b)
    api.synchronize()
    nnz = nnz_local
    if sharded:
        t = torch.tensor([nnz_local], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        nnz = int(t.item())

    m = torch.zeros_like(xt)
    ws = [torch.empty_like(xt) for _ in range(7)]
    api.set_cg_schedule({"auto": api.CG_AUTO, "classic": api.CG_CLASSIC, "one-reduction": api.CG_ONE_REDUCTION}[args.cg_schedule])
    one_red = args.solver == "cg" and (args.cg_schedule == "one-reduction" or (args.cg_schedule == "auto" and sharded))

    def solve(iters):
        m.zero_()
        torch.cuda.synchronize()
        p = api.lcg_default_parameters(epsilon=1e-300, max_iterations=iters)
        if args.solver == "cg":
            return api.lcg("lcg_hip_csr_ax", None, m, b, nloc, p, A, ws[0], ws[1], ws[2])
        if args.solver == "pcg":
            return api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, nloc, p, A)
        if args.solver == "cgs":
            return api.lcgs("lcg_hip_csr_ax", None, m, b, nloc, p, A, *ws)
        return api.lcg_solver("lcg_hip_csr_ax", None, m, b, nloc, p, A, api.LCG_BICGSTAB)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        solve(args.warmup)
    # HIP events around A.x: every call on one GPU (0.3 % of a 10M-row iteration); every 8th call when
    # sharded, where the two stream markers of a timed call are ~3 % of a 150 us iteration
    lib.lcg_hip_set_profiling(0 if os.environ.get("LCG_BENCH_NO_EVENTS") else (8 if sharded else 1))
    barrier()
    t0 = time.perf_counter()
    info = solve(args.steps)
    api.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    ax_us = lib.lcg_hip_last_ax_mean_us()
    ax_calls = lib.lcg_hip_last_ax_calls()
    lib.lcg_hip_set_profiling(0)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if info.iterations != args.steps:
        raise SystemExit(f"timed solve ran {info.iterations} iterations, expected {args.steps} (ret={info.ret})")

    # accuracy after the timed iterations (the system is solved to the fp64 floor well before 200)
    err = torch.tensor([(m - xt).pow(2).sum().item(), xt.pow(2).sum().item()], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(err)
    rel_err = float((err[0] / err[1]).sqrt().item())
    # a fast wrong answer is not a result: CG on this system is at 1e-9 of x_true after 100 iterations
    if args.solver == "cg" and args.band and args.steps >= 100 and not rel_err < 1e-8:
        raise SystemExit(f"solution check failed: |m - x_true|/|x_true| = {rel_err:.3e} after {args.steps} iterations")

    ax_per_it = {"cg": 1, "pcg": 1, "cgs": 2, "bicgstab": 2}[args.solver]
    blas1_words = {"cg": 13, "pcg": 18, "cgs": 21, "bicgstab": 22}[args.solver]
    iter_bytes = ax_per_it * spmv_bytes(n, nnz) + 8 * blas1_words * n
    out = {
        "metric": "cg_iterations_per_sec", "value": args.steps / elapsed, "unit": "iter/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic SPD CSR, {'banded-random W=%d' % args.band if args.band else 'scrambled affine maps'}, "
                               f"plain {args.solver.upper()} via lcg_hip_lcg (BASELINE configs[2]/[3])",
                   "rows": n, "nnz": nnz, "nnz_per_row": nnz / n, "solver": args.solver, "index": "int32",
                   "cg_schedule": ("one reduction per iteration (Chronopoulos-Gear)" if one_red else "classic, two reductions per iteration"),
                   "partition": "single" if not sharded else f"row-block x{world}, x exchange = {exchange} "
                                f"({lib.lcg_hip_csr_exchange_volume(A.h)} doubles received per rank per A.x) + "
                                + ("direct all-reduce(dots) over peer-mapped mailboxes, fused into the scalar step" if p2p else "RCCL all-reduce(dots)")},
        "whole_iteration_algorithmic_GBs": iter_bytes / (elapsed / args.steps) / 1e9,
        "frac_of_hbm_peak_whole_iteration": iter_bytes / (elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * world),
        "rel_err_vs_x_true": rel_err,
    }
    # dominant kernel: the CSR A.x.  Duration from HIP events on the solver stream around every
    # A.x of the timed region; bytes = algorithmic bytes of this rank's shard.
    if ax_calls > 0 and ax_us > 0:
        shard_bytes = spmv_bytes(nloc, nnz_local) if world == 1 else 12 * nnz_local + 4 * (nloc + 1) + 8 * n + 8 * nloc
        achieved = shard_bytes / (ax_us * 1e-6) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if world == 1 and os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("spmv_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": "k_spmv_ldsp / k_spmv_lds1 (LDS-staged CSR A.x; packed 18/21-bit columns when eligible)" if world == 1 else "A.x (local product + x exchange + remote columns)",
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "bytes_per_launch": shard_bytes, "avg_launch_us": ax_us, "launches": ax_calls}

    if sharded:
        # outside the timed region: what the two collectives of an iteration cost on this node's links
        # (dependent back-to-back calls; all ranks take part).  Diagnostic fields for the next tuning step.
        probe = torch.ones(8, dtype=torch.float64, device="cuda")
        xs = torch.rand(nloc, dtype=torch.float64, device="cuda"); ys = torch.empty_like(xs)
        res = {}
        for name, call, reps in (("allreduce_4_doubles_us", lambda: lib.lcg_hip_allreduce_sum(probe.data_ptr(), 4), 200),
                                 ("ax_with_exchange_us", lambda: A.spmv(xs, ys), 50)):
            for _ in range(5):
                call()
            api.synchronize(); barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            api.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / reps * 1e6], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res[name] = float(t.item())
        if p2p:     # the same all-reduce through RCCL, for comparison (all ranks switch together)
            lib.lcg_hip_p2p_enable(0)
            call = lambda: lib.lcg_hip_allreduce_sum(probe.data_ptr(), 4)
            for _ in range(5):
                call()
            api.synchronize(); barrier()
            t0 = time.perf_counter()
            for _ in range(200):
                call()
            api.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / 200 * 1e6], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            res["allreduce_4_doubles_rccl_us"] = float(t.item())
            lib.lcg_hip_p2p_enable(1)
        res["allreduce_path"] = "direct (peer mailboxes)" if p2p else "rccl"
        res["ax_with_exchange_us_by_mode"] = exchange_probe
        out["comm_probe"] = res

    if rank == 0 and world == 1 and "roofline" in out:
        # what THIS box's memory system sustains on a plain device copy (1 GiB read + 1 GiB written),
        # next to the nominal 8 TB/s the fractions above are quoted against
        src = torch.empty(1 << 27, dtype=torch.float64, device="cuda").normal_()
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dst.copy_(src)
        torch.cuda.synchronize()
        copy_gbs = 10 * 2 * src.numel() * 8 / (time.perf_counter() - t0) / 1e9
        out["roofline"]["device_copy_GBs_this_box"] = copy_gbs
        out["roofline"]["frac_of_device_copy"] = out["roofline"]["achieved"] / copy_gbs
        del src, dst

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(A, b, n, args, np)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        if p2p:
            api.synchronize(); barrier()        # nobody unmaps a mailbox a peer may still write to
            lib.lcg_hip_p2p_disconnect()
        lib.lcg_hip_comm_destroy()
        dist.destroy_process_group()


def cpu_baseline(A, b, n, args, np):
    """liblcg's own CPU loop on the same matrix, on this box's host cores (bounded sample).
    kind 'reference' = the real liblcg native/OpenMP back-end (oracle/_ref, built from
    /root/reference in the build container); 'port' = the C restatement (oracle/)."""
    from oracle import pyoracle as po
    # a one-GPU box is entitled to 16 host cores (the node shows all of them)
    cores = min(16, len(os.sched_getaffinity(0)))
    kind = "reference" if po.have_ref() else "port"
    orc = po.Oracle(kind)
    rp, ci, v = A.arrays_to_host()
    bh = b.cpu().numpy()
    sid = {"cg": po.LCG_CG, "pcg": po.LCG_PCG, "cgs": po.LCG_CGS, "bicgstab": po.LCG_BICGSTAB}[args.solver]
    jac = args.solver == "pcg"

    def set_team(k):    # libgomp is already initialised (torch loaded it): set the team size at run time
        try:
            import ctypes
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(k)
        except OSError:
            pass

    def run(iters, k):
        set_team(k)
        t0 = time.perf_counter()
        r = orc.solve(sid, rp, ci, v, bh, para=po.default_para(epsilon=1e-300, max_iterations=iters), jacobi=jac, threads=k)
        return time.perf_counter() - t0, r

    def sample(k, seconds):
        t_probe, _ = run(3, k)
        iters = int(max(5, min(400, seconds / max(t_probe / 3, 1e-3))))
        t, _ = run(iters, k)
        return iters, t
    team = cores if kind == "reference" else 1
    iters, t = sample(team, args.cpu_seconds)
    out = {"value": iters / t, "unit": "iter/s", "cores": team, "kind": kind,
           "sample": f"{iters} {args.solver.upper()} iterations of the same {n}-row system "
                     f"({'liblcg lcg_solver + OpenMP CSR callback' if kind == 'reference' else 'serial C restatement'}), {t:.1f} s"}
    if team > 1:        # all entitled cores AND one
        i1, t1 = sample(1, args.cpu_seconds / 3)
        out["single_thread"] = {"value": i1 / t1, "unit": "iter/s", "cores": 1, "sample": f"{i1} iterations, {t1:.1f} s"}
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")]
        out["cpu_model"] = f"{model[0]} ({len(model)} logical CPUs on the node)"
    except (OSError, IndexError):
        pass
    return out


if __name__ == "__main__":
    main()
