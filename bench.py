#!/usr/bin/env python3
"""Headline benchmark: CG iterations/s on the synthetic 10M-row, ~33 nnz/row fp64 SPD CSR.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for every N.  The first one, given --gpus N > 1 outside a launcher's environment, starts the second one
itself (N fresh child processes, one per GPU, on a free port of 127.0.0.1) BEFORE anything touches the GPU, relays rank 0's
JSON line and exits with the children's code; when they fail or overrun --budget-seconds it still prints one line
({"value": 0.0, "error": ...}).

One "step" is one CG iteration (lcg.cpp:206-264: A.d, three inner products, three vector
updates, stop test) executed by lcg_hip_lcg() through the C ABI with inputs resident in HBM.
The K-step solve is timed `--reps` times (default 5), each bracketed by barrier + synchronise;
`value` is K / the MEDIAN time (min and max are reported beside it).  A solution check that does
not depend on K guards the number: the residual a 25-iteration solve monitored must be the residual
of its iterate (recomputed with a second A.x), and 100 iterations must come within 1e-3 of x_true.

N = 1 also reports `variants`: the same CG on the three column patterns of the synthetic family
(constant diagonals -- the headline --, row-random band, scrambled) and on a matrix that mixes two of them by rows,
so that the headline cannot be mistaken for the whole family.
N > 1: the same 10M-row system is row-partitioned over the ranks (strong scaling).  The
north-star exchange -- RCCL all-gather of x + RCCL all-reduce of the dots -- is measured FIRST and
always reported (`value_rccl_allgather`); cheaper exchanges are then validated against its product
and timed, and `value` is the best validated configuration (never below the baseline).

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for every field.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
PATTERNS = {"constant_diagonals": 1, "row_random_band": 2, "scrambled": 0}
AX_PER_IT = {"cg": 1, "pcg": 1, "cgs": 2, "bicgstab": 2}
BLAS1_WORDS = {"cg": 13, "pcg": 18, "cgs": 21, "bicgstab": 22}


def spmv_bytes(n, nnz):     # 12*nnz + 4*(N+1) + 8*N (x) + 8*N (y)
    return 12 * nnz + 4 * (n + 1) + 16 * n


def workload_name(pattern, band, npairs, solver):
    what = {"constant_diagonals": f"{2 * npairs + 1} constant diagonals (a DIA matrix stored as CSR; offsets <= {band}, the same in every row)",
            "row_random_band": f"row-random band (every row draws its own ~{2 * npairs} columns within +-{band})",
            "scrambled": f"scrambled columns (~{2 * npairs} affine maps per row, anywhere in the matrix)"}[pattern]
    return f"synthetic SPD CSR, {what}, plain {solver.upper()} via lcg_hip_lcg (BASELINE configs[2]/[3])"


def emit(line):
    """The ONE line this program writes to its standard output (file descriptor saved in main())."""
    os.write(_REAL_STDOUT, (line + "\n").encode())


_REAL_STDOUT = 1
_T0 = float(os.environ.get("LCG_BENCH_T0", time.time()))      # the self-launcher hands its own start time down


def time_left(args):
    """Seconds of --budget-seconds not yet spent (the clock started when the command did)."""
    return args.budget_seconds - (time.time() - _T0)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks as FRESH child processes (this process never
    initialises the GPU -- nothing before this point imports torch), one per GPU, rendezvous on a free port of 127.0.0.1; relay rank
    0's JSON line; exit with the children's code.  A failure or an overrun still leaves ONE line on the standard output."""
    import signal
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL and the peer-mapped mailboxes need it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env["LCG_BENCH_T0"] = repr(time.time())                # the children's budget clock starts with the launcher's
    if args.one_gpu_rehearsal:
        env["LCG_HIP_RCCL_LIB"] = fake_rccl_path()
        env.setdefault("FAKE_RCCL_TIMEOUT_S", "30")
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    t0 = time.time()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True, text=True)      # stderr: inherited
    lines, result_at = [], []

    def is_result(cand):
        cand = cand.strip()
        if not (cand.startswith("{") and cand.endswith("}")):
            return False
        try:
            d = json.loads(cand)
        except ValueError:
            return False
        return "metric" in d or "dry_launch" in d

    def reader():
        for ln in p.stdout:
            lines.append(ln)
            if is_result(ln):
                result_at.append(time.time())
    th = threading.Thread(target=reader, daemon=True)
    th.start()
    killed = ""
    grace = float(os.environ.get("LCG_BENCH_EXIT_GRACE", 60.0))
    while p.poll() is None:
        now = time.time()
        if result_at and now - result_at[-1] > grace:
            killed = f"the ranks did not exit within {grace:.0f} s of their result line: ended by the launcher"
        elif now - t0 > args.budget_seconds + 60.0:
            killed = f"no result after {now - t0:.0f} s (budget {args.budget_seconds:.0f} s + 60 s): ranks ended by the launcher"
        if killed:
            for sig, wait in ((signal.SIGTERM, 15.0), (signal.SIGKILL, 15.0)):
                try:
                    os.killpg(p.pid, sig)      # exactly the session started above
                except ProcessLookupError:
                    break
                try:
                    p.wait(timeout=wait)
                    break
                except subprocess.TimeoutExpired:
                    pass
            break
        time.sleep(0.2)
    th.join(timeout=10.0)
    line = next((ln.strip() for ln in reversed(lines) if is_result(ln)), None)
    rc = p.returncode if p.returncode is not None else 1
    if line is not None and killed:         # measured, reported, and then stuck in its teardown: the measurement stands, flagged
        got = json.loads(line)
        got["launcher_note"] = killed
        line, rc = json.dumps(got), 0
    elif line is None:
        line = json.dumps({"metric": "cg_iterations_per_sec", "unit": "iter/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                           "value": 0.0, "ms_per_step": None, "error": killed or f"the ranks ended with code {rc} and no result line",
                           "config": {"workload": workload_name(args.pattern, args.band, args.npairs, args.solver)}})
        rc = rc or 1
    emit(line)
    return rc


def fake_rccl_path():
    """tests/fake_rccl/librccl_fake.so (built here if the toolchain is at hand): the stand-in for librccl that lets several ranks share
    ONE GPU.  Test infrastructure -- only --one-gpu-rehearsal ever points the library at it."""
    import subprocess
    d = os.path.join(ROOT, "tests", "fake_rccl")
    so = os.path.join(d, "librccl_fake.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", d])
    return so


def dry_launch(args):
    """The launch rehearsed on CPUs (tests/test_bench_launch.py): every rank joins a gloo group under the launcher's environment and
    rank 0 reports who came.  No GPU call, no library load."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if os.environ.get("LCG_BENCH_DRY_FAIL") == str(rank):      # lets a test watch a rank die before it reports
        raise SystemExit(3)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = torch.tensor([rank, int(os.environ.get("LOCAL_RANK", -1)), world, os.getpid()], dtype=torch.int64)
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    if rank == 0:
        emit(json.dumps({"dry_launch": True, "n_gpus": world, "gpus_asked": args.gpus, "ranks": [int(g[0]) for g in got],
                         "local_ranks": [int(g[1]) for g in got], "world_sizes": [int(g[2]) for g in got],
                         "pids": [int(g[3]) for g in got], "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                         "steps": args.steps, "warmup": args.warmup}))
    dist.barrier()
    dist.destroy_process_group()
    time.sleep(float(os.environ.get("LCG_BENCH_DRY_SLEEP", 0)))        # lets a test watch the launcher end ranks that overrun


def main():
    # Libraries (RCCL prints a version banner) write to descriptor 1 as they please: from here on descriptor 1 IS the
    # standard error, and the JSON line goes to a private copy of the original standard output.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5, help="timed repetitions of the K-step solve (value = K / median)")
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--band", type=int, default=131072)
    ap.add_argument("--pattern", default="constant_diagonals", choices=list(PATTERNS))
    ap.add_argument("--npairs", type=int, default=16)
    ap.add_argument("--solver", default="cg", choices=["cg", "pcg", "cgs", "bicgstab"])
    ap.add_argument("--cg-schedule", default="auto", choices=["auto", "classic", "one-reduction"],
                    help="lcg_hip_set_cg_schedule: auto = classic on one GPU, one all-reduce per iteration when sharded")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not collect roofline.traffic with rocprofv3 child runs (N = 1); use the committed figure")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--budget-seconds", type=float, default=540.0,
                    help="wall-clock budget of the whole run: optional parts (exchange configurations beyond the RCCL baseline, variants, "
                         "live counters) are skipped when the clock says they no longer fit; the self-launcher ends its children 60 s after it")
    ap.add_argument("--one-gpu-rehearsal", action="store_true",
                    help="REHEARSAL, not a measurement: all N ranks on GPU 0, torch side on gloo, the library's collectives from tests/fake_rccl "
                         "(the real RCCL refuses two ranks on one device) -- launch_ranks, run_sharded, votes, budget, fallbacks and the line's "
                         "fields run end to end with N > 1 before an N-GPU node has to")
    ap.add_argument("--dry-launch", action="store_true",
                    help="rehearse the launch only: every rank reports its RANK / LOCAL_RANK / WORLD_SIZE over gloo, no GPU is touched")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    if args.dry_launch:
        return dry_launch(args)

    import ctypes as C

    import numpy as np
    import torch

    from liblcg_amd import _lib, api, partition

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    args.gpus = world       # inside a launcher its environment decides
    one_gpu = args.one_gpu_rehearsal
    if one_gpu:             # every rank on GPU 0; the library binds the stand-in collectives (also when somebody else's launcher started us)
        local_rank = 0
        os.environ.setdefault("LCG_HIP_RCCL_LIB", fake_rccl_path())
    if os.environ.get("LCG_HIP_LAB") == "1":
        # the LAB build (closed experiments' knobs compiled in) is for scripts/ only: a bench line must come from the shipped library
        emit(json.dumps({"metric": "cg_iterations_per_sec", "unit": "iter/s", "value": 0.0, "n_gpus": world,
                         "error": "LCG_HIP_LAB=1 selects the LAB build of the library: bench.py measures the shipped one only"}))
        raise SystemExit(2)
    torch.cuda.set_device(local_rank)
    lib = _lib.load()
    rc = lib.lcg_hip_init(local_rank)
    if rc:
        raise SystemExit(f"lcg_hip_init failed: {lib.lcg_hip_last_error().decode()}")
    tdev = "cpu" if one_gpu else "cuda"        # where the torch-side collectives of this program live (gloo in the rehearsal)

    dist = None
    p2p, p2p_why = False, "disabled"
    sharded = world > 1 or bool(os.environ.get("LCG_HIP_FORCE_COMM"))    # the env var rehearses the RCCL path on one GPU
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        partition.init_comm_from_torch(lib)

    n = args.rows
    r0, r1 = partition.shard_range(n, world, rank)
    nloc = r1 - r0
    symmetric = args.solver in ("cg", "pcg")
    api.set_cg_schedule({"auto": api.CG_AUTO, "classic": api.CG_CLASSIC, "one-reduction": api.CG_ONE_REDUCTION}[args.cg_schedule])
    one_red = args.solver == "cg" and (args.cg_schedule == "one-reduction" or (args.cg_schedule == "auto" and sharded))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=tdev if dist is not None else "cuda")
        if dist is not None:
            dist.all_reduce(t)
        return [float(v) for v in t.tolist()]

    def mixed_rows_matrix():
        """80 % of the rows constant diagonals, the last 20 % scrambled columns (inside their own block: block-diagonal, SPD):
        two generated matrices joined on the device.  One GPU only.  What `row ranges` (lcg_hip_csr_set_ranges) are for."""
        import ctypes as C
        n1 = (n * 4 // 5) // 64 * 64
        parts = []
        for rows, band, pat, seed in ((n1, args.band, PATTERNS["constant_diagonals"], 3), (n - n1, 0, PATTERNS["scrambled"], 5)):
            G = api.CsrMatrix.generate(rows, args.npairs, band, True, seed, 0.01, pattern=pat)
            pr, pc, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
            assert lib.lcg_hip_csr_arrays(G.h, C.byref(pr), C.byref(pc), C.byref(pv)) == 0
            nz = G.nnz
            rp = torch.empty(rows + 1, dtype=torch.int32, device="cuda"); ci = torch.empty(nz, dtype=torch.int32, device="cuda")
            vv = torch.empty(nz, dtype=torch.float64, device="cuda")
            for dst, src in ((rp, pr), (ci, pc), (vv, pv)):
                assert lib.lcg_hip_memcpy(dst.data_ptr(), src, dst.numel() * dst.element_size(), 3) == 0
            G.destroy()
            parts.append((rp, ci, vv))
        (rp1, c1, v1), (rp2, c2, v2) = parts
        rp = torch.cat([rp1, rp2[1:] + int(rp1[-1].item())]); ci = torch.cat([c1, c2 + n1]); vv = torch.cat([v1, v2])
        del parts, rp1, c1, v1, rp2, c2, v2
        return api.CsrMatrix.from_csr(rp, ci, vv, n_cols=n)

    def stencil27_matrix():
        """A REAL structured matrix beside the synthetic family: the 27-point stencil of a cubic grid with about 0.8 x --rows points
        (200^3 = 8M rows, 2.1e8 entries at the default size), 27 on the diagonal and -1 elsewhere (SPD), built on the device."""
        g = max(4, int(round((0.8 * n) ** (1.0 / 3.0))))
        ns = g * g * g
        idx = torch.arange(ns, device="cuda", dtype=torch.int64).reshape(g, g, g)
        rows, cols, vals = [], [], []
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    src = idx[max(0, -dz):g - max(0, dz), max(0, -dy):g - max(0, dy), max(0, -dx):g - max(0, dx)].reshape(-1)
                    dst = idx[max(0, dz):g - max(0, -dz), max(0, dy):g - max(0, -dy), max(0, dx):g - max(0, -dx)].reshape(-1)
                    rows.append(src); cols.append(dst)
                    vals.append(torch.full((src.numel(),), 27.0 if (dx, dy, dz) == (0, 0, 0) else -1.0, device="cuda", dtype=torch.float64))
        r = torch.cat(rows); c = torch.cat(cols); v = torch.cat(vals)
        del rows, cols, vals, idx
        order = torch.argsort(r * ns + c)
        r, c, v = r[order], c[order], v[order]
        rp = torch.zeros(ns + 1, dtype=torch.int64, device="cuda"); rp[1:] = torch.cumsum(torch.bincount(r, minlength=ns), 0)
        A = api.CsrMatrix.from_csr(rp.to(torch.int32), c.to(torch.int32), v)
        return A, ns

    class System:
        """One generated system resident in HBM: A (this rank's rows), x_true, b = A.x_true, workspaces."""

        def __init__(self, pattern):
            self.pattern = pattern
            self.n, self.nloc = n, nloc
            if pattern == "mixed_rows":
                self.A = mixed_rows_matrix()
            elif pattern == "stencil27":
                self.A, self.n = stencil27_matrix()
                self.nloc = self.n
            else:
                band = args.band if PATTERNS[pattern] else 0
                self.A = api.CsrMatrix.generate(n, args.npairs, band, symmetric, 1, 0.01, r0, r1, pattern=PATTERNS[pattern])
            if args.solver == "pcg":
                self.A.build_jacobi()
            self.xt = torch.empty(self.nloc, dtype=torch.float64, device="cuda")
            if pattern == "stencil27":
                api.gen_xtrue(self.n, 1, 0, self.n, self.xt)
            else:
                api.gen_xtrue(n, 1, r0, r1, self.xt)
            self.b = torch.empty_like(self.xt)
            self.m = torch.zeros_like(self.xt)
            self.scratch = torch.empty_like(self.xt)     # (residual_check's second product; the solvers' work vectors are the library's own)
            self.placement = None
            self.walks_before = lib.lcg_hip_last_placement_walk(None, None, None, None, None)
            self.nnz_local = self.A.nnz
            self.nnz = int(allsum([self.nnz_local])[0])

        def rhs(self):
            self.A.spmv(self.xt, self.b); api.synchronize()

        def solve(self, iters, fresh=True):
            if fresh:       # the initial guess m = 0 (timed(): prepared in front of the opening barrier, not inside the timed region)
                self.m.zero_()
                torch.cuda.synchronize()
            p = api.lcg_default_parameters(epsilon=1e-300, max_iterations=iters)
            A, m, b = self.A, self.m, self.b
            # The reference's plain calls: lcg() / lcgs() with their workspace arguments left at nullptr (lcg.h:135-137, 166-169), so the
            # work vectors are the library's own (kept between solves) -- and the library may give the products' output roles to those of
            # them that are written fastest (lcg_hip_set_placement: timed once, in the FIRST solve of a system, i.e. in the warm-up)
            if args.solver == "cg":
                info = api.lcg("lcg_hip_csr_ax", None, m, b, self.nloc, p, A)
            elif args.solver == "pcg":
                info = api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, self.nloc, p, A)
            elif args.solver == "cgs":
                info = api.lcgs("lcg_hip_csr_ax", None, m, b, self.nloc, p, A)
            else:
                info = api.lcg_solver("lcg_hip_csr_ax", None, m, b, self.nloc, p, A, api.LCG_BICGSTAB)
            if self.placement is None:
                t, mv = C.c_int(0), C.c_int(0); u0, u1 = C.c_double(0.0), C.c_double(0.0)
                lib.lcg_hip_last_placement(C.byref(t), C.byref(mv), C.byref(u0), C.byref(u1))
                self.placement = {"vectors_timed_in_first_solve": t.value, "roles_moved": mv.value,
                                  "first_output_us_as_allocated": u0.value, "first_output_us_as_placed": u1.value}
                # the walk this system's first solve made, if it made one (lcg_hip.h: its hard bounds; a walk belongs to the first large
                # system of a process -- the variants that follow the headline make none)
                ch, ms, held, found, why = C.c_int(0), C.c_double(0.0), C.c_int64(0), C.c_int(0), C.c_char_p()
                if lib.lcg_hip_last_placement_walk(C.byref(ch), C.byref(ms), C.byref(held), C.byref(found), C.byref(why)) > self.walks_before:
                    self.placement["walk"] = {"chunks_of_1GiB": ch.value, "wall_ms": round(ms.value, 2), "held_at_most_GiB": round(held.value / 2**30, 1),
                                              "kept_a_chunk": bool(found.value), "ended_by": (why.value or b"").decode(), "wall_bound_ms": 60}
                else:
                    self.placement["walk"] = None
            return info

        def timed(self, steps, reps, events):
            """`reps` timed K-step solves.  Returns (times [s, max over ranks], ax_us, ax_calls) of the median run."""
            runs = []
            for _ in range(reps):
                lib.lcg_hip_set_profiling(events)
                self.m.zero_()          # input of the solve (the initial guess), resident before the timed region starts
                barrier()
                if os.environ.get("LCG_BENCH_TEST_DIE_RANK") == str(rank):
                    # test hook (tests/test_gpu_rehearsal.py): this rank dies in the middle of a timed solve -- the others must not hang
                    import threading
                    threading.Timer(0.02, lambda: os._exit(17)).start()
                t0 = time.perf_counter()
                info = self.solve(steps, fresh=False)
                api.synchronize()       # this rank's K steps are done (stream drained) ...
                mine = time.perf_counter() - t0
                barrier()               # ... everybody's are; the MAX over ranks is the job's time (the closing barrier's own
                el = allmax(mine)       # collective is not one of the K steps)
                runs.append((el, lib.lcg_hip_last_ax_mean_us(), lib.lcg_hip_last_ax_calls()))
                lib.lcg_hip_set_profiling(0)
                if info.iterations != steps:
                    raise RuntimeError(f"timed solve ran {info.iterations} iterations, expected {steps} (ret={info.ret})")
            self.last_order = [r[0] for r in runs]      # (as they were run: a box that is still warming up shows here)
            runs.sort()
            return [r[0] for r in runs], runs[len(runs) // 2][1], runs[len(runs) // 2][2]

        def rel_err(self):
            e = allsum([(self.m - self.xt).pow(2).sum().item(), self.xt.pow(2).sum().item()])
            return (e[0] / e[1]) ** 0.5

        def residual_check(self, info):
            """|A m - b| / N recomputed with a second A.x against the residual the solve monitored (abs_diff = 0 reports
            |g|^2 / max(|m|^2, 1): lcg.cpp:209) -- a stale halo or a broken kernel cannot pass this."""
            r = self.scratch
            self.A.spmv(self.m, r); api.synchronize()
            g2, m2 = allsum([(r - self.b).pow(2).sum().item(), self.m.pow(2).sum().item()])
            mine = g2 / max(m2, 1.0)
            return mine, info.residual, abs(mine - info.residual) <= 1e-6 * max(mine, info.residual) + 1e-300

        def residual_ok(self):
            """25 iterations (the recurrence's residual and the true one still agree to rounding there; at the fp64 floor
            the recurrence keeps falling and the true residual does not)."""
            info = self.solve(25)
            api.synchronize()
            return self.residual_check(info)

        def guard(self):
            """A fast wrong answer is not a result, whatever --steps is.  (1) The residual the solver monitored after 25
            iterations is the residual of its iterate, recomputed with a second A.x (to 1e-6 relative: a stale halo, a
            broken kernel or a wrong coefficient cannot pass); (2) 100 iterations from m = 0 have brought the iterate
            within 1e-3 of x_true (the family's condition number is ~3e3: CG gains ~3.5 % per iteration, 1.6e-5 after
            100 on the headline system)."""
            mine, theirs, ok = self.residual_ok()
            self.solve(100)
            api.synchronize()
            err = self.rel_err()
            out = {"rel_err_vs_x_true_after_100_iterations": err, "residual_recomputed_after_25": mine, "residual_monitored_after_25": theirs}
            if not ok or not err < 1e-3:
                raise RuntimeError(f"solution check failed on {self.pattern}: {out}")
            return out

    def iteration_bytes(nnz):
        return AX_PER_IT[args.solver] * spmv_bytes(n, nnz) + 8 * BLAS1_WORDS[args.solver] * n

    out = {"metric": "cg_iterations_per_sec", "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    S = System(args.pattern)
    exchange, comm_probe = "none", None

    try:
        if not sharded:
            S.rhs()
            if args.warmup > 0:
                S.solve(args.warmup)
            check = S.guard()
            times, ax_us, ax_calls = S.timed(args.steps, args.reps, 1)
        else:
            times, ax_us, ax_calls, check, exchange, comm_probe, p2p = run_sharded(args, S, lib, api, partition, dist, torch, n, rank,
                                                                                  barrier, allmax, allsum, out, tdev)
    except Exception as exc:        # every rank still leaves a JSON line behind (rank 0 prints it)
        out.update({"value": 0.0, "ms_per_step": None, "error": f"{type(exc).__name__}: {exc}",
                    "config": {"workload": workload_name(args.pattern, args.band, args.npairs, args.solver)}})
        if rank == 0:
            emit(json.dumps(out))
        raise

    med = statistics.median(times)
    headline_order = list(getattr(S, "last_order", [])) if not sharded else []
    nnz = S.nnz
    out.update({
        "value": args.steps / med, "ms_per_step": 1e3 * med / args.steps,
        "value_min": args.steps / max(times), "value_max": args.steps / min(times), "timed_repetitions": len(times),
        "value_by_repetition_in_order": [round(args.steps / t, 1) for t in headline_order],
        "config": {"workload": workload_name(args.pattern, args.band, args.npairs, args.solver),
                   "rows": n, "nnz": nnz, "nnz_per_row": nnz / n, "solver": args.solver, "index": "int32", "pattern": args.pattern,
                   "cg_schedule": ("one reduction per iteration (Chronopoulos-Gear)" if one_red else "classic, two reductions per iteration"),
                   "partition": "single" if not sharded else f"row-block x{world}, x exchange = {exchange} "
                                f"({lib.lcg_hip_csr_exchange_volume(S.A.h)} doubles received per rank per A.x) + "
                                + ("direct all-reduce(dots) over peer-mapped mailboxes, fused into the scalar step" if p2p else "RCCL all-reduce(dots)")},
        "whole_iteration_algorithmic_GBs": iteration_bytes(nnz) / (med / args.steps) / 1e9,
        "frac_of_hbm_peak_whole_iteration": iteration_bytes(nnz) / (med / args.steps) / 1e9 / (HBM_PEAK_GBS * world),
        "solution_check": check,
        "placement": S.placement,
    })
    # dominant kernel: the CSR A.x.  Duration from HIP events on the solver stream around every A.x of the timed
    # region (median repetition); bytes = algorithmic bytes of this rank's shard.
    if ax_calls > 0 and ax_us > 0:
        shard_bytes = spmv_bytes(nloc, S.nnz_local) if world == 1 else 12 * S.nnz_local + 4 * (nloc + 1) + 8 * n + 8 * nloc
        achieved = shard_bytes / (ax_us * 1e-6) / 1e9
        kernel = lib.lcg_hip_csr_last_kernel(S.A.h).decode()
        traffic, source = pmc_traffic(args.pattern, kernel, S.nnz_local) if world == 1 and not sharded else (None, "not collected for sharded runs")
        if world == 1 and not sharded:
            # the same counters collected NOW, on this box, by two rocprofv3 child runs of the product alone (counters cannot be read
            # from inside this process), each held to 75 s and started only while more than half of the budget is left: the
            # committed figure is the fallback, and both are always in the line under fixed keys
            out["traffic_committed"] = {"traffic": traffic, "traffic_source": source}
            if args.no_live_pmc:
                live, why = None, "--no-live-pmc"
            elif time_left(args) < 0.5 * args.budget_seconds:
                live, why = None, f"{time_left(args):.0f} s of the budget left"
            else:
                live, why = live_pmc_traffic(args, kernel)
            out["traffic_live"] = {"traffic": live, "traffic_source": why if live else f"not collected: {why}"}
            if live:
                traffic, source = live, why
            else:
                source = f"{source}; live collection skipped: {why}"
        out["roofline"] = {"bound": "hbm", "kernel": kernel if world == 1 else "A.x (local product + x exchange + remote columns): " + kernel,
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "traffic_source": source, "bytes_per_launch": shard_bytes, "avg_launch_us": ax_us,
                           "launches": ax_calls}
        out["roofline"].update(physical_fractions(lib, S.A, ax_us, traffic, sharded))
        if "k_spmv_ldsp" in kernel and "carrying the dot" in kernel:
            out["roofline"]["launch_note"] = ("avg_launch_us spans the product kernel AND k_axp_fold, the ~4 us second stage of the d.Ad sums it "
                                              "carries (rocprofv3 lists the two separately: profiles/*_kernel_stats.csv)")
    if comm_probe is not None:
        out["comm_probe"] = comm_probe
    if sharded:
        out["rccl_ranks"] = int(lib.lcg_hip_comm_size())       # what the library's RCCL communicator spans (not torch's)
        out["rccl_library"] = lib.lcg_hip_comm_library().decode()
        out["torch_world_size"] = dist.get_world_size()
    out["library"] = os.path.relpath(_lib.SO_PATH, ROOT)
    if one_gpu:
        out["rehearsal"] = (f"{world} ranks on ONE GPU, torch side on gloo, the library's collectives from tests/fake_rccl (HIP-IPC staging buffers + host "
                            "shared memory): the code path of an N-GPU run end to end -- NO number in this line is a measurement of anything")
    if not sharded and args.solver == "cg":
        # the other CG schedule on one GPU too, so that a scaling ratio can be formed schedule by schedule (N > 1 reports both as well)
        by = {"classic" if not one_red else "one_reduction": round(args.steps / med, 1)}
        other = "one_reduction" if not one_red else "classic"
        api.set_cg_schedule(api.CG_ONE_REDUCTION if other == "one_reduction" else api.CG_CLASSIC)
        try:
            S.solve(max(1, min(args.warmup, 5)))
            t_o = S.timed(args.steps, 3, 0)[0]
            by[other] = round(args.steps / sorted(t_o)[len(t_o) // 2], 1)
        finally:
            api.set_cg_schedule({"auto": api.CG_AUTO, "classic": api.CG_CLASSIC, "one-reduction": api.CG_ONE_REDUCTION}[args.cg_schedule])
        out["value_by_cg_schedule"] = by

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
        del dst
        big = torch.ones(1 << 29, dtype=torch.float64, device="cuda")      # a pure read: the sum of 4 GiB
        for _ in range(2):
            big.sum()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            big.sum()
        torch.cuda.synchronize()
        read_gbs = 5 * big.numel() * 8 / (time.perf_counter() - t0) / 1e9
        del src, big
        rf = out["roofline"]
        rf["device_copy_GBs_this_box"] = copy_gbs
        rf["device_read_GBs_this_box"] = read_gbs
        # `achieved` counts ALGORITHMIC bytes, more than a compressing kernel moves: beside a measured rate of this box
        # it is an effective rate, not a physical one -- the physical one is must_move_GBs
        rf["effective_over_device_copy"] = rf["achieved"] / copy_gbs
        rf["must_move_over_device_read"] = rf["must_move_GBs"] / read_gbs if rf.get("must_move_GBs") else None

    if world == 1 and not sharded and not args.no_variants:
        out["variants"] = variants(args, System, S, lib, api, n, spmv_bytes, iteration_bytes)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        def gpu_iterate(k):         # the iterate the GPU path holds after k iterations of the same solve (the reference's, below, beside it)
            S.solve(k); api.synchronize()
            return S.m.cpu().numpy()
        out["cpu_baseline"] = cpu_baseline(S.A, S.b, n, args, np, gpu_iterate)
        if out["cpu_baseline"].get("parity_ok") is False:
            out["error"] = ("the GPU path's iterate differs from the reference's on the same arrays by more than 1e-9 (cpu_baseline.rel_diff_vs_gpu_after_k / "
                            "_after_4): the throughput below is not a result")
            out["value_unverified"], out["value"] = out["value"], 0.0

    if rank == 0:
        emit(json.dumps(out))
    if dist is not None:
        api.synchronize(); barrier()        # nobody unmaps a mailbox a peer may still write to
        lib.lcg_hip_p2p_disconnect()
        lib.lcg_hip_comm_destroy()
        dist.destroy_process_group()


def variants(args, System, S, lib, api, n, spmv_bytes, iteration_bytes):
    """The same K-step CG on each column pattern of the family (three repetitions, median): it/s, A.x time and its
    fraction of the 8 TB/s peak on ALGORITHMIC bytes, the kernel that ran, and the same solution check.  `mixed_rows`: 80 % of the
    rows constant diagonals, 20 % scrambled -- multiplied range by range (the kernel string names the ranges); `stencil27`: a real
    27-point stencil on a cubic grid of ~0.8 x --rows points (run blocks + template blocks)."""
    res = {}
    dearest = 0.0
    for pattern in ("constant_diagonals", "row_random_band", "scrambled", "mixed_rows", "stencil27"):
        if time_left(args) < 1.5 * dearest + (0 if args.no_cpu_baseline else 1.5 * args.cpu_seconds) + 20.0:
            res[pattern] = {"skipped": f"{time_left(args):.0f} s of the budget left"}
            continue
        t_var = time.time()
        if pattern == S.pattern:
            V, own = S, False
        else:
            V, own = System(pattern), True
            V.rhs()
            V.solve(min(args.warmup, 5) or 1)
        check = V.guard() if own else None
        times, ax_us, ax_calls = V.timed(args.steps, 3, 1)
        med = sorted(times)[len(times) // 2]
        byts = spmv_bytes(V.n, V.nnz)
        entry = {"rows": V.n, "it_per_s": args.steps / med, "ax_us": ax_us, "frac": byts / (ax_us * 1e-6) / 1e9 / HBM_PEAK_GBS if ax_us > 0 else None,
                 "algorithmic_GBs": byts / (ax_us * 1e-6) / 1e9 if ax_us > 0 else None, "nnz": V.nnz,
                 "whole_iteration_algorithmic_GBs": (AX_PER_IT[args.solver] * byts + 8 * BLAS1_WORDS[args.solver] * V.n) / (med / args.steps) / 1e9,
                 "kernel": lib.lcg_hip_csr_last_kernel(V.A.h).decode()}
        entry.update(physical_fractions(lib, V.A, ax_us, pmc_traffic(pattern, entry["kernel"], V.nnz)[0], False))
        if own:
            entry["placement"] = V.placement
        if check:
            entry["solution_check"] = check
        res[pattern] = entry
        if own:
            V.A.destroy()
            del V
        dearest = max(dearest, time.time() - t_var)
    return res


def physical_fractions(lib, A, ax_us, traffic, sharded):
    """What moves, beside the contractual `frac` (algorithmic bytes of the CSR formula / time / 8 TB/s): `must_move_bytes` = what the
    kernel family that ran streams by construction (values, its own column format, row pointers, x once, y:
    lcg_hip_csr_last_traffic_model), `frac_must_move` = that / time / peak; `frac_traffic` = the PMC-counted bytes / time / peak
    (when a collection on the same kernel exists); and what the first product paid for the kernel's copy of the matrix."""
    import ctypes as C
    out = {}
    model = int(lib.lcg_hip_csr_last_traffic_model(A.h))
    if model > 0 and ax_us > 0:
        out["must_move_bytes"] = model
        out["must_move_GBs"] = model / (ax_us * 1e-6) / 1e9
        out["frac_must_move"] = out["must_move_GBs"] / HBM_PEAK_GBS
        if sharded:
            out["must_move_note"] = "local part of the shard only (the remote-column part and the exchange are not in the model)"
    if traffic and ax_us > 0:
        out["frac_traffic"] = traffic / (ax_us * 1e-6) / 1e9 / HBM_PEAK_GBS
    ms, extra = C.c_double(0.0), C.c_int64(0)
    if lib.lcg_hip_csr_plan_info(A.h, C.byref(ms), C.byref(extra)) == 0:
        out["plan_build_ms"] = ms.value
        out["plan_extra_bytes"] = extra.value
    return out


def live_pmc_traffic(args, kernel_description):
    """HBM bytes per A.x launch on THIS box: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, the program directly
    behind `--`, no trace domain beside the counters) around scripts/ax_variants.py on the same generated matrix; FETCH_SIZE x 2 +
    WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950 (KB per dispatch; wide coalesced reads are tallied at half their size).
    Returns (bytes, source) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "no rocprofv3"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself being profiled"
    family = kernel_description.split(" ")[0]       # k_spmv_ldsp, k_tile_spmv, k_bin_expand, ...
    fams = {"k_tile_spmv": ("k_tile_spmv",), "k_bin_expand": ("k_bin_expand", "k_bin_reduce")}.get(family, (family,))
    if not family.startswith("k_"):
        return None, f"kernel family not recognised in '{kernel_description[:40]}'"
    tmp = tempfile.mkdtemp(prefix="lcg_pmc_", dir="/tmp")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", os.path.join(tmp, counter), "-o", "pmc", "--",
                   sys.executable, os.path.join(ROOT, "scripts", "ax_variants.py"), "--rows", str(args.rows), "--band", str(args.band),
                   "--patterns", str(PATTERNS[args.pattern]), "--modes", "auto", "--reps", "3", "--dot", "1"]
            p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=75)
            if p.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} ended with {p.returncode}"
            got = {}
            for f in glob.glob(os.path.join(tmp, counter, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in fams):
                        got.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
            if not got:
                return None, f"no {counter} rows for {fams}"
            # one launch of the product = one launch of each kernel of the family that ran in the steady state (the kernels seen most often)
            most = max(len(v) for v in got.values())
            vals[counter] = sum(sum(v) / len(v) for v in got.values() if len(v) >= most - 1) * 1024.0
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as exc:
        return None, f"{type(exc).__name__}: {exc}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], ("live on this box: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child runs of "
                                                            "scripts/ax_variants.py on the same matrix), FETCH_SIZE x 2 + WRITE_SIZE")


def pmc_traffic(pattern, kernel, nnz=None):
    """HBM bytes per A.x launch from the committed rocprofv3 --pmc runs (FETCH_SIZE x 2 + WRITE_SIZE as the guide
    prescribes for gfx950), or (None, why).  Counters cannot be read from inside this process; the figure is tied to the
    kernel it was collected on and dropped when another kernel ran."""
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        summ = json.load(open(path))
    except (OSError, ValueError):
        return None, "profiles/pmc_summary.json missing"
    ent = summ.get("variants", {}).get(pattern)
    if not ent:
        return None, f"profiles/pmc_summary.json holds no entry for {pattern}"
    if nnz is not None and ent.get("nnz") not in (None, nnz):
        return None, f"profiles/pmc_summary.json was collected on a matrix of {ent.get('nnz')} entries, this run has {nnz}"
    def form(d):      # the product with and without the dot riding in it is the same kernel family and format (the dot adds one read of u: 2 %)
        d = d.strip().replace(" carrying the dot that follows the product", "")
        for a, b in (("18-bit packed columns", "packed columns"), ("21-bit packed columns", "packed columns")):
            d = d.replace(a, b)
        return d
    if form(kernel) != form(ent.get("kernel_description", "")):       # the library's own description of what ran (lcg_hip_csr_last_kernel)
        return None, f"profiles/pmc_summary.json was collected on '{ent.get('kernel_description', ent.get('kernel', '?'))[:70]}', this run used '{kernel[:70]}'"
    return ent.get("hbm_bytes_per_launch"), f"profiles/pmc_summary.json ({summ.get('tag')}, {ent.get('collected', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes')})"


def run_sharded(args, S, lib, api, partition, dist, torch, n, rank, barrier, allmax, allsum, out, tdev="cuda"):
    """N > 1 (or the one-rank rehearsal): baseline first, then every cheaper exchange that reproduces its product."""
    import ctypes as C
    world = dist.get_world_size()
    A, xt, b = S.A, S.xt, S.b
    labels = {0: "all-gather", 1: "neighbour ranges", 2: "direct peer writes"}
    bad = torch.zeros(1, dtype=torch.float64, device=tdev)
    sched_codes = {"classic": api.CG_CLASSIC, "one_reduction": api.CG_ONE_REDUCTION}
    sched_now = {"auto": "one_reduction", "classic": "classic", "one-reduction": "one_reduction"}[args.cg_schedule]     # (sharded: auto = one reduction)

    def both_schedules(known_value):
        """it/s of the CURRENT exchange configuration under both CG schedules (the one the line's value ran is known already): the
        1 -> N ratio can then be read classic / classic and one-reduction / one-reduction as well as best / best."""
        res = {sched_now: known_value}
        if args.solver != "cg":
            return None
        for label, code in sched_codes.items():
            if label in res:
                continue
            api.set_cg_schedule(code)
            try:
                S.solve(max(1, min(args.warmup, 5)))
                t = S.timed(args.steps, 3, 0)[0]
                res[label] = args.steps / sorted(t)[len(t) // 2]
            finally:
                api.set_cg_schedule(sched_codes[sched_now])
        return {k: round(v, 1) for k, v in res.items()}

    def anybody(failed):
        bad[0] = 1.0 if failed else 0.0
        dist.all_reduce(bad)
        return bad.item() != 0.0

    def attempt(fn, what):
        """Run fn on every rank; True only if it succeeded everywhere (exceptions become a vote, never a hang:
        every collective the library makes has a time-out or is an RCCL call all ranks reach)."""
        failed = False
        try:
            fn()
        except Exception as exc:
            print(f"[rank {rank}] {what}: {exc}", file=sys.stderr)
            failed = True
        return not anybody(failed)

    # ---- 1. the north-star configuration: RCCL all-gather of x, RCCL all-reduce of the dots --------------------------
    t_phase = time.time()
    A.distribute(n, 0)
    S.rhs()
    x2 = 2.0 * xt + 1.0
    b2 = torch.empty_like(b)
    A.spmv(x2, b2); api.synchronize()
    if args.warmup > 0:
        S.solve(args.warmup)
    check = S.guard()
    base_times, base_ax_us, base_ax_calls = S.timed(args.steps, args.reps, 8)
    base_med = sorted(base_times)[len(base_times) // 2]
    out["value_rccl_allgather"] = args.steps / base_med
    out["ms_per_step_rccl_allgather"] = 1e3 * base_med / args.steps
    tried = {"all-gather + rccl all-reduce": args.steps / base_med}
    out["value_rccl_allgather_by_cg_schedule"] = both_schedules(args.steps / base_med)
    best = (base_med, 0, False, base_times, base_ax_us, base_ax_calls)
    cost = time.time() - t_phase        # what one configuration costs on this node (plan, guard, timed solves); the dearest seen so far
    skipped = []

    # ---- 2. cheaper exchanges, each admitted only if it reproduces the all-gather product on this node ----------------
    want = int(os.environ.get("LCG_HIP_DIST_MODE", "2"))
    p2p, p2p_why = False, "disabled (LCG_HIP_P2P=0)"
    if os.environ.get("LCG_HIP_P2P", "1") != "0" and want >= 0:
        try:
            p2p, p2p_why = partition.init_p2p_from_torch(lib)
        except Exception as exc:
            p2p, p2p_why = False, f"{type(exc).__name__}: {exc}"
        p2p = not anybody(not p2p)
        if p2p:
            lib.lcg_hip_p2p_set_timeout_ms(2000)        # while the node's links are probed; 20 s afterwards
        elif rank == 0:
            print(f"[bench] direct paths not used: {p2p_why}", file=sys.stderr)
        if not p2p:
            lib.lcg_hip_p2p_disconnect()

    def close(u, v):    # the direct path adds a row's remote part in another order: rounding only
        return bool(((u - v).abs().max() <= 1e-12 * v.abs().max()).item())

    def validate(mode):
        t = [torch.empty_like(b) for _ in range(12)]
        for i, ti in enumerate(t):          # alternating inputs expose stale buffers and parity slips
            A.spmv(xt if i % 2 == 0 else x2, ti)
        api.synchronize()
        for i, ti in enumerate(t):
            ref = b if i % 2 == 0 else b2
            ok = torch.equal(ti, ref) if mode == 1 else close(ti, ref)
            if not ok:
                raise RuntimeError(f"product {i} of exchange mode {mode} differs from the all-gather product")

    def teardown_p2p(why):
        nonlocal p2p
        p2p = False
        lib.lcg_hip_p2p_enable(0)
        A.distribute(n, 0)
        lib.lcg_hip_p2p_disconnect()
        if rank == 0:
            print(f"[bench] direct paths switched off: {why}", file=sys.stderr)

    forced = "LCG_HIP_DIST_MODE" in os.environ     # an explicit request: only that exchange is tried, and it wins when it validates
    configs = []        # (exchange mode, all-reduce over the mailboxes?)
    for mode in (0, 1, 2):
        if mode > max(want, 0) or (mode == 2 and not p2p) or (forced and mode != want):
            continue
        for direct_sum in ((False, True) if p2p else (False,)):
            if mode == 0 and not direct_sum:
                continue        # the baseline, measured above
            configs.append((mode, direct_sum))
    for mode, direct_sum in reversed(configs):      # the cheapest exchange first: if the clock allows only one, it is the one that matters
        if (mode == 2 or direct_sum) and not p2p:
            continue
        name = f"{labels[mode]} + {'direct' if direct_sum else 'rccl'} all-reduce"
        # the clock: this configuration + the final measurement of the winner + the probes must still fit (every rank votes)
        if anybody(time_left(args) < 2.5 * cost + 30.0):
            skipped.append(name)
            continue
        t_phase = time.time()
        if p2p:
            lib.lcg_hip_p2p_enable(1 if (direct_sum or mode == 2) else 0)   # mode 2's plan is agreed over the mailboxes
        if not attempt(lambda: A.distribute(n, mode), f"exchange mode {mode} unavailable"):
            attempt(lambda: A.distribute(n, 0), "back to all-gather")
            continue
        if p2p and not direct_sum:
            lib.lcg_hip_p2p_enable(0)
        ok = attempt(lambda: validate(mode), f"{name} failed its check")
        if ok:
            res = {}

            def time_it():
                S.solve(max(1, min(args.warmup, 5)))
                res["t"] = S.timed(args.steps, 1, 8)
            ok = attempt(time_it, f"{name} failed while timed")
            if ok:
                def resid():
                    mine, theirs, good = S.residual_ok()
                    if not good:
                        raise RuntimeError(f"recomputed {mine:.6e} vs monitored {theirs:.6e}")
                ok = attempt(resid, f"{name}: monitored residual is not the true one")
            if ok:
                t1 = res["t"][0][0]
                tried[name] = args.steps / t1
                if t1 < best[0] or (forced and best[1] != want):
                    best = (t1, mode, direct_sum, None, None, None)
        if p2p and anybody(lib.lcg_hip_p2p_status() < 0):
            teardown_p2p("an exchange timed out")
        attempt(lambda: A.distribute(n, 0), "back to all-gather")
        cost = max(cost, time.time() - t_phase)
    # ---- 3. the best validated configuration, measured like the baseline ------------------------------------------------
    _, mode, direct_sum, times, ax_us, ax_calls = best
    if p2p:
        lib.lcg_hip_p2p_set_timeout_ms(20000)
        lib.lcg_hip_p2p_enable(1 if (direct_sum or mode == 2) else 0)
    if mode != 0 or direct_sum:
        done = attempt(lambda: A.distribute(n, mode), "re-distribute under the chosen mode")
        if done and p2p and not direct_sum:
            lib.lcg_hip_p2p_enable(0)
        res = {}

        def final():
            S.solve(max(1, min(args.warmup, 5)))
            res["t"] = S.timed(args.steps, args.reps, 8)
        if done and attempt(final, "chosen configuration failed while timed") and \
                (forced or sorted(res["t"][0])[len(res["t"][0]) // 2] < base_med):
            times, ax_us, ax_calls = res["t"]
        else:       # never report less than the baseline that was measured
            mode, direct_sum = 0, False
            if p2p:
                lib.lcg_hip_p2p_enable(0)
            attempt(lambda: A.distribute(n, 0), "back to all-gather")
            times, ax_us, ax_calls = base_times, base_ax_us, base_ax_calls
    out["value_by_cg_schedule"] = (both_schedules(args.steps / sorted(times)[len(times) // 2]) if (mode != 0 or direct_sum)
                                   else out["value_rccl_allgather_by_cg_schedule"])
    probe = {"configurations_it_per_s": {k: round(v, 1) for k, v in tried.items()},
             "configurations_skipped_for_the_clock": skipped, "seconds_per_configuration": round(cost, 1),
             "chosen": f"{labels[mode]} + {'direct' if direct_sum else 'rccl'} all-reduce",
             "direct_paths": "connected and self-tested" if p2p else f"not used: {p2p_why}"}
    # what the two collectives of an iteration cost on this node's links (dependent back-to-back calls)
    try:
        one = torch.ones(8, dtype=torch.float64, device="cuda")
        ys = torch.empty_like(xt)
        for name, call, reps in (("allreduce_4_doubles_us", lambda: lib.lcg_hip_allreduce_sum(one.data_ptr(), 4), 200),
                                 ("ax_with_exchange_us", lambda: A.spmv(xt, ys), 50)):
            for _ in range(5):
                call()
            api.synchronize(); barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
            api.synchronize()
            probe[name] = allmax((time.perf_counter() - t0) / reps * 1e6)
    except Exception as exc:
        probe["probe_error"] = str(exc)
    # ---- where an iteration goes (optional parts, dropped last-in-first-out when the clock is short; every rank votes) --------
    dropped = []

    def timed_call(call, reps=50):
        for _ in range(5):
            call()
        api.synchronize(); barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        api.synchronize()
        return allmax((time.perf_counter() - t0) / reps * 1e6)

    def chk(rc, what):
        if rc:
            raise RuntimeError(f"{what}: rc={rc}: {lib.lcg_hip_last_error().decode()}")

    if anybody(time_left(args) < 45.0):
        dropped.append("iteration_parts")
    else:
        try:
            # the NORTH-STAR configuration is the one taken apart: all-gather of x, RCCL all-reduce of the sums
            if p2p:
                lib.lcg_hip_p2p_enable(0)
            if not attempt(lambda: A.distribute(n, 0), "probe: back to all-gather"):
                raise RuntimeError("the all-gather exchange could not be set up again")
            S.solve(max(1, min(args.warmup, 5)))
            # (a) inside the loop: the K-step solve once more with an event pair around every product
            tt, ax_in_loop, _calls = S.timed(args.steps, 1, 1)
            it_us = 1e6 * tt[0] / args.steps
            v, sc, rr, pr = (C.c_int(0) for _ in range(4))
            lib.lcg_hip_last_launches(C.byref(v), C.byref(sc), C.byref(rr), C.byref(pr))
            parts = {"iteration_us": round(it_us, 2), "ax_in_loop_us": round(allmax(ax_in_loop), 2),
                     "rest_in_loop_us": round(it_us - allmax(ax_in_loop), 2),
                     "per_iteration": {"vector_passes": round(v.value / args.steps, 2), "scalar_steps": round(sc.value / args.steps, 2),
                                       "rank_reductions": round(rr.value / args.steps, 2), "products": round(pr.value / args.steps, 2)},
                     "configuration": "all-gather + rccl all-reduce (the north-star exchange), whatever configuration the line's value ran",
                     "what": "rest = vector passes + scalar steps + the reductions over ranks; the products' own kernels: roofline.kernel"}
            # (b) the product's parts alone, back to back (stream-ordered like the product; the exchange is collective)
            ys = torch.empty_like(xt)
            for label, m_ in (("all_gather", 0), ("neighbour_ranges", 1)):
                if attempt(lambda: A.distribute(n, m_), f"probe: exchange mode {m_}"):
                    parts[f"x_exchange_alone_{label}_us"] = round(timed_call(lambda: chk(lib.lcg_hip_csr_ax_part_for_probe(A.h, xt.data_ptr(), ys.data_ptr(), 1), "exchange probe")), 2)
                    parts[f"x_exchange_{label}_doubles_received"] = int(lib.lcg_hip_csr_exchange_volume(A.h))
            parts["local_column_product_alone_us"] = round(timed_call(lambda: chk(lib.lcg_hip_csr_ax_part_for_probe(A.h, xt.data_ptr(), ys.data_ptr(), 2), "local product probe")), 2)
            parts["remote_column_part_alone_us"] = round(timed_call(lambda: chk(lib.lcg_hip_csr_ax_part_for_probe(A.h, xt.data_ptr(), ys.data_ptr(), 4), "remote part probe")), 2)
            ex = parts.get("x_exchange_alone_all_gather_us", 0.0)
            parts["serial_sum_us"] = round(ex + parts["local_column_product_alone_us"] + parts["remote_column_part_alone_us"] + parts["rest_in_loop_us"], 2)
            parts["overlap_model_us"] = round(max(parts["local_column_product_alone_us"], ex + parts["remote_column_part_alone_us"]) + parts["rest_in_loop_us"], 2)
            parts["model_note"] = ("serial_sum = exchange + local + remote + rest; overlap_model = max(local, exchange + remote) + rest (the exchange and the "
                                   "remote part run on the second stream beside the local product); both to be read against iteration_us")
            probe["iteration_parts"] = parts
        except Exception as exc:
            probe["iteration_parts_error"] = f"{type(exc).__name__}: {exc}"
        # back to the configuration the line reports
        if p2p:
            lib.lcg_hip_p2p_enable(1 if (direct_sum or mode == 2) else 0)
        attempt(lambda: A.distribute(n, mode), "back to the chosen exchange")
        if p2p and not direct_sum:
            lib.lcg_hip_p2p_enable(0)
        attempt(lambda: (A.spmv(xt, torch.empty_like(xt)), api.synchronize()), "a product under the chosen exchange")    # (the handle names its kernel again)
    probe["parts_dropped_for_the_clock"] = dropped
    return times, ax_us, ax_calls, check, labels[mode], probe, (p2p and (direct_sum or mode == 2))


def cpu_baseline(A, b, n, args, np, gpu_iterate=None):
    """liblcg's own CPU loop on the same matrix, on this box's host cores (bounded sample).
    kind 'reference' = the real liblcg native/OpenMP back-end (oracle/_ref, built from
    /root/reference in the build container); 'port' = the C restatement (oracle/).
    The iterate the reference reaches in its timed sample is KEPT and compared with the GPU path's after the same number of iterations
    (`rel_diff_vs_gpu_after_k`): the headline's solver against the real liblcg on the headline's own 3.3e8 entries, in every bench
    line (rounding of 1e7-term inner products summed in another order, amplified by the recurrence: 1e-12 .. 1e-9 over tens of iterations)."""
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

    kept = {}

    def sample(k, seconds):
        t_probe, _ = run(3, k)
        iters = int(max(5, min(400, seconds / max(t_probe / 3, 1e-3))))
        t, r = run(iters, k)
        if not kept:
            kept.update(iters=iters, x=r["x"], ret=r["ret"], done=r["iters"])
        return iters, t
    team = cores if kind == "reference" else 1
    iters, t = sample(team, args.cpu_seconds)
    out = {"value": iters / t, "unit": "iter/s", "cores": team, "kind": kind,
           "sample": f"{iters} {args.solver.upper()} iterations of the same {n}-row system "
                     f"({'liblcg lcg_solver + OpenMP CSR callback' if kind == 'reference' else 'serial C restatement'}), {t:.1f} s"}
    if gpu_iterate is not None and kept and kept["done"] == kept["iters"]:
        xg = gpu_iterate(kept["iters"])
        nx = float(np.linalg.norm(kept["x"]))
        out["rel_diff_vs_gpu_after_k"] = {"k": kept["iters"], "rel_l2": float(np.linalg.norm(xg - kept["x"]) / nx) if nx > 0 else None,
                                          "max_abs": float(np.max(np.abs(xg - kept["x"]))),
                                          "what": f"|x_gpu - x_{kind}| / |x_{kind}| after the same {kept['iters']} {args.solver.upper()} iterations from m = 0 on the same arrays"}
        del xg
        # the same after FOUR iterations (far from the fixed point both sides end at: a wrong coefficient shows at 1e-1 here)
        _, r4 = run(4, team)
        if r4["iters"] == 4:
            x4 = gpu_iterate(4)
            n4 = float(np.linalg.norm(r4["x"]))
            out["rel_diff_vs_gpu_after_4"] = float(np.linalg.norm(x4 - r4["x"]) / n4) if n4 > 0 else None
            del x4
        worst = max(v for v in (out["rel_diff_vs_gpu_after_k"]["rel_l2"], out.get("rel_diff_vs_gpu_after_4")) if v is not None)
        out["parity_ok"] = bool(worst <= 1e-9)      # main() fails the line on False: a fast iterate that is not the reference's is no result
    kept.clear()
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
