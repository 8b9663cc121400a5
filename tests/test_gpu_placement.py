"""-m gpu: where the product's output lies (lcg_hip_set_placement, driver.hpp: Placement).
Placement hands the solver's work vectors to other ROLES -- it must not change a bit of any iterate, must leave vectors the caller
supplied where they are, must answer later solves from its memory, and must forget when vectors or matrices go away.
The reference allocates its temporaries per call and knows no such thing (lcg.cpp:158-166); the oracle is the referee of the iterates."""
import ctypes as C
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def api():
    from liblcg_amd import api as a
    assert torch.cuda.is_available()
    return a


@pytest.fixture(scope="module")
def lib():
    from liblcg_amd import _lib
    return _lib.load()


def _last(lib):
    t, mv = C.c_int(-1), C.c_int(-1)
    a, b = C.c_double(-1.0), C.c_double(-1.0)
    assert lib.lcg_hip_last_placement(C.byref(t), C.byref(mv), C.byref(a), C.byref(b)) == 0
    return t.value, mv.value, a.value, b.value


def _sha(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()


def test_placement_changes_no_bit_and_remembers(api, lib, port):
    from oracle import pyoracle as po
    n = 300_000
    A = api.CsrMatrix.generate(n, 16, 4096, True, 3, 0.01, pattern=api.GEN_DIAGONALS)
    A.build_jacobi()
    xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 3, 0, n, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    para = api.lcg_default_parameters(epsilon=1e-20, abs_diff=1, max_iterations=12)
    runs = {
        "cg": lambda m: api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, api.LCG_CG),
        "pcg": lambda m: api.lcg_solver_preconditioned("lcg_hip_csr_ax", "lcg_hip_jacobi_mx", None, m, b, n, para, A),
        "cgs": lambda m: api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, api.LCG_CGS),
        "bicgstab": lambda m: api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, api.LCG_BICGSTAB),
    }
    api.set_cg_schedule(1)      # classic: the big systems placement is for run this schedule
    try:
        ref = {}
        assert lib.lcg_hip_set_placement(0) == 0
        for name, solve in runs.items():
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            info = solve(m)
            assert _last(lib)[:2] == (0, 0)
            ref[name] = (info.ret, info.iterations, _sha(m), m.cpu().numpy())
        # the oracle's loop on the same arrays: the iterates placement must not touch are the right ones to begin with
        rp, ci, v = A.arrays_to_host()
        o = port.solve(po.LCG_CG, rp, ci, v, b.cpu().numpy(), para=po.default_para(epsilon=1e-20, abs_diff=1, max_iterations=12))
        assert np.linalg.norm(ref["cg"][3] - o["x"]) <= 1e-12 * np.linalg.norm(o["x"])
        assert lib.lcg_hip_set_placement(1) == 0      # force: this system is far below the automatic threshold
        assert lib.lcg_hip_trim() == 0                # an empty pool and an empty memory
        first = True
        for name, solve in runs.items():
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            info = solve(m)
            timed, moved, us0, us1 = _last(lib)
            assert (info.ret, info.iterations, _sha(m)) == ref[name][:3], name
            assert us0 > 0 and 0 < us1 <= us0 * 1.0001
            if first:
                assert timed >= 3       # g, d, A.d at least
                first = False
            m2 = torch.zeros(n, dtype=torch.float64, device="cuda")
            solve(m2)
            t2, mv2, _, _ = _last(lib)
            assert t2 == 0, (name, t2)          # the same vectors against the same matrix: from memory
            assert _sha(m2) == ref[name][2]
        # vectors of the caller's own keep their roles (lcg.h:135-137): nothing to place, nothing timed
        ws = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3)]
        m = torch.zeros(n, dtype=torch.float64, device="cuda")
        info = api.lcg("lcg_hip_csr_ax", None, m, b, n, para, A, *ws)
        assert _last(lib)[:2] == (0, 0) and _sha(m) == ref["cg"][2]
        # giving the idle vectors back forgets what was measured against them
        assert lib.lcg_hip_trim() == 0
        m = torch.zeros(n, dtype=torch.float64, device="cuda")
        runs["cg"](m)
        assert _last(lib)[0] >= 3 and _sha(m) == ref["cg"][2]
        assert lib.lcg_hip_set_placement(2) != 0
    finally:
        lib.lcg_hip_set_placement(-1)
        api.set_cg_schedule(0)
        A.destroy()


def test_placement_is_off_for_what_it_was_not_measured_on(api, lib, case10k):
    """automatic mode: small systems, complex systems and callbacks of the caller's own are left alone"""
    n, rp, ci, v, b, xs = case10k
    A = api.CsrMatrix.from_csr(rp, ci, v)
    m = torch.zeros(n, dtype=torch.float64, device="cuda")
    info = api.lcg_solver("lcg_hip_csr_ax", None, m, torch.from_numpy(b).cuda(), n, api.lcg_default_parameters(epsilon=1e-12, abs_diff=1), A, api.LCG_CG)
    assert info.ret == 0 and _last(lib) == (0, 0, 0.0, 0.0)
    A.destroy()


def _pool(lib):
    v, b, a = C.c_int(-1), C.c_int64(-1), C.c_int(-1)
    assert lib.lcg_hip_pool_info(C.byref(v), C.byref(b), C.byref(a)) == 0
    return v.value, b.value, a.value


def test_arena_slots_serve_solves_and_leave_together(api, lib):
    """What the placement's walk leaves behind -- one allocation cut into idle work vectors (an arena of the pool) -- without the walk:
    solves take their vectors from it (same bits), a slot in use keeps the whole arena through lcg_hip_trim, and an idle arena leaves
    as one allocation."""
    n = 200_000
    A = api.CsrMatrix.generate(n, 16, 4096, True, 5, 0.01, pattern=api.GEN_DIAGONALS)
    xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 5, 0, n, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    para = api.lcg_default_parameters(epsilon=1e-20, abs_diff=1, max_iterations=10)
    api.set_cg_schedule(1)
    try:
        assert lib.lcg_hip_set_placement(0) == 0 and lib.lcg_hip_trim() == 0
        assert _pool(lib) == (0, 0, 0)
        m = torch.zeros(n, dtype=torch.float64, device="cuda")
        api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, api.LCG_CG)
        ref = _sha(m)
        own = _pool(lib)
        assert own[0] == 3 and own[2] == 0                  # g, d, A.d: three vectors of the library's own
        assert lib.lcg_hip_trim() == 0 and _pool(lib) == (0, 0, 0)
        # an arena of five slots: CG's three vectors come out of it (nothing else is allocated), bits unchanged, with and without placement
        assert lib.lcg_hip_pool_add_arena_for_test(8 * n, 5) == 0
        assert _pool(lib)[0] == 5 and _pool(lib)[2] == 5
        for mode in (0, 1):
            assert lib.lcg_hip_set_placement(mode) == 0
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, api.LCG_CG)
            assert _sha(m) == ref and _pool(lib)[0] == 5 and _pool(lib)[2] == 5
        # BiCGStab needs six: the sixth is allocated beside the arena; after the solve everything is idle and trim takes all of it
        m = torch.zeros(n, dtype=torch.float64, device="cuda")
        api.lcg_solver("lcg_hip_csr_ax", None, m, b, n, para, A, api.LCG_BICGSTAB)
        assert _pool(lib)[0] == 6 and _pool(lib)[2] == 5
        assert lib.lcg_hip_trim() == 0 and _pool(lib) == (0, 0, 0)
        assert lib.lcg_hip_pool_add_arena_for_test(0, 3) != 0
    finally:
        lib.lcg_hip_set_placement(-1)
        api.set_cg_schedule(0)
        A.destroy()


def _walk(lib):
    ch, ms, held, found, why = C.c_int(-1), C.c_double(-1.0), C.c_int64(-1), C.c_int(-1), C.c_char_p()
    made = lib.lcg_hip_last_placement_walk(C.byref(ch), C.byref(ms), C.byref(held), C.byref(found), C.byref(why))
    return made, ch.value, ms.value, held.value, found.value, (why.value or b"").decode()


MB = 1 << 20


def test_the_walk_its_arenas_and_its_bounds(api, lib, port):
    """driver.hpp: Placement::run, the part automatic mode reaches only from 768 MB of stream: chunks allocated one after the other,
    the product timed into every fourth, the chunk found kept as an arena and cut into work vectors, a second and a third matrix,
    the idle arena evicted, lcg_hip_trim.  The test hook lowers the thresholds (64 MB chunks, 256 MB of stream) and takes the second
    timed chunk as the faster place, so the whole path runs at 2M rows whatever this box's memory looks like.  Iterates: bit-identical
    (SHA-256) to placement off, and those are the oracle's (lcg.cpp:206-264 restated) at 1e-10.  Then the hard bounds, one by one."""
    from oracle import pyoracle as po
    n, cap = 2_000_000, 6
    para = api.lcg_default_parameters(epsilon=1e-300, abs_diff=1, max_iterations=cap)
    systems = []
    api.set_cg_schedule(1)
    try:
        assert lib.lcg_hip_set_placement(0) == 0 and lib.lcg_hip_trim() == 0
        for seed in (3, 4, 5):
            A = api.CsrMatrix.generate(n, 16, 65536, True, seed, 0.01, pattern=api.GEN_DIAGONALS)
            assert A.nnz * 12 > 700 * MB
            xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, seed, 0, n, xt)
            b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            info = api.lcg("lcg_hip_csr_ax", None, m, b, n, para, A)          # the reference's plain call: library-owned work vectors
            assert (info.ret, info.iterations) == (-1019, cap) and _last(lib)[:2] == (0, 0)
            systems.append((A, b, _sha(m)))
            if seed == 3:
                rp, ci, v = A.arrays_to_host()
                o = port.solve(po.LCG_CG, rp, ci, v, b.cpu().numpy(), para=po.default_para(epsilon=1e-300, abs_diff=1, max_iterations=cap), threads=8)
                assert o["iters"] == cap and np.linalg.norm(m.cpu().numpy() - o["x"]) <= 1e-10 * np.linalg.norm(o["x"])
                del rp, ci, v

        def solve(k):
            A, b, _ = systems[k]
            m = torch.zeros(n, dtype=torch.float64, device="cuda")
            info = api.lcg("lcg_hip_csr_ax", None, m, b, n, para, A)
            assert (info.ret, info.iterations) == (-1019, cap)
            return _sha(m)

        assert lib.lcg_hip_trim() == 0 and _pool(lib) == (0, 0, 0)
        walks0 = _walk(lib)[0]
        # automatic mode, thresholds lowered: stream >= 256 MB, chunks of 64 MB, <= 24 of them, 2 s, <= 1 GiB held, 2nd timed chunk = found
        assert lib.lcg_hip_placement_tune_for_test(256 * MB, 64 * MB, 24, 2000.0, 1024 * MB, 1, 1) == 0
        assert lib.lcg_hip_set_placement(-1) == 0
        assert solve(0) == systems[0][2]
        made, chunks, ms, held, found, why = _walk(lib)
        assert made == walks0 + 1 and found == 1 and why == "found", (made, chunks, found, why)
        assert chunks == 5 and held == 5 * 64 * MB and 0 < ms <= 2000.0      # chunks 0 and 4 are timed; 0..4 were held at once
        timed, moved, us0, us1 = _last(lib)
        assert timed >= 5 and moved >= 1 and us1 < us0          # three own vectors + two chunks; the roles moved into the arena
        vecs, _, slots = _pool(lib)
        assert slots == 4 and vecs == 3 + 4                     # a 64 MB chunk holds four 16 MB vectors
        # the same matrix again: from memory, no second walk
        assert solve(0) == systems[0][2]
        assert _last(lib)[0] == 0 and _walk(lib)[0] == walks0 + 1
        # a second matrix walks for itself: a second arena
        assert solve(1) == systems[1][2]
        assert _walk(lib)[0] == walks0 + 2 and _walk(lib)[4] == 1 and _pool(lib)[2] == 8
        # a third: two arenas at most -- the older idle one is evicted (its slots leave the pool and the memory)
        assert solve(2) == systems[2][2]
        assert _walk(lib)[0] == walks0 + 3 and _pool(lib)[2] == 8
        # every system still solves to the same bits, whichever arena its memory pointed into
        for k in (0, 1, 2, 0):
            assert solve(k) == systems[k][2]
        assert _pool(lib)[2] <= 8
        assert lib.lcg_hip_trim() == 0 and _pool(lib) == (0, 0, 0)
        # ---- the bounds, one at a time (the clock decides what is "found" here: force off) ----
        for tune, check in (
            ((256 * MB, 64 * MB, 6, 2000.0, 1024 * MB, -1, 1), lambda ch, ms, held, why: ch <= 6 and held <= 6 * 64 * MB),
            ((256 * MB, 64 * MB, 24, 2000.0, 128 * MB, -1, 1), lambda ch, ms, held, why: ch <= 2 and held <= 128 * MB and why in ("hold limit", "found", "ours are the fast kind")),
            ((256 * MB, 64 * MB, 24, 0.001, 1024 * MB, -1, 1), lambda ch, ms, held, why: ch == 1 and why == "wall clock"),
        ):
            assert lib.lcg_hip_trim() == 0      # (forgets the "has had its walk" marks too)
            assert lib.lcg_hip_placement_tune_for_test(*tune) == 0
            before = _walk(lib)[0]
            assert solve(0) == systems[0][2]
            made, chunks, ms, held, found, why = _walk(lib)
            assert made == before + 1 and check(chunks, ms, held, why), (tune, chunks, ms, held, found, why)
        # production thresholds again: 792 MB streamed is above 768 MB, but a vector of 16 MB shows nothing on the clock of a ~100 us
        # product ... whatever the rules decide, the bits stay
        assert lib.lcg_hip_trim() == 0
        assert lib.lcg_hip_placement_tune_for_test(0, 0, 0, 0.0, 0, -1, 0) == 0
        assert solve(0) == systems[0][2]
    finally:
        lib.lcg_hip_placement_tune_for_test(0, 0, 0, 0.0, 0, -1, 0)
        lib.lcg_hip_set_placement(-1)
        api.set_cg_schedule(0)
        lib.lcg_hip_trim()
        for A, _, _ in systems:
            A.destroy()


def test_no_walk_on_a_shared_device(tmp_path):
    """A process that finds the device in use when it initialises the library (here: 6 GiB of the host program's own tensors; in
    the field: other ranks on the same GPU) times and deals its own vectors but never walks -- what a walk holds, the others cannot have."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import ctypes as C, json, sys, torch
sys.path.insert(0, %r)
held = torch.empty(6 << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
from liblcg_amd import _lib, api
lib = _lib.load()
MB = 1 << 20
n = 2_000_000
assert lib.lcg_hip_placement_tune_for_test(256 * MB, 64 * MB, 24, 2000.0, 1024 * MB, 1, int(sys.argv[1])) == 0
A = api.CsrMatrix.generate(n, 16, 65536, True, 3, 0.01, pattern=api.GEN_DIAGONALS)
xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 3, 0, n, xt)
b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
m = torch.zeros_like(xt)
api.set_cg_schedule(1)
info = api.lcg("lcg_hip_csr_ax", None, m, b, n, api.lcg_default_parameters(epsilon=1e-300, abs_diff=1, max_iterations=6), A)
t = C.c_int(); lib.lcg_hip_last_placement(C.byref(t), None, None, None)
print(json.dumps({"walks": lib.lcg_hip_last_placement_walk(None, None, None, None, None), "timed": t.value, "its": info.iterations}))
''' % ROOT
    out = {}
    for allow in (0, 1):
        p = subprocess.run([sys.executable, "-c", code, str(allow)], capture_output=True, text=True, timeout=280, env=dict(os.environ, LCG_HIP_DEBUG="1"))
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
        out[allow] = (json.loads(p.stdout.strip().splitlines()[-1]), p.stderr)
    assert out[0][0] == {"walks": 0, "timed": out[0][0]["timed"], "its": 6} and out[0][0]["timed"] >= 3 and "the device is shared" in out[0][1]
    assert out[1][0]["walks"] == 1 and "placement walk: 5 chunks" in out[1][1]


def test_no_walk_once_the_library_has_given_memory_back():
    """The walk is for a fresh allocator: a 1 GiB hipMalloc out of memory that was released before costs 30 ms .. 0.5 s a call (bench.py's
    variants under LCG_HIP_DEBUG=1: walks of 74 / 381 / 528 ms against a bound of 60 that can only be looked at between calls).  So the
    first large system of a process may walk; after the library has given back more than 1 GiB (matrices destroyed, vectors trimmed, the
    walk's own chunks) a later system times and deals its vectors but allocates nothing to look for a better place."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import ctypes as C, json, sys, torch
sys.path.insert(0, %r)
from liblcg_amd import _lib, api
lib = _lib.load()
MB = 1 << 20
n = 2_000_000
assert lib.lcg_hip_placement_tune_for_test(256 * MB, 64 * MB, 24, 2000.0, 1024 * MB, 1, 2) == 0      # (2: the fresh-allocator rule stays)
api.set_cg_schedule(1)
para = api.lcg_default_parameters(epsilon=1e-300, abs_diff=1, max_iterations=6)
walks = []
for seed in (3, 4, 5):
    A = api.CsrMatrix.generate(n, 16, 65536, True, seed, 0.01, pattern=api.GEN_DIAGONALS)
    xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, seed, 0, n, xt)
    b = torch.empty_like(xt); A.spmv(xt, b); api.synchronize()
    m = torch.zeros_like(xt)
    info = api.lcg("lcg_hip_csr_ax", None, m, b, n, para, A)
    assert info.iterations == 6
    walks.append(lib.lcg_hip_last_placement_walk(None, None, None, None, None))
    A.destroy()         # ~0.8 GiB of CSR arrays given back per system
print(json.dumps({"walks": walks}))
''' % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=280, env=dict(os.environ, LCG_HIP_DEBUG="1"))
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    walks = json.loads(p.stdout.strip().splitlines()[-1])["walks"]
    # the first system walks; the second may (0.8 GiB + five 64 MB chunks given back: at the limit); the third must not
    assert walks[0] == 1 and walks[2] == walks[1] <= 2 and "the library has given back" in p.stderr, (walks, p.stderr[-2000:])
