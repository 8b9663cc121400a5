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
