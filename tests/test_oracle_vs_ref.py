"""Restatement vs the compiled reference, live (build container only).

Skipped where oracle/_ref/liblcg_ref.so is absent.  On the GPU box the prebuilt
.so travels with the snapshot, so the check runs there too; nothing here reads
/root/reference at run time.
"""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.have_ref(), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def ref():
    return po.Oracle("reference")


def test_struct_sizes(ref):
    """64-byte lcg_para, 24-byte clcg_para."""
    import ctypes
    assert ref.lib.ref_sizeof_lcg_para() == ctypes.sizeof(po.Para) == 64
    assert ref.lib.ref_sizeof_clcg_para() == ctypes.sizeof(po.CPara) == 24


@pytest.mark.parametrize("sid,jac", [(0, 0), (1, 1), (2, 0), (3, 0)])
@pytest.mark.parametrize("eps,ad", [(1e-6, 0), (1e-10, 1)])
def test_real_bit_exact_random_rhs(ref, port, case10k, sid, jac, eps, ad):
    n, rp, ci, v, _, _ = case10k
    rng = np.random.default_rng(100 + sid)
    b = rng.standard_normal(n); m0 = rng.standard_normal(n) * 0.1
    para = po.default_para(epsilon=eps, abs_diff=ad)
    a = ref.solve(sid, rp, ci, v, b, m0=m0, para=para, jacobi=bool(jac))
    c = port.solve(sid, rp, ci, v, b, m0=m0, para=para, jacobi=bool(jac))
    assert (a["ret"], a["iters"], a["residual"]) == (c["ret"], c["iters"], c["residual"])
    assert np.array_equal(a["x"], c["x"])


@pytest.mark.parametrize("sid", [po.CLCG_BICG_SYM, po.CLCG_CGS, po.CLCG_TFQMR])
def test_complex_bit_exact_seed_replay(ref, port, case1kc, sid):
    n, rp, ci, v, b, _ = case1kc
    para = po.default_cpara(epsilon=1e-8, abs_diff=1)
    for _ in range(5):
        a = ref.csolve(sid, rp, ci, v, b, para=para)
        if a["seed_before"] == a["seed_after"]:
            break
    c = port.csolve(sid, rp, ci, v, b, para=para, rbar0=port.vecrnd(n, a["seed_before"]))
    assert (a["ret"], a["iters"], a["residual"]) == (c["ret"], c["iters"], c["residual"])
    assert np.array_equal(a["x"], c["x"])


def test_dot_and_coo_matvec(ref, port):
    rng = np.random.default_rng(5)
    a, b = rng.standard_normal(4097), rng.standard_normal(4097)
    assert ref.dot(a, b) == port.dot(a, b)


@pytest.mark.parametrize("tag,n,band,sym,pattern,sids", [
    ("diagonals", 4000, 300, True, 1, (0, 1, 2, 3)),
    ("scrambled", 2500, 0, True, 0, (0, 1)),
    ("row-random band", 3000, 200, True, 2, (0, 1, 3)),
    ("non-symmetric", 3000, 250, False, 1, (2, 3)),
])
def test_real_bit_exact_on_generated_systems(ref, port, tag, n, band, sym, pattern, sids):
    """The pin on MORE than the one bundled matrix: systems of the synthetic family (BASELINE configs 2-5 in small) with a random
    right-hand side and a random initial guess -- return code, count, residual and every component of x of the restatement equal
    those of the compiled liblcg, at a loose and at a tight tolerance."""
    g = port.gen_init(n, 16, band, sym, seed=7, diag_shift=0.01, pattern=pattern)
    rp, ci, v = port.gen_rows(g)
    rng = np.random.default_rng(sum(map(ord, tag)))
    b = rng.standard_normal(n); m0 = rng.standard_normal(n) * 0.1
    for sid in sids:
        for eps, ad in ((1e-6, 0), (1e-11, 1)):
            para = po.default_para(epsilon=eps, abs_diff=ad, max_iterations=400)
            a = ref.solve(sid, rp, ci, v, b, m0=m0, para=para, jacobi=(sid == 1))
            c = port.solve(sid, rp, ci, v, b, m0=m0, para=para, jacobi=(sid == 1))
            assert (a["ret"], a["iters"], a["residual"]) == (c["ret"], c["iters"], c["residual"]), (tag, sid, eps)
            assert np.array_equal(a["x"], c["x"]), (tag, sid, eps)


@pytest.mark.parametrize("sid", [po.CLCG_BICG, po.CLCG_BICGSTAB])
def test_complex_bit_exact_the_other_two(ref, port, case1kc, sid):
    """clbicg (adjoint product, no shadow residual) and clbicgstab (which does not converge on the bundled systems: both sides stop at
    the same cap with the same numbers), beside the three of test_complex_bit_exact_seed_replay."""
    n, rp, ci, v, b, _ = case1kc
    para = po.default_cpara(epsilon=1e-8, abs_diff=1, max_iterations=150)
    for _ in range(5):
        a = ref.csolve(sid, rp, ci, v, b, para=para)
        if a.get("seed_before") == a.get("seed_after"):
            break
    rbar0 = None if sid == po.CLCG_BICG else port.vecrnd(n, a["seed_before"])
    c = port.csolve(sid, rp, ci, v, b, para=para, rbar0=rbar0) if rbar0 is not None else port.csolve(sid, rp, ci, v, b, para=para)
    assert (a["ret"], a["iters"], a["residual"]) == (c["ret"], c["iters"], c["residual"])
    assert np.array_equal(a["x"], c["x"])
