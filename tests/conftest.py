"""Shared pytest plumbing.  `-m "not gpu"` runs everywhere; `-m gpu` needs an MI355X."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the tests hold the SHIPPED library to the oracle: the LAB build (liblcg_amd/lib/lab, closed experiments' knobs) is for scripts/ only
    if os.environ.get("LCG_HIP_LAB") == "1":
        pytest.exit("LCG_HIP_LAB=1 selects the LAB build of liblcg_hip.so: the tests run against the shipped library only", returncode=4)


@pytest.fixture(scope="session")
def port():
    """The C restatement of the reference (oracle/liblcg_oracle.so): the checker."""
    from oracle import pyoracle as po
    po.build(ref=False)
    return po.Oracle("port")


@pytest.fixture(scope="session")
def goldens():
    return np.load(os.path.join(GOLDEN, "ref_goldens.npz"))


@pytest.fixture(scope="session")
def case10k():
    """case_10K_A/B as CSR: (n, rowptr, col, val, b, x_star)."""
    from liblcg_amd.coo_io import coo_to_csr_host, read_coo_system, read_solution
    n, row, col, val, b = read_coo_system(os.path.join(GOLDEN, "case_10K_A"))
    rp, ci, v = coo_to_csr_host(n, row, col, val)
    return n, rp, ci, v, b, read_solution(os.path.join(GOLDEN, "case_10K_B"))


def _ccase(tag):
    from liblcg_amd.coo_io import coo_to_csr_host, read_coo_system, read_solution
    n, row, col, val, b = read_coo_system(os.path.join(GOLDEN, f"case_{tag}_cA"), True)
    rp, ci, v = coo_to_csr_host(n, row, col, val)
    return n, rp, ci, v, b, read_solution(os.path.join(GOLDEN, f"case_{tag}_cB"), True)


@pytest.fixture(scope="session")
def case1kc():
    return _ccase("1K")


@pytest.fixture(scope="session")
def case10kc():
    return _ccase("10K")


# The fuzz tests draw their matrices from fixed seeds (the suite is deterministic); LCG_FUZZ_SEED_OFFSET=k shifts every one of them,
# for soak runs over other draws:  for k in 1 2 3; do LCG_FUZZ_SEED_OFFSET=$k python -m pytest tests -m gpu -k "fuzz or ragged"; done
FUZZ_SEED_OFFSET = int(os.environ.get("LCG_FUZZ_SEED_OFFSET", "0"))


def check_converged_run(port, solve_gpu, sid, rp, ci, v, bh, eps, abs_diff, jacobi=False, tag=(), samples=2, wide=False, floor=1e-9, factor=50.0,
                        cache=None, xt=None, late=False):
    """A converged run of an iterative solver against the oracle's, without leaning on the two sides stopping at the SAME iteration
    or walking the SAME rounding errors for hundreds of iterations.  (Soak over shifted seeds, round 5: the earlier form -- |x - x_oracle|
    <= max(1e-9, 20 x the oracle's response to 1-ulp changes of b) on the converged iterates -- held on the suite's seeds and failed on
    four others: two runs that meet the stop rule one iteration apart differ by the size of the last step; the one-reduction CG schedule
    (the automatic choice below 2^20 rows) is another recurrence and 160 iterations later 2.7e-8 away from the classic one, 200 x what
    1 ulp of b moves the oracle; BiCGStab's late iterates move by 3 % where two perturbed oracle runs had moved by 5e-5.)  So:
      1. same return code (converged); iteration counts within a band taken from the oracle's own response to 1-ulp changes of b
         (CGS / BiCGStab -- `wide` -- whose counts move by a quarter from one rounding to the next: 30 %); the GPU's reported residual
         meets the criterion;
      2. xt given (the solution b was made from): the GPU's iterate is as close to it as the oracle's, within a factor (10; wide: 100);
      3. late=True (a recurrence that IS the oracle's: the classic CG / PCG schedule): the iterate two iterations short of the earlier
         stop, at the same count on both sides, within max(floor, factor x the oracle's response at that count, a quarter of the
         error the oracle has left there).
    The strict statement -- the first iterations walk the oracle's iterates to 1e-13 -- is made by the callers' capped runs.
    solve_gpu(max_iterations) -> (ret, iterations, residual, x); 0: to convergence.  cache: a dict the caller keeps per (system,
    solver) so that several GPU configurations of the same solve share the oracle's runs."""
    import numpy as np
    from oracle import pyoracle as po
    cache = {} if cache is None else cache

    def oracle(b, cap):
        return port.solve(sid, rp, ci, v, b, para=po.default_para(epsilon=eps, abs_diff=abs_diff, max_iterations=cap), jacobi=jacobi)

    n = len(bh)
    if "ref" not in cache:
        cache["ref"] = oracle(bh, 0)
        cache["pert"] = [bh * (1.0 + 1e-16 * np.random.default_rng(1000 + k).standard_normal(n)) for k in range(samples)]
        cache["dit"] = max(abs(oracle(p, 0)["iters"] - cache["ref"]["iters"]) for p in cache["pert"])
    ref, pert, dit = cache["ref"], cache["pert"], cache["dit"]
    ret, its, res, x = solve_gpu(0)
    assert ret == ref["ret"] == 0, (tag, ret, ref["ret"])
    band = max(3, 4 * dit, 0.3 * ref["iters"]) if wide else max(3, 3 * dit, 0.05 * ref["iters"])
    assert abs(its - ref["iters"]) <= band, (tag, its, ref["iters"], dit)
    assert res <= eps, (tag, res, eps)
    if xt is not None:
        e_gpu, e_ref = np.linalg.norm(x - xt), np.linalg.norm(ref["x"] - xt)
        assert e_gpu <= (100.0 if wide else 10.0) * max(e_ref, 1e-14 * np.linalg.norm(xt)), (tag, "distance to the solution", e_gpu, e_ref)
    K = min(its, ref["iters"]) - 2
    if not late or K < 1:
        return
    if ("late", K) not in cache:
        rK = oracle(bh, K)
        cache[("late", K)] = (rK, max(np.linalg.norm(oracle(p, K)["x"] - rK["x"]) / np.linalg.norm(rK["x"]) for p in pert))
    rK, sens = cache[("late", K)]
    retK, itsK, _, xK = solve_gpu(K)
    assert retK == rK["ret"] == -1019 and itsK == rK["iters"] == K, (tag, retK, rK["ret"], itsK, rK["iters"], K)
    # two finite-precision CG runs whose products and dots round differently drift apart by a fraction of the error that is LEFT
    # (classic CG, 163 iterations, a 20,000-row system: 1.7e-8 between the GPU's iterate and the oracle's, both 1e-6 from the solution):
    # the late iterates agree to rounding's response OR to a quarter of the oracle's remaining error, whichever is larger
    nx = np.linalg.norm(rK["x"])
    d = np.linalg.norm(xK - rK["x"])
    left = np.linalg.norm(rK["x"] - xt) if xt is not None else 0.0
    assert d <= max(floor * nx, factor * sens * nx, 0.25 * left), (tag, "late iterate", K, d / nx, sens, left / nx)
