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
