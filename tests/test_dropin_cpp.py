"""The C++ drop-in layer (include/lcg_dropin.hpp, include/lcg_solver_classes.hpp): programs written
against liblcg's own lcg_solver()/LCG_Solver interfaces compile with plain g++ against the headers
and link the C-ABI library; on the GPU box they reproduce the reference samples' reports."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

BIN_DIR = os.path.join(ROOT, "examples", "bin")


def _build(name):
    from liblcg_amd import _lib
    _lib.build()
    os.makedirs(BIN_DIR, exist_ok=True)
    out = os.path.join(BIN_DIR, name)
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".cpp"),
                           "-L" + os.path.join(ROOT, "liblcg_amd", "lib"), "-llcg_hip",
                           "-Wl,-rpath,$ORIGIN/../../liblcg_amd/lib", "-o", out])
    return out


@pytest.mark.parametrize("name", ["sample_csr", "sample_class"])
def test_samples_compile_with_plain_gxx_and_fail_loudly_without_gpu(name):
    exe = _build(name)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True)
    assert p.returncode == 3


@pytest.mark.gpu
def test_sample_reproduces_sample8_report():
    exe = _build("sample_csr")
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    got = dict(re.findall(r"^(\w+): ret=0 .*iterations=(\d+)", p.stdout, flags=re.M))
    # BASELINE.md 2a (eps=1e-10, abs_diff=1): CG 183, CGS 99, BiCGStab 119, PCG 181 on the real liblcg
    assert abs(int(got["CG"]) - 183) <= 3 and abs(int(got["CGS"]) - 99) <= 3 and abs(int(got["PCG"]) - 181) <= 3
    assert abs(int(got["BICGSTAB"]) - 119) <= 18
    assert "Iteration-times" in p.stderr                      # the progress callback fired


@pytest.mark.gpu
def test_class_wrappers_sample():
    """LCG_Solver / CLCG_Solver front ends (solver.h:32-283) over the same entry points."""
    exe = _build("sample_class")
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    its = dict(re.findall(r"^class (\w+)[^:]*: iterations=(\d+)", p.stdout, flags=re.M))
    assert abs(int(its["CG"]) - 183) <= 3 and abs(int(its["PCG"]) - 181) <= 3 and 1500 < int(its["TFQMR"]) < 2000
    assert "Solver: CG. Time cost:" in p.stderr and "Solver: TFQMR" in p.stderr
    assert p.stderr.count("Iteration-times: 50\t") == 1     # report interval honoured
    # LCG_Solver::MinimizeConstrained (solver.h:174-176, solver.cpp:171-212) against the real liblcg's run of the same
    # box (tests/golden/make_golden.py, box/pg_40 and box/spg_40: 40 iterations, residual 1.2648e-2), silent mode
    import numpy as np
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "ref_goldens.npz"))
    box = dict((k, float(r)) for k, r in re.findall(r"^class (S?PG): iterations=40 residual=([0-9.e+-]+)", p.stdout, flags=re.M))
    for name, tag in (("PG", "pg_40"), ("SPG", "spg_40")):
        assert abs(box[name] - g[f"box/{tag}/fl"][1]) <= 1e-8 * g[f"box/{tag}/fl"][1], (name, box)
