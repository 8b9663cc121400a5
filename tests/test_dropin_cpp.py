"""The C++ drop-in layer (include/lcg_dropin.hpp): a program written against liblcg's own
lcg_solver()/lcg_solver_preconditioned() signatures compiles with plain g++ against the header and
links the C-ABI library; on the GPU box it reproduces sample8.cu's report."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "examples", "sample_csr.cpp")
BIN = os.path.join(ROOT, "examples", "bin", "sample_csr")


def _build():
    from liblcg_amd import _lib
    _lib.build()
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC,
                           "-L" + os.path.join(ROOT, "liblcg_amd", "lib"), "-llcg_hip",
                           "-Wl,-rpath,$ORIGIN/../../liblcg_amd/lib", "-o", BIN])


def test_sample_compiles_with_plain_gxx_and_fails_loudly_without_gpu():
    _build()
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([BIN, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True)
    assert p.returncode == 3 and "no HIP device" in p.stderr


@pytest.mark.gpu
def test_sample_reproduces_sample8_report():
    _build()
    p = subprocess.run([BIN, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    got = dict(re.findall(r"^(\w+): ret=0 .*iterations=(\d+)", p.stdout, flags=re.M))
    # BASELINE.md 2a (eps=1e-10, abs_diff=1): CG 183, CGS 99, BiCGStab 119, PCG 181 on the real liblcg
    assert abs(int(got["CG"]) - 183) <= 3 and abs(int(got["CGS"]) - 99) <= 3 and abs(int(got["PCG"]) - 181) <= 3
    assert abs(int(got["BICGSTAB"]) - 119) <= 18
    assert "Iteration-times" in p.stderr                      # the progress callback fired
