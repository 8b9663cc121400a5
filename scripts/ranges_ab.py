#!/usr/bin/env python3
"""A/B of the row-range split on the mixed systems of tests/test_gpu_ranges.py (stencil rows + coupled rows): automatic choice against
ranges forced on / off.   python scripts/ranges_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from liblcg_amd import _lib, api
from test_gpu_ranges import mixed_system

lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
api.use_torch_stream()
for grid in ((64, 64, 64), (96, 96, 96), (128, 128, 128)):
    rng = np.random.default_rng(64)
    n, cut, (rp, ci, v) = mixed_system(rng, False, grid)
    A = api.CsrMatrix.from_csr(rp, ci, v)
    x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    for mode in (-1, 0, 1):
        assert lib.lcg_hip_csr_set_ranges(A.h, mode) == 0
        for _ in range(3):
            A.spmv(x, y)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
        for e0, e1 in ev:
            e0.record(); A.spmv(x, y); e1.record()
        torch.cuda.synchronize()
        ts = sorted(1e3 * e0.elapsed_time(e1) for e0, e1 in ev)
        print(f"grid {grid} rows {n} entries {len(ci)} cut {cut} ranges mode {mode:2d}: {ts[len(ts) // 2]:8.1f} us  {lib.lcg_hip_csr_last_kernel(A.h).decode()[:150]}", flush=True)
    A.destroy()
