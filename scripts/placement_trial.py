#!/usr/bin/env python3
"""Placement study, the step before the remedy: in a FRESH process, generate the headline matrix (10M rows, 33 constant diagonals) and
multiply it into output vectors that were allocated the way the solver allocates its work vectors (three hipMalloc calls of 8 N bytes:
g, d, A.d) and into a few allocations of OTHER sizes -- lab 6 (profiles/r04_placement.txt) showed that the slow class is a property of
the PAIR (stream that is read, buffer that is written) and that allocations of different sizes come from different stretches of memory.
Question: how often does a process find a fast output among its own work vectors, and among which extras otherwise?

    python scripts/placement_trial.py [--rows 10000000] [--reps 3]        (one line per candidate, then the summary)
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from liblcg_amd import _lib, api

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--pattern", type=int, default=1)
args = ap.parse_args()
lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
n = args.rows
A = api.CsrMatrix.generate(n, 16, 131072 if args.pattern else 0, True, 1, 0.01, pattern=args.pattern)
x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, x)
api.use_torch_stream()


def dev_alloc(nbytes):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), nbytes) == 0
    return p.value


def spmv_us(yptr, reps):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    assert lib.lcg_hip_spmv(A.h, x.data_ptr(), yptr) == 0
    e0.record()
    for _ in range(reps):
        assert lib.lcg_hip_spmv(A.h, x.data_ptr(), yptr) == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


b8 = 8 * n
cands = [("work0", b8), ("work1", b8), ("work2", b8), ("+2MB", b8 + (2 << 20)), ("x1.25", b8 * 5 // 4), ("pow2", 1 << (b8 - 1).bit_length()),
         ("x2", 2 * b8), ("x3", 3 * b8), ("pow2x4", 4 << (b8 - 1).bit_length()), ("1GiB", 1 << 30)]
ptrs = [(name, dev_alloc(sz), sz) for name, sz in cands]
y0 = torch.empty_like(x)
spmv_us(y0.data_ptr(), 5)        # plan + clocks
out = []
for rnd in range(2):
    for name, p, sz in ptrs:
        us = spmv_us(p, args.reps)
        out.append((rnd, name, us))
        print(f"round {rnd} {name:8s} {sz >> 20:5d} MB  {us:7.1f} us", flush=True)
best = min(u for _, _, u in out); worst = max(u for _, _, u in out)
work = [u for r, nm, u in out if r == 1 and nm.startswith("work")]
print(json.dumps({"best_us": round(best, 1), "worst_us": round(worst, 1), "work_us": [round(u, 1) for u in work],
                  "work_has_fast": min(work) < best * 1.03,
                  "fast_candidates": sorted({nm for r, nm, u in out if u < best * 1.03}),
                  "repeatable": all(abs(a[2] - b[2]) < 0.02 * a[2] for a, b in zip(out[:len(ptrs)], out[len(ptrs):]))}))
