#!/usr/bin/env python3
"""Placement study: WHICH work vector's place matters inside the CG loop?  The headline matrix (--pattern 1) or the row-random band
(--pattern 2: the tiled product, THREE kinds of places -- classes A / B / C by the stand-alone product's time); candidate vectors (hipMalloc of 8 N bytes,
and the starts of 1 GiB chunks allocated one after the other) classed by the stand-alone product into them (fast / slow: the pair
property of profiles/r04_placement.txt); then lcg_hip_lcg with caller-supplied workspaces (lcg.h:135-137) in every combination of
classes for the three roles g, d, A.d -- iterations/s and the in-loop A.x time of each.
    python scripts/placement_roles.py [--rows 10000000] [--iters 60]"""
import argparse
import ctypes as C
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import time

import torch

from liblcg_amd import _lib, api

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--iters", type=int, default=60)
ap.add_argument("--chunks", type=int, default=40)
ap.add_argument("--pattern", type=int, default=1)
args = ap.parse_args()
lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
n = args.rows
A = api.CsrMatrix.generate(n, 16, 131072, True, 1, 0.01, pattern=args.pattern)
xt = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, xt)
b = torch.empty_like(xt); m = torch.zeros_like(xt)
api.use_torch_stream()
A.spmv(xt, b); api.synchronize()
lib.lcg_hip_set_placement(0)


def dev_alloc(nbytes):
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), nbytes) == 0
    return p.value


def spmv_us(xptr, yptr, reps=3):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    assert lib.lcg_hip_spmv(A.h, xptr, yptr) == 0
    e0.record()
    for _ in range(reps):
        assert lib.lcg_hip_spmv(A.h, xptr, yptr) == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


cands = [("v%d" % i, dev_alloc(8 * n)) for i in range(6)]
for i in range(args.chunks):
    p = dev_alloc(1 << 30)
    if i % 4 == 0:
        cands.append(("c%d" % i, p))
spmv_us(b.data_ptr(), cands[0][1], 5)
cls = {}
for name, p in cands:
    cls[name] = spmv_us(b.data_ptr(), p)
lo = min(cls.values())
print("stand-alone product (x = b) into each candidate:", " ".join(f"{k}:{v:.0f}" for k, v in cls.items()))
print("into m:", f"{spmv_us(b.data_ptr(), m.data_ptr()):.0f}", " with x = each candidate, y = fastest:",
      " ".join(f"{k}:{spmv_us(p, dict(cands)[min(cls, key=cls.get)]):.0f}" for k, p in cands[:8]))
ptr = dict(cands)


def run(g, d, ad):
    p = api.lcg_default_parameters(epsilon=1e-300, max_iterations=args.iters)
    best = None
    for _ in range(3):
        m.zero_(); torch.cuda.synchronize()
        lib.lcg_hip_set_profiling(1)
        t0 = time.perf_counter()
        rc = lib.lcg_hip_lcg(_lib.fnptr(lib, "lcg_hip_csr_ax"), None, m.data_ptr(), b.data_ptr(), n, C.byref(p), A.h, ptr[g], ptr[d], ptr[ad], 1)
        api.synchronize()
        el = time.perf_counter() - t0
        ax = lib.lcg_hip_last_ax_mean_us()
        lib.lcg_hip_set_profiling(0)
        if best is None or el < best[0]:
            best = (el, ax)
    return args.iters / best[0], best[1]


if args.pattern == 1:
    fast = [k for k, v in cls.items() if v < lo * 1.03]
    slow = [k for k, v in cls.items() if v > lo * 1.07]
    print("fast:", fast, "slow:", slow)
    if len(fast) < 3 or len(slow) < 3:
        print("not enough of both classes on this box"); sys.exit(0)
    print("roles  g d A.d (F = a fast vector, S = a slow one)  ->  it/s, in-loop A.x us")
    for combo in itertools.product("FS", repeat=3):
        pool = {"F": list(fast), "S": list(slow)}
        names = [pool[c].pop(0) for c in combo]
        its, ax = run(*names)
        print("  ", " ".join(combo), " ", " ".join(names), f" -> {its:7.1f} it/s  {ax:6.1f} us", flush=True)
else:
    # three kinds: A within 3 % of the best, C 11 % and more above it, B between 4.5 and 10 %
    kinds = {"A": [k for k, v in cls.items() if v < lo * 1.03], "B": [k for k, v in cls.items() if lo * 1.045 < v < lo * 1.10],
             "C": [k for k, v in cls.items() if v > lo * 1.11]}
    print("kinds:", kinds)
    have = [c for c in "ABC" if len(kinds[c]) >= 1]
    print("roles  g d A.d by kind  ->  it/s, in-loop A.x us   (a kind with fewer than three members lends the same vector to several roles' runs, never within one run)")
    for combo in itertools.product(have, repeat=3):
        pool = {c: list(kinds[c]) for c in have}
        try:
            names = [pool[c].pop(0) for c in combo]
        except IndexError:
            continue
        its, ax = run(*names)
        print("  ", " ".join(combo), " ", " ".join(names), f" -> {its:7.1f} it/s  {ax:6.1f} us", flush=True)
