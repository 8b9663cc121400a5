#!/usr/bin/env python3
"""Placement study: how far away is a fast output?  The headline matrix, then NCH allocations of CH MiB one after the other (held), the
product timed into the start of each: the sequence of timings along the allocations is the map of the three groups of memory
(scripts/placement_lab7.hip) as THIS matrix sees them -- slow where a chunk shares the group of the value array.
    python scripts/placement_walk.py [--chunks 96] [--mib 1024]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from liblcg_amd import _lib, api

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--chunks", type=int, default=96)
ap.add_argument("--mib", type=int, default=1024)
args = ap.parse_args()
lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
n = args.rows
A = api.CsrMatrix.generate(n, 16, 131072, True, 1, 0.01, pattern=1)
x = torch.empty(n, dtype=torch.float64, device="cuda"); api.gen_xtrue(n, 1, 0, n, x)
api.use_torch_stream()


def spmv_us(yptr, reps=2):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    assert lib.lcg_hip_spmv(A.h, x.data_ptr(), yptr) == 0
    e0.record()
    for _ in range(reps):
        assert lib.lcg_hip_spmv(A.h, x.data_ptr(), yptr) == 0
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


work = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3)]
spmv_us(work[0].data_ptr(), 5)
print("work vectors:", " ".join(f"{spmv_us(w.data_ptr()):.0f}" for w in work), flush=True)
us = []
for i in range(args.chunks):
    p = C.c_void_p()
    if hip.hipMalloc(C.byref(p), args.mib << 20) != 0:
        print("hipMalloc failed at chunk", i); break
    us.append(spmv_us(p.value))
lo = min(us)
print(f"chunks of {args.mib} MiB, fastest {lo:.0f} us, slowest {max(us):.0f} us")
print("us:  ", " ".join(f"{u:.0f}" for u in us))
print("map: ", "".join("f" if u < lo * 1.03 else ("S" if u > lo * 1.07 else "m") for u in us))

# ---- the walk the library would make on a box whose free memory is one long stretch: candidates behind spacers of growing size
import time
if os.environ.get("WALK_GEOMETRIC", "1") == "1":
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    fr, tot = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
    print(f"free {fr.value >> 30} GiB of {tot.value >> 30}")
    held = []
    for g in (1, 2, 4, 8, 16, 32, 64):
        hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
        if fr.value < 2 * (g << 30) + (4 << 30):
            print("not enough free memory for a spacer of", g, "GiB"); break
        t0 = time.perf_counter()
        sp = C.c_void_p(); rc = hip.hipMalloc(C.byref(sp), g << 30)
        t1 = time.perf_counter()
        if rc != 0:
            print("hipMalloc failed for", g, "GiB"); break
        held.append(sp)
        p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), 1 << 30) == 0
        held.append(p)
        print(f"behind a spacer of {g:3d} GiB (hipMalloc {1e3 * (t1 - t0):.1f} ms): {spmv_us(p.value):.0f} us", flush=True)
    t0 = time.perf_counter()
    for h in held:
        hip.hipFree(h)
    torch.cuda.synchronize()
    print(f"all given back in {1e3 * (time.perf_counter() - t0):.1f} ms")
