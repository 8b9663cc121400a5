#!/usr/bin/env python3
"""Is the two-state behaviour of the headline product a matter of where its arrays lie?  Several instances of the same 10M-row matrix
alive at once in ONE process (different addresses), each timed with the same x / y and with vectors of its own; then again in reverse.
  python scripts/placement_lab.py [instances=5]"""
import sys, time; sys.path.insert(0, ".")
import torch
from liblcg_amd import _lib, api
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n = 10_000_000
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)


def t(A, xx, yy, reps=20):
    A.spmv(xx, yy); A.spmv(xx, yy); api.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): A.spmv(xx, yy)
    api.synchronize(); return (time.perf_counter() - t0) / reps * 1e6


import ctypes as C


class Dev:      # a device array of the library seen by torch (no copy)
    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def val_of(A):
    pr, pc, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert lib.lcg_hip_csr_arrays(A.h, C.byref(pr), C.byref(pc), C.byref(pv)) == 0
    return torch.as_tensor(Dev(pv.value, A.nnz, "<f8"), device="cuda"), pv.value


def read_rate(tsr, reps=5):
    tsr.sum(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): tsr.sum()
    torch.cuda.synchronize(); return tsr.numel() * 8 / ((time.perf_counter() - t0) / reps) / 1e12


As, vecs = [], []
for i in range(k):
    pad = torch.empty((i * 37 + 1) * 4099, dtype=torch.float64, device="cuda")     # shifts what is allocated next
    A = api.CsrMatrix.generate(n, 16, 131072, True, 1, 0.01, pattern=api.GEN_DIAGONALS)
    xi = torch.rand(n, dtype=torch.float64, device="cuda"); yi = torch.empty_like(xi)
    As.append(A); vecs.append((xi, yi, pad))
    vt, vp = val_of(A)
    print(f"instance {i}: shared x/y {t(A, x, y):6.1f} us, own x/y {t(A, xi, yi):6.1f} us; val at {vp:#x} (mod 2 MB {vp % (2 << 20):#x}), pure read of val {read_rate(vt):.2f} TB/s, "
          f"x at {xi.data_ptr():#x}, y at {yi.data_ptr():#x}", flush=True)
for i in reversed(range(k)):
    print(f"instance {i} again: shared x/y {t(As[i], x, y):6.1f} us, own x/y {t(As[i], *vecs[i][:2]):6.1f} us", flush=True)
