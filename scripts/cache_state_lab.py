#!/usr/bin/env python3
"""Why a product inside the CG loop is slower than the same product back to back (tiled: 697 vs 610 us; 27-point stencil: 382 vs 335):
times A.x alone while, between products, (a) nothing happens, (b) x is rewritten in place, (c) 720 MB of OTHER vectors stream through
the memory system (what the iteration's vector passes move), (d) both -- and three controls: any tiny kernel, x only read, another vector rewritten.   python scripts/cache_state_lab.py [pattern=2] [band=131072]
"""
import sys
import time

sys.path.insert(0, ".")
import torch

from liblcg_amd import _lib, api

lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
pattern = int(sys.argv[1]) if len(sys.argv) > 1 else 2
band = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
n = 10_000_000
A = api.CsrMatrix.generate(n, 16, band, True, 1, 0.01, pattern=pattern)
x = torch.rand(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
bufs = [torch.rand(n, dtype=torch.float64, device="cuda") for _ in range(6)]
A.spmv(x, y); api.synchronize()
print(lib.lcg_hip_csr_last_kernel(A.h).decode())


tiny = torch.zeros(128, dtype=torch.float64, device="cuda")


def between(mode):
    if mode == "tiny":
        tiny.add_(1.0)                         # any kernel at all
    if mode == "read_x":
        tiny[0] = x.sum()                      # x read, not written
    if mode == "write_other":
        bufs[0].mul_(1.0)                      # 160 MB of another vector rewritten
    if mode in ("x", "both"):
        x.mul_(1.0)
    if mode in ("sweep", "both"):
        for i in range(0, 6, 2):
            bufs[i].add_(bufs[i + 1])          # read 2, write 1: 240 MB each, 720 MB in all
    torch.cuda.synchronize()


for mode in ("none", "tiny", "read_x", "write_other", "x", "sweep", "both", "none"):
    ts = []
    for rep in range(12):
        between(mode)
        t0 = time.perf_counter()
        A.spmv(x, y); api.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"between products: {mode:6s}  A.x median {ts[len(ts) // 2] * 1e6:7.1f} us  best {ts[0] * 1e6:7.1f}")
