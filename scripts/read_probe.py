#!/usr/bin/env python3
"""What this box's memory system sustains: a pure read (sum of a 4 GiB fp64 array, torch's reduction and dot), a copy
(1 GiB read + 1 GiB written) and a fill -- the yardsticks beside the 8 TB/s the roofline is quoted against."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
n = 1 << 29
a = torch.ones(n, dtype=torch.float64, device="cuda")
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
out = {}
out["sum_4GiB_GBs"] = 8 * n / t(lambda: a.sum()) / 1e9
out["dot_4GiB_GBs"] = 8 * n / t(lambda: torch.dot(a[: n // 2], a[n // 2:])) / 1e9
b = torch.empty(n // 4, dtype=torch.float64, device="cuda")
out["copy_1GiB_GBs_r_plus_w"] = 2 * 8 * (n // 4) / t(lambda: b.copy_(a[: n // 4])) / 1e9
out["fill_1GiB_GBs"] = 8 * (n // 4) / t(lambda: b.fill_(1.0)) / 1e9
h = a.view(torch.float32)
out["sum_f32_4GiB_GBs"] = 8 * n / t(lambda: h.sum()) / 1e9
print(json.dumps(out))
