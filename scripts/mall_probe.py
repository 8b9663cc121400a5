#!/usr/bin/env python3
"""Does the Infinity Cache (MALL, 256 MB) absorb a buffer that is written and re-read at once?  Device-to-device copies of
growing size, many in a row on one stream: bytes moved (read + write) per second against the working set."""
import json, time, torch
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048):
    n = mb * (1 << 20) // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)
    for _ in range(5): b.copy_(a); a.copy_(b)
    torch.cuda.synchronize(); reps = max(10, 4096 // mb)
    t0 = time.perf_counter()
    for _ in range(reps): b.copy_(a); a.copy_(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (2 * reps)
    print(json.dumps({"buffer_MB": mb, "working_set_MB": 2 * mb, "copy_us": round(dt * 1e6, 2), "TBps_read_plus_write": round(2 * n * 8 / dt / 1e12, 2)}), flush=True)
    del a, b
