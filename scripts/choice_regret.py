#!/usr/bin/env python3
"""Chosen against best forced.  The automatic choice of the A.x kernel family rests on thresholds fitted to the
synthetic families of earlier rounds; this walks a set of matrices -- the bench's variants, row-random bands of growing
width, real stencils with 1-3 unknowns per point -- and times the automatic choice against every family that accepts the matrix when
forced (plain row blocks, packed row blocks, tiled, binned; row ranges on / off for the mixed matrix).  Regret = time of the choice over
the best forced time - 1.  Exit code 1 when the worst regret exceeds --limit (default 10 %).  Every forced product is checked against the
first one (1e-12 relative to the largest entry of y).

    python scripts/choice_regret.py [--rows 10000000] [--reps 8] [--limit 0.10] >> profiles/r05_choice_regret.txt
(round 5: run at 10M, 4M, 2M and 1M rows; every set is a guard -- exit code 1 above the limit)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from liblcg_amd import _lib, api

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--limit", type=float, default=0.10)
ap.add_argument("--only", default="", help="comma-separated case names (e.g. with LCG_HIP_DEBUG=1: what the rules saw)")
args = ap.parse_args()
lib = _lib.load(); assert lib.lcg_hip_init(0) == 0
api.use_torch_stream()
n = args.rows


def stencil(nx, ny, nz, points, dof):
    dev = "cuda"
    nn = nx * ny * nz
    idx = torch.arange(nn, device=dev, dtype=torch.int64).reshape(nz, ny, nx)
    rows, cols, vals = [], [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if points == 7 and abs(dx) + abs(dy) + abs(dz) > 1:
                    continue
                src = idx[max(0, -dz):nz - max(0, dz), max(0, -dy):ny - max(0, dy), max(0, -dx):nx - max(0, dx)].reshape(-1)
                dst = idx[max(0, dz):nz - max(0, -dz), max(0, dy):ny - max(0, -dy), max(0, dx):nx - max(0, -dx)].reshape(-1)
                rows.append(src); cols.append(dst)
                vals.append(torch.full((src.numel(),), float(points) if (dx, dy, dz) == (0, 0, 0) else -1.0, device=dev, dtype=torch.float64))
    r = torch.cat(rows); c = torch.cat(cols); v = torch.cat(vals)
    if dof > 1:
        a = torch.arange(dof, device=dev, dtype=torch.int64)
        r = (r[:, None, None] * dof + a[None, :, None]).expand(-1, dof, dof).reshape(-1)
        c = (c[:, None, None] * dof + a[None, None, :]).expand(-1, dof, dof).reshape(-1)
        v = v[:, None, None].expand(-1, dof, dof).reshape(-1).clone()
        off = (r % dof) != (c % dof)
        v[off] = v[off] / (2.0 * dof)
        nn = nn * dof
    order = torch.argsort(r * nn + c)
    r, c, v = r[order], c[order], v[order]
    rp = torch.zeros(nn + 1, dtype=torch.int64, device=dev); rp[1:] = torch.cumsum(torch.bincount(r, minlength=nn), 0)
    A = api.CsrMatrix.from_csr(rp.to(torch.int32), c.to(torch.int32), v)
    del r, c, v, rp, order
    torch.cuda.empty_cache()
    return A, nn


def mixed(nrows):
    import ctypes as C
    n1 = (nrows * 4 // 5) // 64 * 64
    A1 = api.CsrMatrix.generate(n1, 16, 131072, True, 1, 0.01, pattern=api.GEN_DIAGONALS)
    A2 = api.CsrMatrix.generate(nrows - n1, 16, 0, True, 2, 0.01, pattern=api.GEN_SCRAMBLED)
    rp1, ci1, v1 = (C.c_void_p() for _ in range(3)); rp2, ci2, v2 = (C.c_void_p() for _ in range(3))
    lib.lcg_hip_csr_arrays(A1.h, C.byref(rp1), C.byref(ci1), C.byref(v1)); lib.lcg_hip_csr_arrays(A2.h, C.byref(rp2), C.byref(ci2), C.byref(v2))
    nz1, nz2 = A1.nnz, A2.nnz

    def view(p, count, dt):
        t = torch.empty(count, dtype=dt, device="cuda")
        lib.lcg_hip_memcpy(t.data_ptr(), p, t.numel() * t.element_size(), 3)
        return t
    rp = torch.cat([view(rp1, n1 + 1, torch.int32), view(rp2, nrows - n1 + 1, torch.int32)[1:] + nz1])
    ci = torch.cat([view(ci1, nz1, torch.int32), view(ci2, nz2, torch.int32) + n1])
    v = torch.cat([view(v1, nz1, torch.float64), view(v2, nz2, torch.float64)])
    A1.destroy(); A2.destroy()
    return api.CsrMatrix.from_csr(rp, ci, v, n_cols=nrows), nrows


def gen(pattern, band):
    return lambda: (api.CsrMatrix.generate(n, 16, band, True, 1, 0.01, pattern=pattern), n)


s = max(0.05, n / 1e7)      # smaller --rows shrink the stencils too (rehearsals)
g3 = lambda k: max(16, int(round(k * s ** (1 / 3))))
CASES = [("constant_diagonals", gen(api.GEN_DIAGONALS, 131072)), ("scrambled", gen(api.GEN_SCRAMBLED, 0)), ("mixed_rows", lambda: mixed(n))]
CASES += [(f"row_random_band_{b}", gen(api.GEN_ROW_RANDOM_BAND, b)) for b in (2048, 8192, 16384, 32768, 131072, 262144, 524288, 1048576) if b <= n // 2 + 24288]
CASES += [("stencil27_200^3", lambda: stencil(g3(200), g3(200), g3(200), 27, 1)), ("stencil7_200^3", lambda: stencil(g3(200), g3(200), g3(200), 7, 1)),
          ("stencil27_128x128x488", lambda: stencil(g3(128), g3(128), g3(488), 27, 1)),
          ("stencil7x3_128^3", lambda: stencil(g3(128), g3(128), g3(128), 7, 3)), ("stencil27x2_128x128x163", lambda: stencil(g3(128), g3(128), g3(163), 27, 2)),
          ("stencil27x3_128x128x163", lambda: stencil(g3(128), g3(128), g3(163), 27, 3)),
          ("laplace2d_3162^2", lambda: (api.CsrMatrix.laplace2d(int(3162 * s ** 0.5), int(3162 * s ** 0.5)), int(3162 * s ** 0.5) ** 2))]
# mode -> (packed, tiled, binned, ranges); -1 = automatic
MODES = {"auto": (-1, -1, -1, -1), "plain": (0, 0, 0, 0), "packed": (1, 0, 0, 0), "tiled": (-1, 1, 0, 0), "binned": (-1, 0, 1, 0), "ranges": (-1, -1, -1, 1)}


def product_us(A, x, y, reps):
    """median of `reps` single products (a plan stage that runs late, a clock that is still rising: one slow call does not count)"""
    A.spmv(x, y); A.spmv(x, y); A.spmv(x, y)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in ev:
        e0.record(); A.spmv(x, y); e1.record()
    torch.cuda.synchronize()
    ts = sorted(1e3 * e0.elapsed_time(e1) for e0, e1 in ev)
    return ts[len(ts) // 2]


print(f"# scripts/choice_regret.py --rows {n} --reps {args.reps}: time per product (us, HIP events, back to back); kernel family in brackets")
print(f"# {'matrix':28s} {'rows':>9s} {'entries':>10s} | " + " | ".join(f"{m:>24s}" for m in MODES) + " | regret")
worst = (0.0, None)
for name, build in CASES:
    if args.only and name not in args.only.split(","):
        continue
    A, rows = build()
    x = torch.rand(rows, dtype=torch.float64, device="cuda"); y = torch.empty_like(x); yref = None
    res = {}
    for mode, (pk, tl, bn, rg) in MODES.items():
        if mode == "ranges" and name != "mixed_rows":
            continue
        assert lib.lcg_hip_csr_set_packed(A.h, pk) == 0 and lib.lcg_hip_csr_set_tiled(A.h, tl) == 0
        assert lib.lcg_hip_csr_set_binned(A.h, bn) == 0 and lib.lcg_hip_csr_set_ranges(A.h, rg) == 0
        us = product_us(A, x, y, args.reps)
        kern = lib.lcg_hip_csr_last_kernel(A.h).decode()
        fam = ("ranges" if kern.startswith("rows [") else "binned" if "k_bin" in kern else "tiled" if "k_tile" in kern else
               "run1" if "k_spmv_run1" in kern else "packed" if "k_spmv_ldsp" in kern else "plain")
        if yref is None:
            yref = y.clone()
        else:
            err = float(((y - yref).abs().max() / yref.abs().max()).item())
            assert err <= 1e-12, (name, mode, err)
        # a forced mode that the matrix does not accept falls back to another family: it then tells nothing new
        took = mode == "auto" or fam == mode or (mode == "plain" and fam in ("plain", "run1")) or (mode == "packed" and fam in ("packed", "run1"))
        res[mode] = (us, fam, took)
    auto_us = res["auto"][0]
    best_mode, best_us = min(((m, r[0]) for m, r in res.items() if m != "auto" and r[2]), key=lambda t: t[1], default=("auto", auto_us))
    regret = auto_us / min(best_us, auto_us) - 1.0
    if regret > worst[0]:
        worst = (regret, name)
    cells = []
    for m in MODES:
        if m in res:
            us, fam, took = res[m]
            cells.append(f"{us:9.1f} [{fam:6s}]{'' if took else ' (refused)':10s}"[:24].rjust(24))
        else:
            cells.append(" " * 24)
    print(f"  {name:28s} {rows:9d} {A.nnz:10d} | " + " | ".join(cells) + f" | {100 * regret:5.1f} % (best forced: {best_mode})", flush=True)
    A.destroy(); del x, y, yref
    torch.cuda.empty_cache(); lib.lcg_hip_trim()
print(f"# worst regret {100 * worst[0]:.1f} % on {worst[1]}; limit {100 * args.limit:.0f} %")
sys.exit(1 if worst[0] > args.limit else 0)
