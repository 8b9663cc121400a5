"""Readers for liblcg's bundled binary systems (host-side ingest, numpy only).

File layout (reference ``data/README:1-10``; reader precedent ``sample8.cu:30-64`` /
``sample10.cu:35-69``), little endian::

    case_*_A : [N:int32][nz:int32] then nz x [row:int32][col:int32][val:f64|c128], then b[N]
    case_*_B : [N:int32] then x[N]      (the known solution)

Complex files store ``val`` and ``b``/``x`` as interleaved (re, im) doubles.
"""
from __future__ import annotations

import numpy as np


def read_coo_system(path: str, complex_values: bool = False):
    """Return (n, row, col, val, b) from a ``case_*_A`` file (COO, base 0)."""
    vt = np.complex128 if complex_values else np.float64
    rec = np.dtype([("row", "<i4"), ("col", "<i4"), ("val", vt)])
    with open(path, "rb") as f:
        n, nz = np.fromfile(f, "<i4", 2)
        ent = np.fromfile(f, rec, int(nz))
        b = np.fromfile(f, vt, int(n))
    if len(ent) != nz or len(b) != n:
        raise ValueError(f"{path}: truncated system file")
    return int(n), ent["row"].copy(), ent["col"].copy(), ent["val"].copy(), b


def read_solution(path: str, complex_values: bool = False):
    """Return x from a ``case_*_B`` file."""
    vt = np.complex128 if complex_values else np.float64
    with open(path, "rb") as f:
        n = int(np.fromfile(f, "<i4", 1)[0])
        x = np.fromfile(f, vt, n)
    if len(x) != n:
        raise ValueError(f"{path}: truncated solution file")
    return x


def coo_to_csr_host(n: int, row, col, val):
    """Stable row sort on the host (numpy).  Returns (rowptr:int32, col:int32, val).

    The device-side ingest (``lcg_hip_csr_from_coo``) is the product path for large
    systems; this helper serves small fixtures and tests.
    """
    row = np.asarray(row, np.int64)
    if row.size and (row.min() < 0 or row.max() >= n):
        raise ValueError("row index out of range")
    order = np.argsort(row, kind="stable")
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, row + 1, 1)
    np.cumsum(rowptr, out=rowptr)
    return rowptr.astype(np.int32), np.asarray(col, np.int32)[order], np.asarray(val)[order]
