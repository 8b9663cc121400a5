"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module.  The product package (``liblcg_amd``) never does.

Two libraries can sit behind it:

* ``oracle/liblcg_oracle.so``  -- the C restatement (``Oracle(kind="port")``)
* ``oracle/_ref/liblcg_ref.so`` -- the real liblcg native back-end compiled from
  ``/root/reference`` (``Oracle(kind="reference")``), present only where
  ``make -C oracle ref`` could run (or where the prebuilt file travelled).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_SO = os.path.join(HERE, "liblcg_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "liblcg_ref.so")

# lcg_solver_enum (util.h:32-64) / clcg_solver_enum (util.h:187-221)
LCG_CG, LCG_PCG, LCG_CGS, LCG_BICGSTAB = 0, 1, 2, 3
CLCG_BICG, CLCG_BICG_SYM, CLCG_CGS, CLCG_BICGSTAB, CLCG_TFQMR = 0, 1, 2, 3, 4


class Para(C.Structure):          # util.h:95-148
    _fields_ = [("max_iterations", C.c_int), ("epsilon", C.c_double), ("abs_diff", C.c_int),
                ("restart_epsilon", C.c_double), ("step", C.c_double), ("sigma", C.c_double),
                ("beta", C.c_double), ("maxi_m", C.c_int)]


class CPara(C.Structure):         # util.h:247-273
    _fields_ = [("max_iterations", C.c_int), ("epsilon", C.c_double), ("abs_diff", C.c_int)]


class CsrInst(C.Structure):       # orc_csr in lcg_oracle.h
    _fields_ = [("n", C.c_int), ("rowptr", C.c_void_p), ("col", C.c_void_p), ("val", C.c_void_p),
                ("invdiag", C.c_void_p), ("threads", C.c_int), ("iters", C.c_int),
                ("last_residual", C.c_double), ("n_ax", C.c_int)]


class Gen(C.Structure):           # orc_gen in lcg_oracle.h
    _fields_ = [("n", C.c_int64), ("npairs", C.c_int), ("a", C.c_int64 * 16),
                ("ainv", C.c_int64 * 16), ("c", C.c_int64 * 16), ("banded", C.c_int),
                ("symmetric", C.c_int), ("seed", C.c_uint64), ("diag_shift", C.c_double),
                ("wb_log2", C.c_int)]


def default_para(**kw) -> Para:   # util.h:153
    p = Para(0, 1e-6, 0, 1e-6, 1.0, 0.95, 0.9, 10)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_cpara(**kw) -> CPara:  # util.h:278
    p = CPara(0, 1e-6, 0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def build(ref: bool = True) -> None:
    """Compile the restatement (always) and the reference (when its sources exist)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir("/root/reference/src/lib"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """kind='port': the C restatement.  kind='reference': compiled liblcg."""

    def __init__(self, kind: str = "port"):
        self.kind = kind
        if kind == "reference":
            if not have_ref():
                raise FileNotFoundError(REF_SO)
            self.lib = C.CDLL(REF_SO)
            self._solve = self.lib.ref_solve_csr
            self._csolve = self.lib.ref_csolve_csr
        else:
            if not os.path.exists(PORT_SO):
                build(ref=False)
            self.lib = C.CDLL(PORT_SO)
            self._solve = self.lib.orc_solve_csr
            self._csolve = self.lib.orc_csolve_csr
        self._solve.restype = C.c_int
        self._csolve.restype = C.c_int

    # ---- solves -----------------------------------------------------------
    def _inst(self, rowptr, col, val, invdiag, threads):
        n = len(rowptr) - 1
        keep = (np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32),
                np.ascontiguousarray(val),
                None if invdiag is None else np.ascontiguousarray(invdiag, np.float64))
        inst = CsrInst(n, _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]),
                       None if keep[3] is None else _ptr(keep[3]), threads, 0, 0.0, 0)
        return inst, keep

    def solve(self, solver_id, rowptr, col, val, b, m0=None, para=None, jacobi=False,
              invdiag=None, threads=1):
        """Real solve.  Returns dict(x, ret, iters, residual, n_ax)."""
        n = len(rowptr) - 1
        if jacobi and invdiag is None:
            invdiag = 1.0 / self.csr_diag(rowptr, col, val)
        inst, keep = self._inst(rowptr, col, np.asarray(val, np.float64), invdiag, threads)
        m = np.zeros(n) if m0 is None else np.array(m0, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        para = para or default_para()
        ret = self._solve(C.c_int(solver_id), C.c_int(int(jacobi)), C.byref(inst), _ptr(m), _ptr(b),
                          C.byref(para))
        return dict(x=m, ret=ret, iters=inst.iters, residual=inst.last_residual, n_ax=inst.n_ax)

    def solve_box(self, solver_id, rowptr, col, val, b, low, hig, m0=None, para=None, threads=1):
        """lcg_solver_constrained (LCG_PG = 5 / LCG_SPG = 6): box-constrained solve."""
        n = len(rowptr) - 1
        inst, keep = self._inst(rowptr, col, np.asarray(val, np.float64), None, threads)
        m = np.zeros(n) if m0 is None else np.array(m0, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        low = np.ascontiguousarray(low, np.float64); hig = np.ascontiguousarray(hig, np.float64)
        para = para or default_para()
        f = self.lib.ref_solve_csr_box if self.kind == "reference" else self.lib.orc_solve_csr_box
        f.restype = C.c_int
        ret = f(C.c_int(solver_id), C.byref(inst), _ptr(m), _ptr(b), _ptr(low), _ptr(hig), C.byref(para))
        return dict(x=m, ret=ret, iters=inst.iters, residual=inst.last_residual, n_ax=inst.n_ax)

    def csolve_pcg(self, rowptr, col, val, b, m0=None, para=None):
        """Complex-symmetric PCG with Jacobi (restated from clcg_cuda.cu:403-558; parity UNPINNED:
        the reference has no CPU implementation of it).  Port only."""
        n = len(rowptr) - 1
        val = np.ascontiguousarray(val, np.complex128)
        inv = np.ascontiguousarray(1.0 / self.csr_diag(rowptr, col, val), np.complex128)
        keep = (np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32), val, inv)
        inst = CsrInst(n, _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]), 1, 0, 0.0, 0)
        m = np.zeros(n, np.complex128) if m0 is None else np.array(m0, np.complex128)
        b = np.ascontiguousarray(b, np.complex128)
        para = para or default_cpara()
        self.lib.orc_csolve_csr_pcg.restype = C.c_int
        ret = self.lib.orc_csolve_csr_pcg(C.byref(inst), _ptr(m), _ptr(b), C.byref(para))
        return dict(x=m, ret=ret, iters=inst.iters, residual=inst.last_residual, n_ax=inst.n_ax)

    def csolve_pbicg(self, rowptr, col, val, b, m0=None, para=None):
        """Preconditioned complex BiCG with Jacobi (restated from clcg_eigen.cpp:685-802; parity UNPINNED: Eigen back-end
        only in the reference).  Port only."""
        n = len(rowptr) - 1
        val = np.ascontiguousarray(val, np.complex128)
        inv = np.ascontiguousarray(1.0 / self.csr_diag(rowptr, col, val), np.complex128)
        keep = (np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32), val, inv)
        inst = CsrInst(n, _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]), 1, 0, 0.0, 0)
        m = np.zeros(n, np.complex128) if m0 is None else np.array(m0, np.complex128)
        b = np.ascontiguousarray(b, np.complex128)
        para = para or default_cpara()
        self.lib.orc_csolve_csr_pbicg.restype = C.c_int
        ret = self.lib.orc_csolve_csr_pbicg(C.byref(inst), _ptr(m), _ptr(b), C.byref(para))
        return dict(x=m, ret=ret, iters=inst.iters, residual=inst.last_residual, n_ax=inst.n_ax)

    def csolve(self, solver_id, rowptr, col, val, b, m0=None, para=None, rbar0=None, threads=1):
        """Complex solve.  `rbar0` is required by the port for CGS/BiCGStab/TFQMR; the
        reference draws its own from time(0) and reports the bracket as seed_before/after."""
        n = len(rowptr) - 1
        inst, keep = self._inst(rowptr, col, np.asarray(val, np.complex128), None, threads)
        m = np.zeros(n, np.complex128) if m0 is None else np.array(m0, np.complex128)
        b = np.ascontiguousarray(b, np.complex128)
        para = para or default_cpara()
        out = dict()
        if self.kind == "reference":
            s0, s1 = C.c_longlong(0), C.c_longlong(0)
            ret = self._csolve(C.c_int(solver_id), C.byref(inst), _ptr(m), _ptr(b), C.byref(para),
                               C.byref(s0), C.byref(s1))
            out.update(seed_before=s0.value, seed_after=s1.value)
        else:
            if rbar0 is None and solver_id not in (CLCG_BICG, CLCG_BICG_SYM):
                raise ValueError("port needs rbar0 (use vecrnd(seed))")
            rb = None if rbar0 is None else np.ascontiguousarray(rbar0, np.complex128)
            ret = self._csolve(C.c_int(solver_id), C.byref(inst), _ptr(m), _ptr(b), C.byref(para),
                               None if rb is None else _ptr(rb))
        out.update(x=m, ret=ret, iters=inst.iters, residual=inst.last_residual, n_ax=inst.n_ax)
        return out

    # ---- primitives (always from the restatement symbols; the ref .so links them too)
    def vecrnd(self, n, seed, lo=1.0 + 0j, hi=2.0 + 0j):
        a = np.empty(n, np.complex128)
        self.lib.orc_clcg_vecrnd(_ptr(a), C.c_double(lo.real), C.c_double(lo.imag),
                                 C.c_double(hi.real), C.c_double(hi.imag), C.c_int(n),
                                 C.c_uint(seed & 0xFFFFFFFF))
        return a

    def dot(self, a, b):
        f = self.lib.ref_dot if self.kind == "reference" else self.lib.orc_dot
        f.restype = C.c_double
        a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
        return f(_ptr(a), _ptr(b), C.c_int(len(a)))

    def coo_matvec(self, row, col, val, x):
        f = self.lib.ref_coo_matvec if self.kind == "reference" else self.lib.orc_coo_matvec
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, np.float64); x = np.ascontiguousarray(x, np.float64)
        y = np.empty(len(x))
        f(_ptr(row), _ptr(col), _ptr(val), _ptr(x), _ptr(y), C.c_int(len(x)), C.c_int(len(val)))
        return y

    def csr_matvec(self, rowptr, col, val, x, threads=1):
        rowptr = np.ascontiguousarray(rowptr, np.int32); col = np.ascontiguousarray(col, np.int32)
        n = len(rowptr) - 1
        if np.iscomplexobj(val) or np.iscomplexobj(x):
            val = np.ascontiguousarray(val, np.complex128); x = np.ascontiguousarray(x, np.complex128)
            y = np.empty(n, np.complex128)
            self.lib.orc_csr_cmatvec(_ptr(rowptr), _ptr(col), _ptr(val), _ptr(x), _ptr(y), C.c_int(n),
                                     C.c_int(threads))
        else:
            val = np.ascontiguousarray(val, np.float64); x = np.ascontiguousarray(x, np.float64)
            y = np.empty(n)
            self.lib.orc_csr_matvec(_ptr(rowptr), _ptr(col), _ptr(val), _ptr(x), _ptr(y), C.c_int(n),
                                    C.c_int(threads))
        return y

    def coo_to_csr(self, row, col, n):
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        rowptr = np.empty(n + 1, np.int32); perm = np.empty(len(row), np.int32)
        rc = self.lib.orc_coo_to_csr(_ptr(row), _ptr(col), C.c_int(n), C.c_int(len(row)), _ptr(rowptr),
                                     _ptr(perm))
        if rc:
            raise ValueError("row index out of range")
        return rowptr, perm

    def csr_diag(self, rowptr, col, val):
        rowptr = np.ascontiguousarray(rowptr, np.int32); col = np.ascontiguousarray(col, np.int32)
        n = len(rowptr) - 1
        if np.iscomplexobj(val):
            val = np.ascontiguousarray(val, np.complex128); d = np.empty(n, np.complex128)
            self.lib.orc_csr_cdiag(_ptr(rowptr), _ptr(col), _ptr(val), C.c_int(n), _ptr(d))
        else:
            val = np.ascontiguousarray(val, np.float64); d = np.empty(n)
            self.lib.orc_csr_diag(_ptr(rowptr), _ptr(col), _ptr(val), C.c_int(n), _ptr(d))
        return d

    # ---- synthetic family -------------------------------------------------
    def gen_init(self, n, npairs=16, band=0, symmetric=True, seed=1, diag_shift=0.01, pattern=None) -> Gen:
        """pattern: 0 scrambled, 1 constant diagonals (offsets <= band), 2 row-random band (columns within
        +-band, drawn per row); None = 1 when band > 0 else 0 (the round-1 call)."""
        g = Gen()
        if pattern is None:
            pattern = 1 if band > 0 else 0
        self.lib.orc_gen_init_ex(C.byref(g), C.c_int64(n), C.c_int(npairs), C.c_int(pattern), C.c_int64(band),
                                 C.c_int(int(symmetric)), C.c_uint64(seed), C.c_double(diag_shift))
        return g

    def gen_rows(self, g: Gen, r0=0, r1=None):
        """CSR of global rows [r0, r1): (rowptr, col, val), global column indices."""
        r1 = g.n if r1 is None else r1
        counts = np.empty(r1 - r0, np.int32)
        self.lib.orc_gen_count(C.byref(g), C.c_int64(r0), C.c_int64(r1), _ptr(counts))
        rowptr = np.zeros(r1 - r0 + 1, np.int32)
        np.cumsum(counts, out=rowptr[1:])
        col = np.empty(rowptr[-1], np.int32); val = np.empty(rowptr[-1], np.float64)
        self.lib.orc_gen_fill(C.byref(g), C.c_int64(r0), C.c_int64(r1), _ptr(rowptr), _ptr(col), _ptr(val))
        return rowptr, col, val

    def gen_xtrue(self, g: Gen, r0=0, r1=None):
        r1 = g.n if r1 is None else r1
        x = np.empty(r1 - r0)
        self.lib.orc_gen_xtrue(C.byref(g), C.c_int64(r0), C.c_int64(r1), _ptr(x))
        return x
