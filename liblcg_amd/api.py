"""Thin Python front of the C ABI, shaped like liblcg's entry points.

``lcg_solver`` / ``lcg_solver_preconditioned`` / ``clcg_solver`` take the same arguments, in
the same order and with the same meaning as the reference's functions (lcg.h:71-72, 90-91;
clcg.h:74-76); vectors are torch CUDA tensors (device resident, nothing copied) or numpy arrays
(host in/out, copied by the library like lcg_solver_cuda does).  torch is used for memory only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from ._lib import CAXFUNC, CPROGRESS, AXFUNC, PROGRESS, ClcgPara, LcgPara  # noqa: F401

LCG_CG, LCG_PCG, LCG_CGS, LCG_BICGSTAB, LCG_BICGSTAB2, LCG_PG, LCG_SPG = range(7)
CLCG_BICG, CLCG_BICG_SYM, CLCG_CGS, CLCG_BICGSTAB, CLCG_TFQMR, CLCG_PCG, CLCG_PBICG = range(7)
MEM_HOST, MEM_DEVICE = 0, 1
GEN_SCRAMBLED, GEN_DIAGONALS, GEN_ROW_RANDOM_BAND = 0, 1, 2


class LcgHipError(RuntimeError):
    pass


def _chk(rc: int, what: str):
    if rc <= -2000:
        raise LcgHipError(f"{what}: rc={rc}: {L.load().lcg_hip_last_error().decode()}")
    return rc


def lcg_default_parameters(**kw) -> LcgPara:
    p = L.load().lcg_hip_default_parameters()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def clcg_default_parameters(**kw) -> ClcgPara:
    p = L.load().clcg_hip_default_parameters()
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def _ptr(x):
    """(address, mem) of a torch CUDA tensor or a numpy array."""
    if isinstance(x, np.ndarray):
        return x.ctypes.data, MEM_HOST
    import torch
    if isinstance(x, torch.Tensor):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr(), (MEM_DEVICE if x.is_cuda else MEM_HOST)
    raise TypeError(type(x))


def use_torch_stream():
    """Run the library on torch's current stream (so torch ops and ours are ordered)."""
    import torch
    _chk(L.load().lcg_hip_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)), "set_stream")


class CsrMatrix:
    """An HBM-resident CSR matrix (handle ``lcg_hip_csr_t``)."""

    def __init__(self, handle: int, n_rows: int, is_complex: bool, keep=()):
        self.h = C.c_void_p(handle)
        self.n = n_rows
        self.is_complex = is_complex
        self._keep = keep

    # -- construction ---------------------------------------------------------------------
    @classmethod
    def from_csr(cls, rowptr, col, val, n_cols=None, adopt=False):
        lib = L.load()
        is_c = bool(np.iscomplexobj(val)) if isinstance(val, np.ndarray) else val.is_complex()
        if isinstance(val, np.ndarray):
            rowptr = np.ascontiguousarray(rowptr, np.int32); col = np.ascontiguousarray(col, np.int32)
            val = np.ascontiguousarray(val, np.complex128 if is_c else np.float64)
        n = len(rowptr) - 1
        nnz = len(col)
        (pr, mem), (pc, _), (pv, _) = _ptr(rowptr), _ptr(col), _ptr(val)
        h = C.c_void_p()
        _chk(lib.lcg_hip_csr_create(C.byref(h), n, n_cols or n, nnz, pr, pc, pv, int(is_c), mem, int(adopt)), "csr_create")
        return cls(h.value, n, is_c, keep=(rowptr, col, val) if adopt else ())

    @classmethod
    def from_coo(cls, n, row, col, val):
        lib = L.load()
        is_c = bool(np.iscomplexobj(val))
        row = np.ascontiguousarray(row, np.int32); col = np.ascontiguousarray(col, np.int32)
        val = np.ascontiguousarray(val, np.complex128 if is_c else np.float64)
        h = C.c_void_p()
        _chk(lib.lcg_hip_csr_from_coo(C.byref(h), n, len(row), row.ctypes.data, col.ctypes.data, val.ctypes.data,
                                      int(is_c), MEM_HOST), "csr_from_coo")
        return cls(h.value, n, is_c)

    @classmethod
    def generate(cls, n, npairs=16, band=0, symmetric=True, seed=1, diag_shift=0.01, r0=0, r1=None, pattern=None):
        """pattern: GEN_SCRAMBLED, GEN_DIAGONALS (offsets <= band, the same in every row), GEN_ROW_RANDOM_BAND
        (columns within +-band drawn per row); None = DIAGONALS when band > 0 else SCRAMBLED."""
        lib = L.load()
        r1 = n if r1 is None else r1
        if pattern is None:
            pattern = GEN_DIAGONALS if band > 0 else GEN_SCRAMBLED
        h = C.c_void_p()
        _chk(lib.lcg_hip_csr_generate_ex(C.byref(h), n, npairs, pattern, band, int(symmetric), seed, diag_shift, r0, r1), "csr_generate")
        return cls(h.value, r1 - r0, False)

    @classmethod
    def laplace2d(cls, nx, ny, r0=0, r1=None):
        lib = L.load()
        r1 = nx * ny if r1 is None else r1
        h = C.c_void_p()
        _chk(lib.lcg_hip_csr_laplace2d(C.byref(h), nx, ny, r0, r1), "csr_laplace2d")
        return cls(h.value, r1 - r0, False)

    # -- queries ---------------------------------------------------------------------------
    @property
    def nnz(self) -> int:
        return L.load().lcg_hip_csr_nnz(self.h)

    def arrays_to_host(self):
        """(rowptr, col, val) copied to numpy (tests / CPU baseline)."""
        lib = L.load()
        pr, pc, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _chk(lib.lcg_hip_csr_arrays(self.h, C.byref(pr), C.byref(pc), C.byref(pv)), "csr_arrays")
        nnz = self.nnz
        rowptr = np.empty(self.n + 1, np.int32); col = np.empty(nnz, np.int32)
        val = np.empty(nnz, np.complex128 if self.is_complex else np.float64)
        for dst, src in ((rowptr, pr), (col, pc), (val, pv)):
            _chk(lib.lcg_hip_memcpy(dst.ctypes.data, src, dst.nbytes, 2), "memcpy d2h")
        return rowptr, col, val

    def set_kernel(self, variant: int):
        _chk(L.load().lcg_hip_csr_set_kernel(self.h, variant), "set_kernel")

    def build_jacobi(self, diag_out=None):
        p = None if diag_out is None else _ptr(diag_out)[0]
        _chk(L.load().lcg_hip_csr_build_jacobi(self.h, p), "build_jacobi")

    def spmv(self, x, y):
        _chk(L.load().lcg_hip_spmv(self.h, _ptr(x)[0], _ptr(y)[0]), "spmv")

    def distribute(self, n_global: int, mode: int = 0):
        _chk(L.load().lcg_hip_csr_distribute(self.h, n_global, mode), "csr_distribute")

    def destroy(self):
        if self.h:
            L.load().lcg_hip_csr_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


@dataclass
class SolveInfo:
    ret: int
    iterations: int
    residual: float


def _cb(fn, proto):
    """None, an exported C symbol name, a raw address, or a Python callable -> (address, keepalive)."""
    lib = L.load()
    if fn is None:
        return None, None
    if isinstance(fn, str):
        return L.fnptr(lib, fn), None
    if isinstance(fn, (int, C.c_void_p)):
        return fn, None
    cfn = proto(fn)
    return C.cast(cfn, C.c_void_p), cfn


def _instance(instance):
    if isinstance(instance, CsrMatrix):
        return instance.h
    return instance


def lcg_solver(Afp, Pfp, m, B, n_size, param, instance, solver_id=LCG_CGS) -> SolveInfo:
    """lcg_solver(), lcg.h:71-72.  Afp: 'lcg_hip_csr_ax' (built-in) or a Python callable
    (instance, x_ptr, Ax_ptr, n) that launches device work on lcg_hip_get_stream()."""
    lib = L.load()
    a, k1 = _cb(Afp, AXFUNC); p, k2 = _cb(Pfp, PROGRESS)
    (pm, mem), (pb, mem_b) = _ptr(m), _ptr(B)
    if mem != mem_b:
        raise ValueError("m and B must live in the same memory space")
    rc = lib.lcg_hip_solver(a, p, pm, pb, n_size, C.byref(param) if param is not None else None,
                            _instance(instance), solver_id, mem)
    _chk(rc, "lcg_solver")
    return SolveInfo(rc, lib.lcg_hip_last_iterations(), lib.lcg_hip_last_residual())


def lcg_solver_preconditioned(Afp, Mfp, Pfp, m, B, n_size, param, instance, solver_id=LCG_PCG) -> SolveInfo:
    """lcg_solver_preconditioned(), lcg.h:90-91."""
    lib = L.load()
    a, k1 = _cb(Afp, AXFUNC); mm, k3 = _cb(Mfp, AXFUNC); p, k2 = _cb(Pfp, PROGRESS)
    (pm, mem), (pb, _) = _ptr(m), _ptr(B)
    rc = lib.lcg_hip_solver_preconditioned(a, mm, p, pm, pb, n_size, C.byref(param) if param is not None else None,
                                           _instance(instance), solver_id, mem)
    _chk(rc, "lcg_solver_preconditioned")
    return SolveInfo(rc, lib.lcg_hip_last_iterations(), lib.lcg_hip_last_residual())


def lcg_solver_constrained(Afp, Pfp, m, B, low, hig, n_size, param, instance, solver_id=LCG_PG) -> SolveInfo:
    """lcg_solver_constrained(), lcg.h:111-113 (LCG_PG / LCG_SPG)."""
    lib = L.load()
    a, k1 = _cb(Afp, AXFUNC); p, k2 = _cb(Pfp, PROGRESS)
    (pm, mem), (pb, _), (pl, _), (ph, _) = _ptr(m), _ptr(B), _ptr(low), _ptr(hig)
    rc = lib.lcg_hip_solver_constrained(a, p, pm, pb, pl, ph, n_size, C.byref(param) if param is not None else None,
                                        _instance(instance), solver_id, mem)
    _chk(rc, "lcg_solver_constrained")
    return SolveInfo(rc, lib.lcg_hip_last_iterations(), lib.lcg_hip_last_residual())


def lcg(Afp, Pfp, m, B, n_size, param, instance, Gk=None, Dk=None, ADk=None) -> SolveInfo:
    """lcg() with optional caller workspaces (device tensors), lcg.h:135-137."""
    lib = L.load()
    a, k1 = _cb(Afp, AXFUNC); p, k2 = _cb(Pfp, PROGRESS)
    (pm, mem), (pb, _) = _ptr(m), _ptr(B)
    ws = [None if w is None else _ptr(w)[0] for w in (Gk, Dk, ADk)]
    rc = lib.lcg_hip_lcg(a, p, pm, pb, n_size, C.byref(param) if param is not None else None, _instance(instance),
                         ws[0], ws[1], ws[2], mem)
    _chk(rc, "lcg")
    return SolveInfo(rc, lib.lcg_hip_last_iterations(), lib.lcg_hip_last_residual())


def lcgs(Afp, Pfp, m, B, n_size, param, instance, *workspaces) -> SolveInfo:
    """lcgs() with optional caller workspaces RK,R0T,PK,AX,UK,QK,WK, lcg.h:166-169."""
    lib = L.load()
    a, k1 = _cb(Afp, AXFUNC); p, k2 = _cb(Pfp, PROGRESS)
    (pm, mem), (pb, _) = _ptr(m), _ptr(B)
    ws = list(workspaces) + [None] * (7 - len(workspaces))
    ws = [None if w is None else _ptr(w)[0] for w in ws]
    rc = lib.lcg_hip_lcgs(a, p, pm, pb, n_size, C.byref(param) if param is not None else None, _instance(instance),
                          *ws, mem)
    _chk(rc, "lcgs")
    return SolveInfo(rc, lib.lcg_hip_last_iterations(), lib.lcg_hip_last_residual())


def clcg_solver(Afp, Pfp, m, B, n_size, param, instance, solver_id=CLCG_BICG, shadow_seed=None, shadow=None) -> SolveInfo:
    """clcg_solver(), clcg.h:74-76.  m, B: complex128 (torch CUDA or numpy)."""
    lib = L.load()
    a, k1 = _cb(Afp, CAXFUNC); p, k2 = _cb(Pfp, CPROGRESS)
    (pm, mem), (pb, _) = _ptr(m), _ptr(B)
    if shadow_seed is not None:
        lib.lcg_hip_set_shadow_seed(shadow_seed)
    if shadow is not None:
        sh = np.ascontiguousarray(shadow, np.complex128)
        lib.lcg_hip_set_shadow_vector(sh.ctypes.data, len(sh))
    rc = lib.clcg_hip_solver(a, p, pm, pb, n_size, C.byref(param) if param is not None else None,
                             _instance(instance), solver_id, mem)
    _chk(rc, "clcg_solver")
    return SolveInfo(rc, lib.lcg_hip_last_iterations(), lib.lcg_hip_last_residual())


def clcg_solver_preconditioned(Afp, Mfp, Pfp, m, B, n_size, param, instance, solver_id=CLCG_PCG) -> SolveInfo:
    """clcg_solver_preconditioned_cuda(), clcg_cuda.h:105-108 -> clpcg."""
    lib = L.load()
    a, k1 = _cb(Afp, CAXFUNC); mm, k3 = _cb(Mfp, CAXFUNC); p, k2 = _cb(Pfp, CPROGRESS)
    (pm, mem), (pb, _) = _ptr(m), _ptr(B)
    rc = lib.clcg_hip_solver_preconditioned(a, mm, p, pm, pb, n_size, C.byref(param) if param is not None else None,
                                            _instance(instance), solver_id, mem)
    _chk(rc, "clcg_solver_preconditioned")
    return SolveInfo(rc, lib.lcg_hip_last_iterations(), lib.lcg_hip_last_residual())


# ---- kernel-level helpers ------------------------------------------------------------------------
def dot(a, b) -> float:
    out = C.c_double()
    _chk(L.load().lcg_hip_dot(a.numel(), _ptr(a)[0], _ptr(b)[0], C.byref(out)), "dot")
    return out.value


def nrm2(a) -> float:
    out = C.c_double()
    _chk(L.load().lcg_hip_nrm2(a.numel(), _ptr(a)[0], C.byref(out)), "nrm2")
    return out.value


def cdot(a, b, conj=False) -> complex:
    out = (C.c_double * 2)()
    f = L.load().clcg_hip_inner if conj else L.load().clcg_hip_dot
    _chk(f(a.numel(), _ptr(a)[0], _ptr(b)[0], out), "cdot")
    return complex(out[0], out[1])


def gen_xtrue(n, seed, r0, r1, out):
    _chk(L.load().lcg_hip_gen_xtrue(n, seed, r0, r1, _ptr(out)[0]), "gen_xtrue")


def synchronize():
    _chk(L.load().lcg_hip_synchronize(), "synchronize")


CG_AUTO, CG_CLASSIC, CG_ONE_REDUCTION = 0, 1, 2


def set_cg_schedule(schedule: int):
    """lcg_hip_set_cg_schedule: classic two-reduction CG or the one-reduction rearrangement."""
    _chk(L.load().lcg_hip_set_cg_schedule(schedule), "set_cg_schedule")
