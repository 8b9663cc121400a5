"""ctypes view of the C ABI in ``include/lcg_hip.h`` (liblcg_amd/lib/liblcg_hip.so).

No compute happens in Python: every function here forwards to the HIP library.  If the
shared object is missing the import fails loudly -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "lib", "liblcg_hip.so")
# A/B runs of scripts/ only: LCG_HIP_LAB=1 loads the LAB build (`make -C liblcg_amd/csrc LAB=1`: the closed experiments' knobs compiled in)
if os.environ.get("LCG_HIP_LAB") == "1" and os.path.exists(os.path.join(HERE, "lib", "lab", "liblcg_hip.so")):
    SO_PATH = os.path.join(HERE, "lib", "lab", "liblcg_hip.so")
CSRC = os.path.join(HERE, "csrc")

c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)
vp = C.c_void_p


class LcgPara(C.Structure):       # lcg_para, util.h:95-148
    _fields_ = [("max_iterations", C.c_int), ("epsilon", C.c_double), ("abs_diff", C.c_int),
                ("restart_epsilon", C.c_double), ("step", C.c_double), ("sigma", C.c_double),
                ("beta", C.c_double), ("maxi_m", C.c_int)]


class ClcgPara(C.Structure):      # clcg_para, util.h:247-273
    _fields_ = [("max_iterations", C.c_int), ("epsilon", C.c_double), ("abs_diff", C.c_int)]


# callback prototypes (lcg.h:37-38,53-54; clcg.h:40-41,56-57)
AXFUNC = C.CFUNCTYPE(None, vp, vp, vp, C.c_int)
PROGRESS = C.CFUNCTYPE(C.c_int, vp, vp, C.c_double, C.POINTER(LcgPara), C.c_int, C.c_int)
CAXFUNC = C.CFUNCTYPE(None, vp, vp, vp, C.c_int, C.c_int, C.c_int)
CPROGRESS = C.CFUNCTYPE(C.c_int, vp, vp, C.c_double, C.POINTER(ClcgPara), C.c_int, C.c_int)

# every symbol include/lcg_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "lcg_hip_init": (C.c_int, [C.c_int]),
    "lcg_hip_set_stream": (C.c_int, [vp]),
    "lcg_hip_get_stream": (vp, []),
    "lcg_hip_synchronize": (C.c_int, []),
    "lcg_hip_memcpy": (C.c_int, [vp, vp, C.c_uint64, C.c_int]),
    "lcg_hip_last_error": (C.c_char_p, []),
    "lcg_hip_default_parameters": (LcgPara, []),
    "clcg_hip_default_parameters": (ClcgPara, []),
    "lcg_hip_last_iterations": (C.c_int, []),
    "lcg_hip_last_residual": (C.c_double, []),
    "lcg_hip_set_profiling": (C.c_int, [C.c_int]),
    "lcg_hip_set_cg_schedule": (C.c_int, [C.c_int]),
    "lcg_hip_set_placement": (C.c_int, [C.c_int]),
    "lcg_hip_last_placement": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "lcg_hip_pool_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "lcg_hip_pool_add_arena_for_test": (C.c_int, [C.c_uint64, C.c_int]),
    "lcg_hip_placement_tune_for_test": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_uint64, C.c_int, C.c_int]),
    "lcg_hip_last_placement_walk": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "lcg_hip_last_ax_mean_us": (C.c_double, []),
    "lcg_hip_last_ax_calls": (C.c_int, []),
    "lcg_hip_last_launches": (C.c_int, [c_int_p, c_int_p, c_int_p, c_int_p]),
    "lcg_hip_last_finisher_steps": (C.c_int, []),
    "lcg_hip_trim": (C.c_int, []),
    "lcg_hip_solver": (C.c_int, [vp, vp, vp, vp, C.c_int, C.POINTER(LcgPara), vp, C.c_int, C.c_int]),
    "lcg_hip_solver_preconditioned": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.POINTER(LcgPara), vp, C.c_int, C.c_int]),
    "lcg_hip_solver_constrained": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(LcgPara), vp, C.c_int, C.c_int]),
    "lcg_hip_set2box": (C.c_int, [C.c_int, vp, vp, vp]),
    "lcg_hip_lcg": (C.c_int, [vp, vp, vp, vp, C.c_int, C.POINTER(LcgPara), vp, vp, vp, vp, C.c_int]),
    "lcg_hip_lcgs": (C.c_int, [vp, vp, vp, vp, C.c_int, C.POINTER(LcgPara), vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]),
    "clcg_hip_solver": (C.c_int, [vp, vp, vp, vp, C.c_int, C.POINTER(ClcgPara), vp, C.c_int, C.c_int]),
    "clcg_hip_solver_preconditioned": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.POINTER(ClcgPara), vp, C.c_int, C.c_int]),
    "clcg_hip_jacobi_mx": (None, [vp, vp, vp, C.c_int, C.c_int, C.c_int]),
    "lcg_hip_set_shadow_seed": (C.c_int, [C.c_uint]),
    "lcg_hip_set_shadow_vector": (C.c_int, [vp, C.c_int]),
    "lcg_hip_csr_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int64, vp, vp, vp, C.c_int, C.c_int, C.c_int]),
    "lcg_hip_csr_from_coo": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int64, vp, vp, vp, C.c_int, C.c_int]),
    "lcg_hip_csr_destroy": (C.c_int, [vp]),
    "lcg_hip_csr_rows": (C.c_int, [vp]),
    "lcg_hip_csr_nnz": (C.c_int64, [vp]),
    "lcg_hip_csr_arrays": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
    "lcg_hip_csr_set_kernel": (C.c_int, [vp, C.c_int]),
    "lcg_hip_csr_set_packed": (C.c_int, [vp, C.c_int]),
    "lcg_hip_csr_packed_runs": (C.c_int64, [vp, C.POINTER(C.c_int64)]),
    "lcg_hip_csr_packed_templates": (C.c_int64, [vp]),
    "lcg_hip_csr_set_binned": (C.c_int, [vp, C.c_int]),
    "lcg_hip_csr_set_tiled": (C.c_int, [vp, C.c_int]),
    "lcg_hip_csr_set_ranges": (C.c_int, [vp, C.c_int]),
    "lcg_hip_csr_ranges": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
    "lcg_hip_csr_tiled_status": (C.c_char_p, [vp]),
    "lcg_hip_csr_binned_status": (C.c_char_p, [vp]),
    "lcg_hip_csr_last_kernel": (C.c_char_p, [vp]),
    "lcg_hip_csr_last_traffic_model": (C.c_int64, [vp]),
    "lcg_hip_csr_plan_info": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "lcg_hip_csr_build_jacobi": (C.c_int, [vp, vp]),
    "lcg_hip_csr_ax": (None, [vp, vp, vp, C.c_int]),
    "lcg_hip_jacobi_mx": (None, [vp, vp, vp, C.c_int]),
    "clcg_hip_csr_ax": (None, [vp, vp, vp, C.c_int, C.c_int, C.c_int]),
    "lcg_hip_spmv": (C.c_int, [vp, vp, vp]),
    "lcg_hip_spmv_op": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
    "lcg_hip_spmv_dot": (C.c_int, [vp, vp, vp, vp, c_double_p]),
    "lcg_hip_dot": (C.c_int, [C.c_int, vp, vp, c_double_p]),
    "lcg_hip_nrm2": (C.c_int, [C.c_int, vp, c_double_p]),
    "lcg_hip_axpy": (C.c_int, [C.c_int, C.c_double, vp, vp]),
    "lcg_hip_scal": (C.c_int, [C.c_int, C.c_double, vp]),
    "lcg_hip_vecmul": (C.c_int, [C.c_int, vp, vp, vp]),
    "lcg_hip_vecdiv": (C.c_int, [C.c_int, vp, vp, vp]),
    "clcg_hip_dot": (C.c_int, [C.c_int, vp, vp, c_double_p]),
    "clcg_hip_inner": (C.c_int, [C.c_int, vp, vp, c_double_p]),
    "clcg_hip_axpy": (C.c_int, [C.c_int, c_double_p, vp, vp]),
    "clcg_hip_vecdiv": (C.c_int, [C.c_int, vp, vp, vp]),
    "lcg_hip_csr_generate": (C.c_int, [C.POINTER(vp), C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_uint64, C.c_double, C.c_int64, C.c_int64]),
    "lcg_hip_csr_generate_ex": (C.c_int, [C.POINTER(vp), C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_uint64, C.c_double, C.c_int64, C.c_int64]),
    "lcg_hip_gen_xtrue": (C.c_int, [C.c_int64, C.c_uint64, C.c_int64, C.c_int64, vp]),
    "lcg_hip_csr_laplace2d": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "lcg_hip_comm_unique_id": (C.c_int, [vp]),
    "lcg_hip_comm_init": (C.c_int, [C.c_int, C.c_int, vp]),
    "lcg_hip_comm_destroy": (C.c_int, []),
    "lcg_hip_comm_rank": (C.c_int, []),
    "lcg_hip_comm_size": (C.c_int, []),
    "lcg_hip_comm_library": (C.c_char_p, []),
    "lcg_hip_csr_distribute": (C.c_int, [vp, C.c_int64, C.c_int]),
    "lcg_hip_allreduce_sum": (C.c_int, [vp, C.c_int]),
    "lcg_hip_barrier": (C.c_int, []),
    "lcg_hip_csr_direct_selfloop_for_test": (C.c_int, [vp, C.c_int, C.c_int]),
    "lcg_hip_p2p_export": (C.c_int, [vp]),
    "lcg_hip_p2p_connect": (C.c_int, [C.c_int, C.c_int, vp]),
    "lcg_hip_p2p_selftest": (C.c_int, [C.c_int]),
    "lcg_hip_p2p_enable": (C.c_int, [C.c_int]),
    "lcg_hip_p2p_status": (C.c_int, []),
    "lcg_hip_p2p_set_timeout_ms": (C.c_int, [C.c_int]),
    "lcg_hip_p2p_disconnect": (C.c_int, []),
    "lcg_hip_csr_split_for_test": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int]),
    "lcg_hip_csr_xfull": (vp, [vp]),
    "lcg_hip_csr_ax_part_for_probe": (C.c_int, [vp, vp, vp, C.c_int]),
    "lcg_hip_csr_local_nnz": (C.c_int64, [vp]),
    "lcg_hip_csr_exchange_volume": (C.c_int64, [vp]),
    "lcg_hip_csr_need_ranges_for_test": (C.c_int, [vp, C.c_int, vp]),
}

_lib = None


def build(verbose: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return SO_PATH


def load():
    """Load liblcg_hip.so and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        # not a fallback: the same HIP library, compiled now if the toolchain is at hand
        try:
            build()
        except Exception as exc:
            raise ImportError(f"{SO_PATH} is not built and could not be built here ({exc}): run "
                              "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)")
    # torch ships its own libamdhip64/librccl; importing it first makes this library bind to
    # the SAME runtime objects instead of mapping a second HIP runtime into the process.
    import torch  # noqa: F401
    lib = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError = header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def fnptr(lib, name: str) -> vp:
    """Address of an exported C function, usable as a callback argument."""
    return C.cast(getattr(lib, name), vp)
