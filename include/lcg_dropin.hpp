// lcg_dropin.hpp -- liblcg's C++ entry points, served by liblcg_hip.so.
//
// A program written against liblcg's lcg.h / clcg.h (native back-end) keeps its calls:
//
//     lcg_solver(Afp, Pfp, m, B, n, &para, instance, LCG_CG);          // lcg.h:71-72
//     lcg_solver_preconditioned(Afp, Mfp, Pfp, m, B, n, &para, inst);  // lcg.h:90-91
//     lcg(Afp, Pfp, m, B, n, &para, inst, Gk, Dk, ADk);                // lcg.h:135-137
//     lcgs(Afp, Pfp, m, B, n, &para, inst, RK, R0T, PK, AX, UK, QK, WK); // lcg.h:166-169
//     clcg_solver(Afp, Pfp, m, B, n, &para, inst, CLCG_TFQMR);         // clcg.h:74-76
//
// with the same argument order, defaults, parameter structs, enums and return codes.  What
// changes is WHERE the iteration runs: m and B are host arrays (copied in, m copied back -- the
// contract of the reference's own GPU entry lcg_solver_cuda, lcg_cuda.cu:103-111,210), the loop
// is device resident, and the callbacks are handed DEVICE pointers and must enqueue their work
// on lcg_hip_get_stream().  The ready-made lcg_hip_csr_ax / lcg_hip_jacobi_mx callbacks
// (instance = lcg_hip_csr_t) make the common CSR case a one-liner.  Device-resident m/B:
// use the *_device variants below (or the C ABI's `mem` argument directly).
//
// Names follow the reference (util.h, lcg.h, clcg.h); this header is only declarations and
// inline forwarding -- the implementation is the C ABI in lcg_hip.h.
#ifndef LCG_DROPIN_HPP
#define LCG_DROPIN_HPP

#include <complex>
#include <stdexcept>
#include <string>

#include "lcg_hip.h"

typedef double lcg_float;                       // algebra.h:50
typedef std::complex<lcg_float> lcg_complex;    // lcg_complex.h:33 (LibLCG_STD_COMPLEX)

enum lcg_matrix_e { MatNormal, MatTranspose };          // algebra.h:31-35
enum clcg_complex_e { NonConjugate, Conjugate };        // algebra.h:40-44
// lcg_solver_enum / clcg_solver_enum (util.h:32-64, 187-221) are the enum types of lcg_hip.h

// clcg.h:40-41 / 56-57 with the reference's C++ types
typedef void (*clcg_axfunc_ptr)(void *instance, const lcg_complex *x, lcg_complex *prod_Ax,
                                const int x_size, lcg_matrix_e layout, clcg_complex_e conjugate);
typedef int (*clcg_progress_ptr)(void *instance, const lcg_complex *m, const lcg_float converge,
                                 const clcg_para *param, const int n_size, const int k);

inline lcg_para lcg_default_parameters() { return lcg_hip_default_parameters(); }       // util.h:163
inline clcg_para clcg_default_parameters() { return clcg_hip_default_parameters(); }    // util.h:287

// util.cpp:39-51
inline lcg_solver_enum lcg_select_solver(const std::string &s)
{
    if (s == "LCG_CG") return LCG_CG;
    if (s == "LCG_PCG") return LCG_PCG;
    if (s == "LCG_CGS") return LCG_CGS;
    if (s == "LCG_BICGSTAB") return LCG_BICGSTAB;
    if (s == "LCG_BICGSTAB2") return LCG_BICGSTAB2;
    if (s == "LCG_PG") return LCG_PG;
    if (s == "LCG_SPG") return LCG_SPG;
    throw std::invalid_argument("Invalid solver type.");
}

// ---- real ------------------------------------------------------------------------------------
inline int lcg_solver(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, lcg_float *m, const lcg_float *B,
                      const int n_size, const lcg_para *param, void *instance,
                      lcg_solver_enum solver_id = LCG_CGS)
{
    return lcg_hip_solver(Afp, Pfp, m, B, n_size, param, instance, solver_id, LCG_HIP_MEM_HOST);
}
inline int lcg_solver_device(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, lcg_float *d_m, const lcg_float *d_B,
                             const int n_size, const lcg_para *param, void *instance,
                             lcg_solver_enum solver_id = LCG_CGS)
{
    return lcg_hip_solver(Afp, Pfp, d_m, d_B, n_size, param, instance, solver_id, LCG_HIP_MEM_DEVICE);
}

inline int lcg_solver_preconditioned(lcg_axfunc_ptr Afp, lcg_axfunc_ptr Mfp, lcg_progress_ptr Pfp,
                                     lcg_float *m, const lcg_float *B, const int n_size,
                                     const lcg_para *param, void *instance,
                                     lcg_solver_enum solver_id = LCG_PCG)
{
    return lcg_hip_solver_preconditioned(Afp, Mfp, Pfp, m, B, n_size, param, instance, solver_id,
                                         LCG_HIP_MEM_HOST);
}
inline int lcg_solver_preconditioned_device(lcg_axfunc_ptr Afp, lcg_axfunc_ptr Mfp, lcg_progress_ptr Pfp,
                                            lcg_float *d_m, const lcg_float *d_B, const int n_size,
                                            const lcg_para *param, void *instance,
                                            lcg_solver_enum solver_id = LCG_PCG)
{
    return lcg_hip_solver_preconditioned(Afp, Mfp, Pfp, d_m, d_B, n_size, param, instance, solver_id,
                                         LCG_HIP_MEM_DEVICE);
}

// lcg.h:111-113
inline int lcg_solver_constrained(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, lcg_float *m, const lcg_float *B,
                                  const lcg_float *low, const lcg_float *hig, const int n_size,
                                  const lcg_para *param, void *instance, lcg_solver_enum solver_id = LCG_PG)
{
    return lcg_hip_solver_constrained(Afp, Pfp, m, B, low, hig, n_size, param, instance, solver_id, LCG_HIP_MEM_HOST);
}

// The workspaces of lcg()/lcgs() exist to avoid per-call allocation (lcg.h:118-119): here they are
// DEVICE vectors of n_size doubles (or nullptr).
inline int lcg(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, lcg_float *m, const lcg_float *B, const int n_size,
               const lcg_para *param, void *instance, lcg_float *Gk = nullptr, lcg_float *Dk = nullptr,
               lcg_float *ADk = nullptr)
{
    return lcg_hip_lcg(Afp, Pfp, m, B, n_size, param, instance, Gk, Dk, ADk, LCG_HIP_MEM_HOST);
}
inline int lcgs(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, lcg_float *m, const lcg_float *B, const int n_size,
                const lcg_para *param, void *instance, lcg_float *RK = nullptr, lcg_float *R0T = nullptr,
                lcg_float *PK = nullptr, lcg_float *AX = nullptr, lcg_float *UK = nullptr,
                lcg_float *QK = nullptr, lcg_float *WK = nullptr)
{
    return lcg_hip_lcgs(Afp, Pfp, m, B, n_size, param, instance, RK, R0T, PK, AX, UK, QK, WK,
                        LCG_HIP_MEM_HOST);
}

// ---- complex -----------------------------------------------------------------------------------
// The reference's complex callbacks take std::complex pointers and two enums (clcg.h:40-41); the C ABI takes
// doubles and ints.  std::complex<double> is layout-compatible with double[2] (so the vectors pass through),
// but a function pointer is not cast across signatures: the caller's callbacks are reached through the
// trampolines below, which the C ABI calls with a small record as its instance.
namespace lcg_dropin_detail {
struct cthunk { clcg_axfunc_ptr Afp, Mfp; clcg_progress_ptr Pfp; void *instance; };
inline void c_ax(void *t, const double *x, double *y, const int n, int layout, int conjugate)
{
    const cthunk *c = static_cast<const cthunk *>(t);
    c->Afp(c->instance, reinterpret_cast<const lcg_complex *>(x), reinterpret_cast<lcg_complex *>(y), n,
           static_cast<lcg_matrix_e>(layout), static_cast<clcg_complex_e>(conjugate));
}
inline void c_mx(void *t, const double *x, double *y, const int n, int layout, int conjugate)
{
    const cthunk *c = static_cast<const cthunk *>(t);
    c->Mfp(c->instance, reinterpret_cast<const lcg_complex *>(x), reinterpret_cast<lcg_complex *>(y), n,
           static_cast<lcg_matrix_e>(layout), static_cast<clcg_complex_e>(conjugate));
}
inline int c_progress(void *t, const double *m, const double converge, const clcg_para *param, const int n, const int k)
{
    const cthunk *c = static_cast<const cthunk *>(t);
    return c->Pfp(c->instance, reinterpret_cast<const lcg_complex *>(m), converge, param, n, k);
}
inline int c_solve(clcg_axfunc_ptr Afp, clcg_progress_ptr Pfp, lcg_complex *m, const lcg_complex *B, const int n_size,
                   const clcg_para *param, void *instance, int solver_id, int mem)
{
    cthunk t = {Afp, nullptr, Pfp, instance};
    return clcg_hip_solver(Afp ? c_ax : nullptr, Pfp ? c_progress : nullptr, reinterpret_cast<double *>(m),
                           reinterpret_cast<const double *>(B), n_size, param, &t, solver_id, mem);
}
} // namespace lcg_dropin_detail

inline int clcg_solver(clcg_axfunc_ptr Afp, clcg_progress_ptr Pfp, lcg_complex *m, const lcg_complex *B,
                       const int n_size, const clcg_para *param, void *instance,
                       clcg_solver_enum solver_id = CLCG_BICG)
{
    return lcg_dropin_detail::c_solve(Afp, Pfp, m, B, n_size, param, instance, solver_id, LCG_HIP_MEM_HOST);
}
inline int clcg_solver_device(clcg_axfunc_ptr Afp, clcg_progress_ptr Pfp, lcg_complex *d_m, const lcg_complex *d_B,
                              const int n_size, const clcg_para *param, void *instance,
                              clcg_solver_enum solver_id = CLCG_BICG)
{
    return lcg_dropin_detail::c_solve(Afp, Pfp, d_m, d_B, n_size, param, instance, solver_id, LCG_HIP_MEM_DEVICE);
}
// clcg_solver_preconditioned_cuda (clcg_cuda.h:105-108) without the vendor handles
inline int clcg_solver_preconditioned(clcg_axfunc_ptr Afp, clcg_axfunc_ptr Mfp, clcg_progress_ptr Pfp, lcg_complex *m,
                                      const lcg_complex *B, const int n_size, const clcg_para *param, void *instance,
                                      clcg_solver_enum solver_id = CLCG_PCG)
{
    using namespace lcg_dropin_detail;
    cthunk t = {Afp, Mfp, Pfp, instance};
    return clcg_hip_solver_preconditioned(Afp ? c_ax : nullptr, Mfp ? c_mx : nullptr, Pfp ? c_progress : nullptr,
                                          reinterpret_cast<double *>(m), reinterpret_cast<const double *>(B), n_size, param,
                                          &t, solver_id, LCG_HIP_MEM_HOST);
}
// the ready-made complex CSR callback with the reference's C++ signature
inline void clcg_csr_ax(void *instance, const lcg_complex *x, lcg_complex *prod_Ax, const int n,
                        lcg_matrix_e layout, clcg_complex_e conjugate)
{
    clcg_hip_csr_ax(instance, reinterpret_cast<const double *>(x), reinterpret_cast<double *>(prod_Ax), n,
                    (int)layout, (int)conjugate);
}

// util.cpp:53-148 (plain text; the reference's terminal colouring is not reproduced)
inline const char *lcg_status_text(int code)
{
    switch (code) {
    case LCG_SUCCESS: return "Success! Iteration reached convergence.";
    case LCG_STOP: return "Iteration stopped by the progress evaluation function.";
    case LCG_ALREADY_OPTIMIZIED: return "The initial solution is already optimized.";
    case LCG_UNKNOWN_ERROR: return "Unknown error.";
    case LCG_INVILAD_VARIABLE_SIZE: return "Invalid variable size.";
    case LCG_INVILAD_MAX_ITERATIONS: return "Invalid maximal iteration times.";
    case LCG_INVILAD_EPSILON: return "Invalid value for epsilon.";
    case LCG_INVILAD_RESTART_EPSILON: return "Invalid value for restart epsilon.";
    case LCG_REACHED_MAX_ITERATIONS: return "Reached the maximal iteration times.";
    case LCG_NULL_PRECONDITION_MATRIX: return "Null precondition matrix.";
    case LCG_NAN_VALUE: return "NaN values found.";
    case LCG_INVALID_POINTER: return "Invalid pointer.";
    case LCG_SIZE_NOT_MATCH: return "Sizes of m and B do not match.";
    case LCG_HIP_E_RUNTIME: case LCG_HIP_E_NO_DEVICE: case LCG_HIP_E_COMM: case LCG_HIP_E_ARG:
        return lcg_hip_last_error();
    default: return "Unknown error.";
    }
}
inline void lcg_error_str(int er_index, bool er_throw = false)
{
    if (er_throw && er_index < 0) throw std::runtime_error(lcg_status_text(er_index));
}

#endif // LCG_DROPIN_HPP
