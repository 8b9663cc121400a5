/*
 * ref_shim.cpp -- TEST INFRASTRUCTURE.  extern "C" doorway into the REAL
 * liblcg native back-end, compiled from the sources where they lie under
 * /root/reference (recipe: oracle/Makefile, target `ref`; output only into
 * oracle/_ref/).  Nothing here restates the reference: it includes the
 * reference's own headers and calls its own lcg_solver()/clcg_solver()
 * (lcg.h:71-72,90-91; clcg.h:74-76) with CSR callbacks supplied by the
 * oracle (liblcg ships no CSR SpMV of its own).
 *
 * Used to (1) pin the C restatement in oracle/ bit-for-bit, (2) generate the
 * golden vectors under tests/golden/, (3) serve as bench.py's
 * cpu_baseline.kind == "reference".
 */
#include "lcg.h"
#include "clcg.h"

#include <ctime>

#include "lcg_oracle.h"

namespace {

/* wall-clock second seen at the first and second A.x of a complex solve: the
 * reference draws rbar0 with srand(time(0)) between those two calls
 * (clcg.cpp:391-404, 548-561, 707-725) */
long long g_sec_first = 0, g_sec_second = 0;

void ref_ax(void *instance, const lcg_float *x, lcg_float *Ax, const int n)
{
    orc_csr_ax(instance, x, Ax, n);
}

void ref_mx(void *instance, const lcg_float *x, lcg_float *Mx, const int n)
{
    orc_jacobi_mx(instance, x, Mx, n);
}

int ref_progress(void *instance, const lcg_float *m, const lcg_float converge,
                 const lcg_para *param, const int n, const int k)
{
    (void)m; (void)param; (void)n;
    orc_csr *A = static_cast<orc_csr *>(instance);
    A->iters = k;
    A->last_residual = converge;
    return 0;
}

void ref_cax(void *instance, const lcg_complex *x, lcg_complex *Ax, const int n,
             lcg_matrix_e layout, clcg_complex_e conjugate)
{
    orc_csr *A = static_cast<orc_csr *>(instance);
    if (layout != MatNormal || conjugate != NonConjugate) {     // A^T / A^H / conj(A): the oracle's scatter product
        A->n_ax++;
        orc_csr_cmatvec_op(A->rowptr, A->col, reinterpret_cast<const double _Complex *>(A->val),
                           reinterpret_cast<const double _Complex *>(x), reinterpret_cast<double _Complex *>(Ax), n,
                           layout == MatTranspose, conjugate == Conjugate);
        return;
    }
    A->n_ax++;
    if (A->n_ax == 1) g_sec_first = (long long)time(0);
    if (A->n_ax == 2) g_sec_second = (long long)time(0);
    /* std::complex<double> and double _Complex share one layout */
    const double *xv = reinterpret_cast<const double *>(x);
    double *yv = reinterpret_cast<double *>(Ax);
    const int *rp = A->rowptr, *ci = A->col;
    const double *av = A->val;
    for (int i = 0; i < n; i++) {
        lcg_complex s(0.0, 0.0);
        for (int k = rp[i]; k < rp[i + 1]; k++)
            s += lcg_complex(av[2 * k], av[2 * k + 1]) * lcg_complex(xv[2 * ci[k]], xv[2 * ci[k] + 1]);
        yv[2 * i] = s.real();
        yv[2 * i + 1] = s.imag();
    }
}

int ref_cprogress(void *instance, const lcg_complex *m, const lcg_float converge,
                  const clcg_para *param, const int n, const int k)
{
    (void)m; (void)param; (void)n;
    orc_csr *A = static_cast<orc_csr *>(instance);
    A->iters = k;
    A->last_residual = converge;
    return 0;
}

} // namespace

extern "C" {

/* sizes of the reference's parameter structs, so Python can check its mirror */
int ref_sizeof_lcg_para(void) { return (int)sizeof(lcg_para); }
int ref_sizeof_clcg_para(void) { return (int)sizeof(clcg_para); }

int ref_solve_csr(int solver_id, int jacobi, orc_csr *A, double *m, const double *B,
                  const orc_para *param)
{
    A->iters = 0; A->last_residual = 0.0; A->n_ax = 0;
    const lcg_para *p = reinterpret_cast<const lcg_para *>(param);
    if (jacobi)
        return lcg_solver_preconditioned(ref_ax, ref_mx, ref_progress, m, B, A->n, p, A, LCG_PCG);
    return lcg_solver(ref_ax, ref_progress, m, B, A->n, p, A, (lcg_solver_enum)solver_id);
}

int ref_solve_csr_box(int solver_id, orc_csr *A, double *m, const double *B, const double *low,
                      const double *hig, const orc_para *param)
{
    A->iters = 0; A->last_residual = 0.0; A->n_ax = 0;
    return lcg_solver_constrained(ref_ax, ref_progress, m, B, low, hig, A->n,
                                  reinterpret_cast<const lcg_para *>(param), A, (lcg_solver_enum)solver_id);
}

/* The reference seeds rbar0 from time(0) inside the call.  seed_before/after
 * bracket the draw so the caller can replay the very same vector through
 * orc_clcg_vecrnd when both agree (i.e. the second did not tick in between). */
int ref_csolve_csr(int solver_id, orc_csr *A, double *m, const double *B,
                   const orc_cpara *param, long long *seed_before, long long *seed_after)
{
    A->iters = 0; A->last_residual = 0.0; A->n_ax = 0;
    const clcg_para *p = reinterpret_cast<const clcg_para *>(param);
    g_sec_first = g_sec_second = 0;
    int ret = clcg_solver(ref_cax, ref_cprogress, reinterpret_cast<lcg_complex *>(m),
                          reinterpret_cast<const lcg_complex *>(B), A->n, p, A,
                          (clcg_solver_enum)solver_id);
    if (seed_before) *seed_before = g_sec_first;
    if (seed_after) *seed_after = g_sec_second ? g_sec_second : (long long)time(0);
    return ret;
}

double ref_dot(const double *a, const double *b, int n)
{
    lcg_float r;
    lcg_dot(r, a, b, n);
    return r;
}

void ref_coo_matvec(const int *row, const int *col, const double *val, const double *x,
                    double *y, int n, int nnz)
{
    lcg_matvec_coo(row, col, val, x, y, n, n, nnz, false);
}

} // extern "C"
