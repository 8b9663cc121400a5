/*
 * lcg_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99) of the iteration hot path of liblcg's
 * native/OpenMP back-end: CG, PCG, CGS, BiCGStab (real, fp64) and
 * BiCG-symmetric, CGS, BiCGStab, TFQMR (complex, c128), plus the CSR / COO
 * matrix-vector products and the Jacobi apply that the callbacks perform.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped path (liblcg_amd/) never links, imports or
 * calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py runs this restatement
 * against the real reference compiled from /root/reference (oracle/_ref,
 * recipe in oracle/Makefile) and demands bit-identical iterates; the outputs
 * of that reference are committed as tests/golden/ *.npz so the pin survives
 * on machines where /root/reference does not exist.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src/lib).
 */
#ifndef LCG_ORACLE_H
#define LCG_ORACLE_H

#include <complex.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* util.h:95-148 -- same field order and types, so one struct serves both the
 * oracle and the reference shim. */
typedef struct orc_para {
    int    max_iterations;
    double epsilon;
    int    abs_diff;
    double restart_epsilon;
    double step;
    double sigma;
    double beta;
    int    maxi_m;
} orc_para;

/* util.h:247-273 */
typedef struct orc_cpara {
    int    max_iterations;
    double epsilon;
    int    abs_diff;
} orc_cpara;

/* util.h:69-90 (real) and :226-242 (complex). */
enum {
    ORC_CONVERGENCE = 0,
    ORC_STOP = 1,
    ORC_ALREADY_OPTIMIZIED = 2,
    ORC_UNKNOWN_ERROR = -1024,
    ORC_INVILAD_VARIABLE_SIZE = -1023,
    ORC_INVILAD_MAX_ITERATIONS = -1022,
    ORC_INVILAD_EPSILON = -1021,
    ORC_INVILAD_RESTART_EPSILON = -1020,
    ORC_REACHED_MAX_ITERATIONS = -1019,
    ORC_NULL_PRECONDITION_MATRIX = -1018,
    ORC_NAN_VALUE = -1017,
    ORC_INVALID_POINTER = -1016,
    /* complex enum has no RESTART_EPSILON entry, so its tail is shifted */
    ORC_C_REACHED_MAX_ITERATIONS = -1020,
    ORC_C_NAN_VALUE = -1019,
    ORC_C_INVALID_POINTER = -1018
};

/* lcg.h:37-38 / lcg.h:53-54 */
typedef void (*orc_axfunc)(void *instance, const double *x, double *Ax, int n);
typedef int (*orc_progress)(void *instance, const double *m, double converge,
                            const orc_para *param, int n, int k);
/* clcg.h:40-41 (layout/conjugate are always MatNormal/NonConjugate on the
 * solvers restated here, clcg.cpp:463,474,620,630,707,759,771) / clcg.h:56-57 */
typedef void (*orc_caxfunc)(void *instance, const double _Complex *x,
                            double _Complex *Ax, int n, int layout, int conjugate);
typedef int (*orc_cprogress)(void *instance, const double _Complex *m, double converge,
                             const orc_cpara *param, int n, int k);

/* ---- real solvers: lcg.cpp ---- */
int orc_lcg(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, int n,
            const orc_para *param, void *instance);                 /* lcg.cpp:143-274 */
int orc_lpcg(orc_axfunc Afp, orc_axfunc Mfp, orc_progress Pfp, double *m, const double *B,
             int n, const orc_para *param, void *instance);         /* lcg.cpp:293-434 */
int orc_lcgs(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, int n,
             const orc_para *param, void *instance);                /* lcg.cpp:437-612 */
int orc_lbicgstab(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, int n,
                  const orc_para *param, void *instance);           /* lcg.cpp:629-794 */
int orc_lbicgstab2(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, int n,
                   const orc_para *param, void *instance);          /* lcg.cpp:812-1034 */
int orc_lpg(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, const double *low,
            const double *hig, int n, const orc_para *param, void *instance);    /* lcg.cpp:1054-1204 */
int orc_lspg(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, const double *low,
             const double *hig, int n, const orc_para *param, void *instance);   /* lcg.cpp:1224-1446 */
int orc_lcg_solver_constrained(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B,
                               const double *low, const double *hig, int n, const orc_para *param,
                               void *instance, int solver_id);      /* lcg.cpp:121-140 */
/* lcg.cpp:59-91 dispatch: solver_id follows lcg_solver_enum (util.h:32-64);
 * ids other than CG/CGS/BICGSTAB fall through to CGS as the reference does. */
int orc_lcg_solver(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, int n,
                   const orc_para *param, void *instance, int solver_id);

/* ---- complex solvers: clcg.cpp.  rbar0 = the shadow residual the reference
 * draws from srand(time(0)) (clcg.cpp:399-403,556-560,721-725); here it is an
 * input so runs are reproducible.  orc_clcg_vecrnd restates the draw. ---- */
int orc_clbicg(orc_caxfunc Afp, orc_cprogress Pfp, double _Complex *m, const double _Complex *B, int n,
               const orc_cpara *param, void *instance);             /* clcg.cpp:77-226 */
/* PARITY UNPINNED (CUDA/Eigen-only in the reference): see clcg_oracle.c */
int orc_clpcg(orc_caxfunc Afp, orc_caxfunc Mfp, orc_cprogress Pfp, double _Complex *m,
              const double _Complex *B, int n, const orc_cpara *param, void *instance);  /* clcg_cuda.cu:403-558 */
int orc_clpbicg(orc_caxfunc Afp, orc_caxfunc Mfp, orc_cprogress Pfp, double _Complex *m,
                const double _Complex *B, int n, const orc_cpara *param, void *instance);  /* clcg_eigen.cpp:685-802 (UNPINNED) */
int orc_clbicg_symmetric(orc_caxfunc Afp, orc_cprogress Pfp, double _Complex *m,
                         const double _Complex *B, int n, const orc_cpara *param,
                         void *instance);                           /* clcg.cpp:228-364 */
int orc_clcgs(orc_caxfunc Afp, orc_cprogress Pfp, double _Complex *m,
              const double _Complex *B, int n, const orc_cpara *param, void *instance,
              const double _Complex *rbar0);                        /* clcg.cpp:366-522 */
int orc_clbicgstab(orc_caxfunc Afp, orc_cprogress Pfp, double _Complex *m,
                   const double _Complex *B, int n, const orc_cpara *param, void *instance,
                   const double _Complex *rbar0);                   /* clcg.cpp:524-679 */
int orc_cltfqmr(orc_caxfunc Afp, orc_cprogress Pfp, double _Complex *m,
                const double _Complex *B, int n, const orc_cpara *param, void *instance,
                const double _Complex *rbar0);                      /* clcg.cpp:681-881 */
void orc_clcg_vecrnd(double _Complex *a, double lre, double lim, double hre, double him,
                     int n, unsigned seed);                         /* lcg_complex.cpp:118-127 */

/* ---- vector / matrix primitives ---- */
double orc_dot(const double *a, const double *b, int n);            /* algebra.cpp:154-163 */
double _Complex orc_cdot(const double _Complex *a, const double _Complex *b, int n);   /* lcg_complex.cpp:143-154 */
double _Complex orc_cinner(const double _Complex *a, const double _Complex *b, int n); /* lcg_complex.cpp:156-167 */
void orc_coo_matvec(const int *row, const int *col, const double *val, const double *x,
                    double *y, int n, int nnz);                     /* algebra.cpp:195-221 */
void orc_coo_cmatvec(const int *row, const int *col, const double _Complex *val,
                     const double _Complex *x, double _Complex *y, int n, int nnz);
void orc_csr_matvec(const int *rowptr, const int *col, const double *val, const double *x,
                    double *y, int n, int threads);                 /* sample8.cu:96-103 semantics */
void orc_csr_cmatvec(const int *rowptr, const int *col, const double _Complex *val,
                     const double _Complex *x, double _Complex *y, int n, int threads);
/* y = op(A).x, op = transpose and/or conjugate (lcg_matrix_e / clcg_complex_e, algebra.h:31-50);
 * the transposed product scatters in storage order, like clcg_matvec's MatTranspose branch
 * (lcg_complex.cpp:169-234) does over its dense rows */
void orc_csr_cmatvec_op(const int *rowptr, const int *col, const double _Complex *val,
                        const double _Complex *x, double _Complex *y, int n, int transpose, int conjugate);
int orc_coo_to_csr(const int *row, const int *col, int n, int nnz, int *rowptr, int *perm);
void orc_csr_diag(const int *rowptr, const int *col, const double *val, int n, double *diag); /* algebra_cuda.cu:40-57 */
void orc_csr_cdiag(const int *rowptr, const int *col, const double _Complex *val, int n,
                   double _Complex *diag);                          /* lcg_complex_cuda.cu:46-63 */

/* ---- ready-made callback instances (CSR A.x and Jacobi) ---- */
typedef struct orc_csr {
    int n;
    const int *rowptr;
    const int *col;
    const double *val;        /* real values, or interleaved re/im when complex */
    const double *invdiag;    /* reciprocal diagonal for the Jacobi callback (sample1.cpp:55-62,98-107) */
    int threads;              /* OpenMP threads for the SpMV callback; <=1 = serial */
    int iters;                /* filled by orc_record_progress */
    double last_residual;
    int n_ax;                 /* number of A.x calls seen */
} orc_csr;
void orc_csr_ax(void *instance, const double *x, double *Ax, int n);
void orc_jacobi_mx(void *instance, const double *x, double *Mx, int n);
int  orc_record_progress(void *instance, const double *m, double converge,
                         const orc_para *param, int n, int k);
void orc_csr_cax(void *instance, const double _Complex *x, double _Complex *Ax, int n,
                 int layout, int conjugate);
int  orc_record_cprogress(void *instance, const double _Complex *m, double converge,
                          const orc_cpara *param, int n, int k);

/* one-call drivers used from Python (ctypes cannot conveniently pass C callbacks) */
int orc_solve_csr(int solver_id, int jacobi, orc_csr *A, double *m, const double *B,
                  const orc_para *param);
int orc_csolve_csr_pcg(orc_csr *A, double *m_interleaved, const double *B_interleaved,
                       const orc_cpara *param);
int orc_csolve_csr_pbicg(orc_csr *A, double *m_interleaved, const double *B_interleaved,
                       const orc_cpara *param);    /* orc_clpbicg with the same complex Jacobi */     /* Jacobi: invdiag holds interleaved complex reciprocals */
int orc_solve_csr_box(int solver_id, orc_csr *A, double *m, const double *B, const double *low,
                      const double *hig, const orc_para *param);
int orc_csolve_csr(int solver_id, orc_csr *A, double *m_interleaved, const double *B_interleaved,
                   const orc_cpara *param, const double *rbar0_interleaved);

/* ---- synthetic matrix family shared with the HIP generator (not in the
 * reference: SURVEY.md section 8d recommends it; defined in DESIGN.md) ---- */
typedef struct orc_gen {
    int64_t n;          /* global rows */
    int     npairs;     /* affine maps (<=16): each gives a column and its mirror */
    int64_t a[16];      /* multiplier, gcd(a,n)=1 (scrambled) or 1 (banded) */
    int64_t ainv[16];   /* a^-1 mod n */
    int64_t c[16];      /* offset */
    int     banded;     /* pattern.  0: col = (a*i+c) mod n and inverse ("scrambled");  1: col = i +/- c, the SAME
                         * offsets for every row (npairs constant diagonals each side, clipped, no wrap);
                         * 2: row-random band -- rows are cut into blocks of 2^wb_log2, map k sends block b onto
                         * block b+1 by a keyed bijection of the in-block position, so every row draws its own
                         * columns within +-2^(wb_log2+1) (c[k] = key of map k) */
    int     symmetric;  /* 1: value depends on {min,max}; 0: on the ordered pair */
    uint64_t seed;
    double  diag_shift; /* A_ii = sum|A_ij| + diag_shift */
    int     wb_log2;    /* pattern 2 only: log2 of the block size */
} orc_gen;
void orc_gen_init(orc_gen *g, int64_t n, int npairs, int64_t band, int symmetric,
                  uint64_t seed, double diag_shift);      /* band > 0: pattern 1, else pattern 0 */
/* pattern as above; band = W: pattern 1 draws offsets in [1,W]; pattern 2 uses blocks of the largest
 * power of two <= W/2, so that every column of row i lies in (i-W, i+W) */
void orc_gen_init_ex(orc_gen *g, int64_t n, int npairs, int pattern, int64_t band, int symmetric,
                     uint64_t seed, double diag_shift);
/* counts[i-r0] = entries of global row i (diagonal included) for i in [r0,r1) */
void orc_gen_count(const orc_gen *g, int64_t r0, int64_t r1, int *counts);
/* fills col/val of rows [r0,r1) given the local rowptr (rowptr[0]=0) */
void orc_gen_fill(const orc_gen *g, int64_t r0, int64_t r1, const int *rowptr, int *col,
                  double *val);
void orc_gen_xtrue(const orc_gen *g, int64_t r0, int64_t r1, double *x);

#ifdef __cplusplus
}
#endif
#endif
