/*
 * csr_oracle.c -- TEST INFRASTRUCTURE (see lcg_oracle.h).
 *
 * Matrix side of the oracle: what the user callbacks of the reference do
 * (COO / CSR A.x, Jacobi M^-1.x, diagonal extraction, COO -> CSR), ready-made
 * callback instances, ctypes-friendly one-call drivers, and the CPU twin of
 * the synthetic matrix generator that the HIP library implements on device.
 */
#include "lcg_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double _Complex zc;

/* algebra.cpp:195-221, pre_position == false branch: zero, then scatter in
 * storage order. */
void orc_coo_matvec(const int *row, const int *col, const double *val, const double *x,
                    double *y, int n, int nnz)
{
    for (int i = 0; i < n; i++) y[i] = 0.0;
    for (int k = 0; k < nnz; k++) y[row[k]] += val[k] * x[col[k]];
}

void orc_coo_cmatvec(const int *row, const int *col, const zc *val, const zc *x, zc *y,
                     int n, int nnz)
{
    for (int i = 0; i < n; i++) y[i] = CMPLX(0.0, 0.0);
    for (int k = 0; k < nnz; k++) y[row[k]] += val[k] * x[col[k]];
}

/* y = A.x for CSR (int32 indices, base 0) -- the operation the reference
 * delegates to cusparseSpMV in its GPU samples (sample8.cu:96-103).  Each row
 * is accumulated left to right, so for a row-sorted COO file it reproduces
 * orc_coo_matvec bit for bit.  threads > 1 only splits rows (static). */
void orc_csr_matvec(const int *rowptr, const int *col, const double *val, const double *x,
                    double *y, int n, int threads)
{
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1)
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int k = rowptr[i]; k < rowptr[i + 1]; k++) s += val[k] * x[col[k]];
        y[i] = s;
    }
}

void orc_csr_cmatvec(const int *rowptr, const int *col, const zc *val, const zc *x, zc *y,
                     int n, int threads)
{
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1)
    for (int i = 0; i < n; i++) {
        zc s = CMPLX(0.0, 0.0);
        for (int k = rowptr[i]; k < rowptr[i + 1]; k++) s += val[k] * x[col[k]];
        y[i] = s;
    }
}

void orc_csr_cmatvec_op(const int *rowptr, const int *col, const zc *val, const zc *x, zc *y,
                        int n, int transpose, int conjugate)
{
    if (!transpose) {
        for (int i = 0; i < n; i++) {
            zc s = CMPLX(0.0, 0.0);
            for (int k = rowptr[i]; k < rowptr[i + 1]; k++) s += (conjugate ? conj(val[k]) : val[k]) * x[col[k]];
            y[i] = s;
        }
        return;
    }
    for (int i = 0; i < n; i++) y[i] = CMPLX(0.0, 0.0);
    for (int i = 0; i < n; i++)
        for (int k = rowptr[i]; k < rowptr[i + 1]; k++)
            y[col[k]] += (conjugate ? conj(val[k]) : val[k]) * x[i];
}

/* Stable counting sort of COO entries by row (what cusparseXcoo2csr assumes is
 * already true, sample8.cu:169).  perm[k] = source index of CSR slot k. */
int orc_coo_to_csr(const int *row, const int *col, int n, int nnz, int *rowptr, int *perm)
{
    (void)col;
    memset(rowptr, 0, sizeof(int) * (size_t)(n + 1));
    for (int k = 0; k < nnz; k++) {
        if (row[k] < 0 || row[k] >= n) return -1;
        rowptr[row[k] + 1]++;
    }
    for (int i = 0; i < n; i++) rowptr[i + 1] += rowptr[i];
    int *next = malloc(sizeof(int) * (size_t)n);
    memcpy(next, rowptr, sizeof(int) * (size_t)n);
    for (int k = 0; k < nnz; k++) perm[next[row[k]]++] = k;
    free(next);
    return 0;
}

/* algebra_cuda.cu:40-57: linear scan of each row for col == row. */
void orc_csr_diag(const int *rowptr, const int *col, const double *val, int n, double *diag)
{
    for (int i = 0; i < n; i++) {
        diag[i] = 0.0;
        for (int k = rowptr[i]; k < rowptr[i + 1]; k++)
            if (col[k] == i) { diag[i] = val[k]; break; }
    }
}

void orc_csr_cdiag(const int *rowptr, const int *col, const zc *val, int n, zc *diag)
{
    for (int i = 0; i < n; i++) {
        diag[i] = CMPLX(0.0, 0.0);
        for (int k = rowptr[i]; k < rowptr[i + 1]; k++)
            if (col[k] == i) { diag[i] = val[k]; break; }
    }
}

/* ------------------------------------------------------------ callbacks */
void orc_csr_ax(void *instance, const double *x, double *Ax, int n)
{
    orc_csr *A = instance;
    A->n_ax++;
    orc_csr_matvec(A->rowptr, A->col, A->val, x, Ax, n, A->threads);
}

/* sample1.cpp:55-62: z = p .* x with p = 1/diag (reciprocal form). */
void orc_jacobi_mx(void *instance, const double *x, double *Mx, int n)
{
    const orc_csr *A = instance;
    for (int i = 0; i < n; i++) Mx[i] = A->invdiag[i] * x[i];
}

int orc_record_progress(void *instance, const double *m, double converge,
                        const orc_para *param, int n, int k)
{
    (void)m; (void)param; (void)n;
    orc_csr *A = instance;
    A->iters = k;
    A->last_residual = converge;
    return 0;
}

void orc_csr_cax(void *instance, const zc *x, zc *Ax, int n, int layout, int conjugate)
{
    orc_csr *A = instance;
    A->n_ax++;
    if (layout || conjugate) orc_csr_cmatvec_op(A->rowptr, A->col, (const zc *)A->val, x, Ax, n, layout, conjugate);
    else orc_csr_cmatvec(A->rowptr, A->col, (const zc *)A->val, x, Ax, n, A->threads);
}

int orc_record_cprogress(void *instance, const zc *m, double converge,
                         const orc_cpara *param, int n, int k)
{
    (void)m; (void)param; (void)n;
    orc_csr *A = instance;
    A->iters = k;
    A->last_residual = converge;
    return 0;
}

/* solver_id: lcg_solver_enum (util.h:32-64).  jacobi != 0 routes through
 * lcg_solver_preconditioned, which always runs lpcg (lcg.cpp:87-91). */
int orc_solve_csr(int solver_id, int jacobi, orc_csr *A, double *m, const double *B,
                  const orc_para *param)
{
    A->iters = 0; A->last_residual = 0.0; A->n_ax = 0;
    if (jacobi)
        return orc_lpcg(orc_csr_ax, orc_jacobi_mx, orc_record_progress, m, B, A->n, param, A);
    return orc_lcg_solver(orc_csr_ax, orc_record_progress, m, B, A->n, param, A, solver_id);
}

/* complex Jacobi, z = x / diag as reciprocal multiply (sample10.cu:117 divides; DESIGN.md deviation 3) */
static void orc_cjacobi_mx(void *instance, const zc *x, zc *Mx, int n, int layout, int conjugate)
{
    (void)layout; (void)conjugate;
    const orc_csr *A = instance;
    const zc *inv = (const zc *)A->invdiag;
    for (int i = 0; i < n; i++) Mx[i] = inv[i] * x[i];
}

int orc_csolve_csr_pcg(orc_csr *A, double *m, const double *B, const orc_cpara *param)
{
    A->iters = 0; A->last_residual = 0.0; A->n_ax = 0;
    return orc_clpcg(orc_csr_cax, orc_cjacobi_mx, orc_record_cprogress, (zc *)m, (const zc *)B, A->n, param, A);
}

int orc_csolve_csr_pbicg(orc_csr *A, double *m, const double *B, const orc_cpara *param)
{
    A->iters = 0; A->last_residual = 0.0; A->n_ax = 0;
    return orc_clpbicg(orc_csr_cax, orc_cjacobi_mx, orc_record_cprogress, (zc *)m, (const zc *)B, A->n, param, A);
}

/* lcg_solver_constrained (lcg.h:111-113): LCG_PG = 5, LCG_SPG = 6 */
int orc_solve_csr_box(int solver_id, orc_csr *A, double *m, const double *B, const double *low,
                      const double *hig, const orc_para *param)
{
    A->iters = 0; A->last_residual = 0.0; A->n_ax = 0;
    return orc_lcg_solver_constrained(orc_csr_ax, orc_record_progress, m, B, low, hig, A->n, param, A, solver_id);
}

/* solver_id: clcg_solver_enum (util.h:187-221): 0 BICG, 1 BICG_SYM, 2 CGS, 3 BICGSTAB,
 * 4 TFQMR; other ids run CGS as clcg.cpp:68-70 does. */
int orc_csolve_csr(int solver_id, orc_csr *A, double *m, const double *B,
                   const orc_cpara *param, const double *rbar0)
{
    A->iters = 0; A->last_residual = 0.0; A->n_ax = 0;
    zc *zm = (zc *)m;
    const zc *zB = (const zc *)B, *zr = (const zc *)rbar0;
    switch (solver_id) {
    case 0: return orc_clbicg(orc_csr_cax, orc_record_cprogress, zm, zB, A->n, param, A);
    case 1: return orc_clbicg_symmetric(orc_csr_cax, orc_record_cprogress, zm, zB, A->n, param, A);
    case 3: return orc_clbicgstab(orc_csr_cax, orc_record_cprogress, zm, zB, A->n, param, A, zr);
    case 4: return orc_cltfqmr(orc_csr_cax, orc_record_cprogress, zm, zB, A->n, param, A, zr);
    default: return orc_clcgs(orc_csr_cax, orc_record_cprogress, zm, zB, A->n, param, A, zr);
    }
}

/* ------------------------------------------------ synthetic matrix family
 * Definition (DESIGN.md "Synthetic systems"): row i holds the diagonal plus,
 * for each of `npairs` maps k, the columns f_k(i) and f_k^-1(i):
 *   banded:    f_k(i) = i + c_k,  f_k^-1(i) = i - c_k   (dropped outside [0,n))
 *   scrambled: f_k(i) = (a_k*i + c_k) mod n, f_k^-1(i) = a_k^-1*(i - c_k) mod n
 * Duplicates and j == i are dropped; columns ascend.  Off-diagonal value
 * = -u(h), u in (0,1], h = mix(min(i,j), max(i,j), seed) when symmetric, or
 * mix(i, j, seed) otherwise.  A_ii = sum_j |A_ij| (ascending j) + diag_shift,
 * so A is strictly diagonally dominant (SPD when symmetric).
 */
static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static uint64_t mix3(uint64_t a, uint64_t b, uint64_t seed)
{
    return splitmix64(splitmix64(a ^ seed) + b * 0xD6E8FEB86659FD93ull);
}

static double unit_open0(uint64_t h) { return (double)((h >> 11) + 1) * (1.0 / 9007199254740992.0); } /* (0,1] */
static double unit_open1(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }       /* [0,1) */

static int64_t gcd64(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }

static int64_t modinv(int64_t a, int64_t n)
{
    int64_t t = 0, nt = 1, r = n, nr = a % n;
    while (nr) {
        int64_t qq = r / nr, tmp = t - qq * nt; t = nt; nt = tmp;
        tmp = r - qq * nr; r = nr; nr = tmp;
    }
    return t < 0 ? t + n : t;
}

void orc_gen_init(orc_gen *g, int64_t n, int npairs, int64_t band, int symmetric,
                  uint64_t seed, double diag_shift)
{
    orc_gen_init_ex(g, n, npairs, band > 0 ? 1 : 0, band, symmetric, seed, diag_shift);
}

void orc_gen_init_ex(orc_gen *g, int64_t n, int npairs, int pattern, int64_t band, int symmetric,
                     uint64_t seed, double diag_shift)
{
    memset(g, 0, sizeof *g);
    g->n = n; g->npairs = npairs > 16 ? 16 : npairs; g->banded = pattern;
    g->symmetric = symmetric; g->seed = seed; g->diag_shift = diag_shift;
    uint64_t s = splitmix64(seed ^ 0xA5A5A5A55A5A5A5Aull);
    if (band > n - 1) band = n - 1;
    if (band < 1) band = 1;
    if (pattern == 2) {
        /* block = largest power of two <= band/2 (at least 1): offsets stay below 2*block <= band */
        int L = 0;
        while (L < 30 && ((int64_t)2 << L) <= band / 2) L++;
        g->wb_log2 = L;
        for (int k = 0; k < g->npairs; k++) {
            s = splitmix64(s);
            g->a[k] = 1; g->ainv[k] = 1; g->c[k] = (int64_t)(s >> 1);
        }
        return;
    }
    for (int k = 0; k < g->npairs; k++) {
        if (g->banded) {
            /* offsets in [1, band]; k == 0 is the nearest neighbour; redraw a
             * repeated offset up to 64 times (a repeat that survives is merged
             * per row, so it only lowers the row length) */
            int64_t c = 1;
            for (int tries = 0; tries < 64; tries++) {
                s = splitmix64(s);
                c = (k == 0 || band < 2) ? 1 : 2 + (int64_t)(s % (uint64_t)(band - 1));
                int dup = 0;
                for (int j = 0; j < k; j++) if (g->c[j] == c) dup = 1;
                if (!dup) break;
            }
            g->a[k] = 1; g->ainv[k] = 1; g->c[k] = c;
        } else {
            int64_t a;
            do { s = splitmix64(s); a = 2 + (int64_t)(s % (uint64_t)(n > 3 ? n - 2 : 1)); }
            while (gcd64(a, n) != 1);
            s = splitmix64(s);
            g->a[k] = a; g->ainv[k] = modinv(a, n); g->c[k] = (int64_t)(s % (uint64_t)n);
        }
    }
}

/* pattern 2: keyed bijection of [0, 2^L) (affine step, xor-shift, odd multiplier, xor-shift: every step
 * is invertible modulo 2^L, and with 2*sh >= L the xor-shift is its own inverse) */
static uint64_t inv_pow2(uint64_t a)    /* a odd: a^-1 modulo 2^64 (Newton) */
{
    uint64_t x = a;
    for (int it = 0; it < 6; it++) x *= 2 - a * x;
    return x;
}
static void blk_keys(const orc_gen *g, int k, int64_t b, uint64_t *a1, uint64_t *c1, uint64_t *a2)
{
    const uint64_t mask = ((uint64_t)1 << g->wb_log2) - 1;
    const uint64_t key = mix3((uint64_t)g->c[k], (uint64_t)b, g->seed);
    *a1 = (key | 1) & mask; *c1 = (key >> 21) & mask; *a2 = ((key >> 42) | 1) & mask;
    if (g->wb_log2 == 0) { *a1 = 1; *a2 = 1; }
}
static uint64_t blk_fwd(const orc_gen *g, int k, int64_t b, uint64_t u)
{
    const int L = g->wb_log2, sh = (L + 1) / 2;
    const uint64_t mask = ((uint64_t)1 << L) - 1;
    uint64_t a1, c1, a2; blk_keys(g, k, b, &a1, &c1, &a2);
    uint64_t x = (u * a1 + c1) & mask;
    if (sh) x ^= x >> sh;
    x = (x * a2) & mask;
    if (sh) x ^= x >> sh;
    return x;
}
static uint64_t blk_inv(const orc_gen *g, int k, int64_t b, uint64_t y)
{
    const int L = g->wb_log2, sh = (L + 1) / 2;
    const uint64_t mask = ((uint64_t)1 << L) - 1;
    uint64_t a1, c1, a2; blk_keys(g, k, b, &a1, &c1, &a2);
    uint64_t x = y;
    if (sh) x ^= x >> sh;
    x = (x * inv_pow2(a2)) & mask;
    if (sh) x ^= x >> sh;
    return ((x - c1) * inv_pow2(a1)) & mask;
}

/* candidate columns of row i, deduplicated and sorted; returns their number
 * (diagonal excluded). */
static int row_cols(const orc_gen *g, int64_t i, int64_t *out)
{
    int cnt = 0;
    for (int k = 0; k < g->npairs; k++) {
        int64_t j[2];
        if (g->banded == 2) {
            const int L = g->wb_log2;
            const int64_t b = i >> L;
            const uint64_t u = (uint64_t)i & (((uint64_t)1 << L) - 1);
            j[0] = ((b + 1) << L) + (int64_t)blk_fwd(g, k, b, u);
            j[1] = b >= 1 ? ((b - 1) << L) + (int64_t)blk_inv(g, k, b - 1, u) : -1;
        } else if (g->banded) {
            j[0] = i + g->c[k];
            j[1] = i - g->c[k];
        } else {
            j[0] = (int64_t)(((__int128)g->a[k] * i + g->c[k]) % g->n);
            int64_t d = i - g->c[k]; if (d < 0) d += g->n;
            j[1] = (int64_t)(((__int128)g->ainv[k] * d) % g->n);
        }
        for (int e = 0; e < 2; e++) {
            int64_t c = j[e];
            if (c < 0 || c >= g->n || c == i) continue;
            int pos = cnt, dup = 0;
            for (int t = 0; t < cnt; t++) if (out[t] == c) { dup = 1; break; }
            if (dup) continue;
            while (pos > 0 && out[pos - 1] > c) { out[pos] = out[pos - 1]; pos--; }
            out[pos] = c; cnt++;
        }
    }
    return cnt;
}

static double offdiag(const orc_gen *g, int64_t i, int64_t j)
{
    uint64_t h = g->symmetric ? mix3((uint64_t)(i < j ? i : j), (uint64_t)(i < j ? j : i), g->seed)
                              : mix3((uint64_t)i, (uint64_t)j, g->seed);
    return -unit_open0(h);
}

void orc_gen_count(const orc_gen *g, int64_t r0, int64_t r1, int *counts)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = r0; i < r1; i++) {
        int64_t tmp[32];
        counts[i - r0] = row_cols(g, i, tmp) + 1;
    }
}

void orc_gen_fill(const orc_gen *g, int64_t r0, int64_t r1, const int *rowptr, int *col,
                  double *val)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = r0; i < r1; i++) {
        int64_t tmp[32];
        int cnt = row_cols(g, i, tmp);
        int base = rowptr[i - r0], w = 0, dpos = -1;
        double sum = 0.0;
        for (int t = 0; t < cnt; t++) {
            if (dpos < 0 && tmp[t] > i) { dpos = base + w; w++; }
            double v = offdiag(g, i, tmp[t]);
            col[base + w] = (int)tmp[t]; val[base + w] = v; w++;
            sum += -v;
        }
        if (dpos < 0) { dpos = base + w; w++; }
        col[dpos] = (int)i;
        val[dpos] = sum + g->diag_shift;
    }
}

void orc_gen_xtrue(const orc_gen *g, int64_t r0, int64_t r1, double *x)
{
    for (int64_t i = r0; i < r1; i++)
        x[i - r0] = unit_open1(mix3((uint64_t)i, 0x7265757274ull, g->seed));
}
