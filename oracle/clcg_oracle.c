/*
 * clcg_oracle.c -- TEST INFRASTRUCTURE (see lcg_oracle.h).  Parity: PINNED
 * against the compiled reference (tests/test_oracle_vs_ref.py) -- except the two
 * preconditioned loops orc_clpcg and orc_clpbicg, which say "PARITY UNPINNED" at
 * their definitions: the reference has them in its CUDA / Eigen back-ends only.
 *
 * Complex (c128) solvers of liblcg's native back-end, restated in C99
 * `double _Complex`.  libstdc++'s std::complex<double> operators and GCC's
 * C99 complex arithmetic lower to the same code (component formulas for + - *,
 * __divdc3 for /, component-wise scaling for real*complex), which is what
 * makes bit-identical comparison with the reference possible.
 *
 * Deviations from the reference, both deliberate and documented in DESIGN.md:
 *  - rbar0 (shadow residual) is an argument instead of srand(time(0)) draws
 *    (clcg.cpp:399-403, 556-560, 721-725); orc_clcg_vecrnd restates the draw
 *    for a given seed so a test can replay the reference's own vector.
 *  - cltfqmr returns when max_iterations is hit.  The reference's `break`
 *    (clcg.cpp:800-804) only leaves the inner j-loop and the outer while(1)
 *    never terminates by itself.
 */
#include "lcg_oracle.h"

#include <math.h>
#include <stdlib.h>

typedef double _Complex zc;

static const orc_cpara orc_cdefaults = {0, 1e-6, 0}; /* util.h:278 */

/* lcg_complex.cpp:143-154: sum a_i*b_i (no conjugate), two real accumulators. */
zc orc_cdot(const zc *a, const zc *b, int n)
{
    double re = 0.0, im = 0.0;
    for (int i = 0; i < n; i++) {
        re += (creal(a[i]) * creal(b[i]) - cimag(a[i]) * cimag(b[i]));
        im += (creal(a[i]) * cimag(b[i]) + cimag(a[i]) * creal(b[i]));
    }
    return CMPLX(re, im);
}

/* lcg_complex.cpp:156-167: sum conj(a_i)*b_i. */
zc orc_cinner(const zc *a, const zc *b, int n)
{
    double re = 0.0, im = 0.0;
    for (int i = 0; i < n; i++) {
        re += (creal(a[i]) * creal(b[i]) + cimag(a[i]) * cimag(b[i]));
        im += (creal(a[i]) * cimag(b[i]) - cimag(a[i]) * creal(b[i]));
    }
    return CMPLX(re, im);
}

static double zsquare(zc a) { return creal(a) * creal(a) + cimag(a) * cimag(a); } /* lcg_complex.cpp:102-105 */
static double zmodule(zc a) { return sqrt(zsquare(a)); }                          /* lcg_complex.cpp:107-110 */

/* lcg_complex.cpp:118-127, with the seed made explicit. */
void orc_clcg_vecrnd(zc *a, double lre, double lim, double hre, double him, int n, unsigned seed)
{
    srand(seed);
    for (int i = 0; i < n; i++) {
        double re = (hre - lre) * rand() * 1.0 / RAND_MAX + lre;
        double im = (him - lim) * rand() * 1.0 / RAND_MAX + lim;
        a[i] = CMPLX(re, im);
    }
}

static int ccheck_args(const orc_cpara *p, int n, const zc *m, const zc *B)
{   /* clcg.cpp:235-240 and twins */
    if (n <= 0) return ORC_INVILAD_VARIABLE_SIZE;
    if (p->max_iterations < 0) return ORC_INVILAD_MAX_ITERATIONS;
    if (p->epsilon <= 0.0 || p->epsilon >= 1.0) return ORC_INVILAD_EPSILON;
    if (m == NULL || B == NULL) return ORC_C_INVALID_POINTER;
    return 0;
}

/* clcg.cpp:273-290: r4 = |<r,r>|^2, m4 = max(|<m,m>|^2, 1): 4th powers of norms. */
static int calready_done(const orc_cpara *p, orc_cprogress Pfp, void *inst, const zc *m,
                         double r4, double m4, int n)
{
    double r;
    if (p->abs_diff && sqrt(r4) / n <= p->epsilon) r = sqrt(r4) / n;
    else if (r4 / m4 <= p->epsilon) r = r4 / m4;
    else return 0;
    if (Pfp) Pfp(inst, m, r, p, n, 0);
    return 1;
}

/* clcg.cpp:295-318: returns 0 to continue. Note the LCG_ (real-enum) code for
 * the iteration cap: -1019. */
static int cloop_head(const orc_cpara *p, orc_cprogress Pfp, void *inst, const zc *m,
                      double r4, double m4, int n, int *t, int *ret)
{
    double r = p->abs_diff ? sqrt(r4) / n : r4 / m4;
    if (Pfp && Pfp(inst, m, r, p, n, *t)) { *ret = ORC_STOP; return 1; }
    if (r <= p->epsilon) { *ret = ORC_CONVERGENCE; return 1; }
    if (p->max_iterations > 0 && *t + 1 > p->max_iterations) {
        *ret = ORC_REACHED_MAX_ITERATIONS; return 1;
    }
    ++*t;
    return 0;
}

static int chas_nan(const zc *m, int n)
{   /* clcg.cpp:335-341: std::complex operator!= is (re!=re || im!=im) */
    for (int i = 0; i < n; i++)
        if (creal(m[i]) != creal(m[i]) || cimag(m[i]) != cimag(m[i])) return 1;
    return 0;
}

static double m4_of(const zc *m, int n)
{   /* clcg.cpp:262-264 */
    double v = zsquare(orc_cinner(m, m, n));
    return v < 1.0 ? 1.0 : v;
}

/* ----------------------------------------------------------------- BiCG */
int orc_clbicg(orc_caxfunc Afp, orc_cprogress Pfp, zc *m, const zc *B, int n,
               const orc_cpara *param, void *inst)
{
    orc_cpara p = param ? *param : orc_cdefaults;
    int ret = ccheck_args(&p, n, m, B);
    if (ret) return ret;
    zc *r1 = malloc(sizeof(zc) * n), *r2 = malloc(sizeof(zc) * n), *d1 = malloc(sizeof(zc) * n),
       *d2 = malloc(sizeof(zc) * n), *Ax = malloc(sizeof(zc) * n);
    int t = 0;

    Afp(inst, m, Ax, n, 0, 0);                                  /* clcg.cpp:99 */
    for (int i = 0; i < n; i++) {                               /* :101-106 */
        d1[i] = r1[i] = B[i] - Ax[i];
        d2[i] = r2[i] = conj(r1[i]);
    }
    zc rho = orc_cinner(r2, r1, n);                             /* :108-109 */
    double m4 = m4_of(m, n);                                    /* :111-115 */
    double r4 = zsquare(orc_cinner(r1, r1, n));                 /* :117-120 */

    if (calready_done(&p, Pfp, inst, m, r4, m4, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    while (!cloop_head(&p, Pfp, inst, m, r4, m4, n, &t, &ret)) {
        Afp(inst, d1, Ax, n, 0, 0);                             /* :169 */
        zc Add = orc_cinner(d2, Ax, n);                         /* :170 */
        zc ak = rho / Add;                                      /* :171 */
        for (int i = 0; i < n; i++) {                           /* :173-178 */
            m[i] = m[i] + ak * d1[i];
            r1[i] = r1[i] - ak * Ax[i];
        }
        m4 = m4_of(m, n);                                       /* :180-182 */
        r4 = zsquare(orc_cinner(r1, r1, n));                    /* :184-185 */
        Afp(inst, d2, Ax, n, 1, 1);                             /* :187: A^H */
        for (int i = 0; i < n; i++) r2[i] = r2[i] - conj(ak) * Ax[i];   /* :189-193 */
        if (chas_nan(m, n)) { ret = ORC_C_NAN_VALUE; goto out; } /* :195-201 */
        zc rho2 = orc_cinner(r2, r1, n);                        /* :203 */
        zc bk = rho2 / rho;                                     /* :204 */
        rho = rho2;
        for (int i = 0; i < n; i++) {                           /* :207-212 */
            d1[i] = r1[i] + bk * d1[i];
            d2[i] = r2[i] + conj(bk) * d2[i];
        }
    }
out:
    free(r1); free(r2); free(d1); free(d2); free(Ax);
    return ret;
}

/* ------------------------------------------------- preconditioned CG, complex-symmetric A
 * PARITY UNPINNED for this one function: clpcg exists only in the reference's CUDA back-end
 * (clcg_cuda.cu:403-558) and its Eigen back-end (clcg_eigen.cpp:577); neither can be built here
 * (no nvcc, no Eigen3), so there is no reference output to pin against.  Restated from the CUDA
 * source with the cuBLAS calls spelled out: Zdotu = sum x_i y_i, Dznrm2 = sqrt(sum |x_i|^2).
 * Its stop rule is the REAL solvers' |r|^2 / max(|m|^2, 1) (clcg_cuda.cu:459,472), not the 4th
 * power of the CPU complex loops.  In abs_diff mode the CUDA code never computes m_mod yet reads it
 * in the "already optimised" else-if (:456-466): here only the abs_diff test is
 * made in that mode.  The CUDA loop has no NaN scan. */
int orc_clpcg(orc_caxfunc Afp, orc_caxfunc Mfp, orc_cprogress Pfp, zc *m, const zc *B, int n,
              const orc_cpara *param, void *inst)
{
    orc_cpara p = param ? *param : orc_cdefaults;
    int ret = ccheck_args(&p, n, m, B);
    if (ret) return ret;
    zc *r = malloc(sizeof(zc) * n), *d = malloc(sizeof(zc) * n), *s = malloc(sizeof(zc) * n), *Ax = malloc(sizeof(zc) * n);
    int t = 0;
    Afp(inst, m, Ax, n, 0, 0);                                  /* clcg_cuda.cu:441 */
    for (int i = 0; i < n; i++) r[i] = B[i] - Ax[i];            /* :442-443 */
    Mfp(inst, r, d, n, 0, 0);                                   /* :445 */
    zc dn = orc_cdot(r, d, n);                                  /* :448 Zdotu */
    double m2 = 1.0, r2 = creal(orc_cinner(r, r, n));           /* :450-457 */
    if (!p.abs_diff) { m2 = creal(orc_cinner(m, m, n)); if (m2 < 1.0) m2 = 1.0; }
    if (p.abs_diff ? sqrt(r2) / n <= p.epsilon : r2 / m2 <= p.epsilon) {
        if (Pfp) Pfp(inst, m, p.abs_diff ? sqrt(r2) / n : r2 / m2, &p, n, 0);
        ret = ORC_ALREADY_OPTIMIZIED; goto out;
    }
    for (;;) {
        const double res = p.abs_diff ? sqrt(r2) / n : r2 / m2;  /* :478-479 */
        if (Pfp && Pfp(inst, m, res, &p, n, t)) { ret = ORC_STOP; goto out; }
        if (res <= p.epsilon) { ret = ORC_CONVERGENCE; goto out; }
        if (p.max_iterations > 0 && t + 1 > p.max_iterations) { ret = ORC_REACHED_MAX_ITERATIONS; break; }
        t++;
        Afp(inst, d, Ax, n, 0, 0);                              /* :500 */
        zc dAx = orc_cdot(d, Ax, n);                            /* :501 */
        zc ak = dn / dAx;                                       /* :502 */
        for (int i = 0; i < n; i++) { m[i] = m[i] + ak * d[i]; r[i] = r[i] - ak * Ax[i]; }   /* :504-505 */
        if (!p.abs_diff) { m2 = creal(orc_cinner(m, m, n)); if (m2 < 1.0) m2 = 1.0; }
        r2 = creal(orc_cinner(r, r, n));                        /* :511 */
        Mfp(inst, r, s, n, 0, 0);                               /* :513 */
        zc dold = dn;
        dn = orc_cdot(r, s, n);                                 /* :516 */
        zc bk = dn / dold;
        for (int i = 0; i < n; i++) d[i] = bk * d[i] + s[i];    /* :519-520 */
    }
out:
    free(r); free(d); free(s); free(Ax);
    return ret;
}

/* ------------------------------------------------- preconditioned BiCG
 * PARITY UNPINNED like orc_clpcg: clpbicg exists only in the reference's Eigen back-end (clcg_eigen.cpp:685-802; the
 * default of clcg_solver_preconditioned_eigen, clcg_eigen.h:87-92) and Eigen3 is not in the image.  Restated from that
 * source with Eigen's conventions spelled out: a.dot(b) = sum conj(a_i) b_i; std::norm(z) = |z|^2, so the stop rule is the
 * 4th-power one of the CPU loops (m_mod = |<m,m>|^2 clamped to 1, rk_mod = |<r,r>|^2; abs_diff: sqrt(rk_mod) / n).  Two
 * products per iteration: A.p and conj(A).ps (Afp(psk, Asx, MatNormal, Conjugate), :761).  The shadow residual is NOT a
 * recurrence: rsk = conj(rk_old) - conj(ak) Asx every iteration (:767), restated as written.  No NaN scan in the loop. */
int orc_clpbicg(orc_caxfunc Afp, orc_caxfunc Mfp, orc_cprogress Pfp, zc *m, const zc *B, int n,
                const orc_cpara *param, void *inst)
{
    orc_cpara p = param ? *param : orc_cdefaults;
    int ret = ccheck_args(&p, n, m, B);                         /* clcg_eigen.cpp:693-697 */
    if (ret) return ret;
    zc *r = malloc(sizeof(zc) * n), *rs = malloc(sizeof(zc) * n), *z = malloc(sizeof(zc) * n), *pk = malloc(sizeof(zc) * n),
       *ps = malloc(sizeof(zc) * n), *Ax = malloc(sizeof(zc) * n), *Asx = malloc(sizeof(zc) * n);
    int t = 0;
    Afp(inst, m, Ax, n, 0, 0);                                  /* :702 */
    for (int i = 0; i < n; i++) r[i] = B[i] - Ax[i];            /* :704 */
    Mfp(inst, r, z, n, 0, 0);                                   /* :705 */
    for (int i = 0; i < n; i++) { pk[i] = z[i]; rs[i] = conj(r[i]); ps[i] = conj(z[i]); }   /* :707-709 */
    zc rho = orc_cinner(rs, z, n);                              /* :711 rsk.dot(zk) */
    double m4 = m4_of(m, n);                                    /* :713-714 */
    double r4 = zsquare(orc_cinner(r, r, n));                   /* :716 */
    if (calready_done(&p, Pfp, inst, m, r4, m4, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }    /* :719-736 */
    while (!cloop_head(&p, Pfp, inst, m, r4, m4, n, &t, &ret)) { /* :741-758 */
        Afp(inst, pk, Ax, n, 0, 0);                             /* :760 */
        Afp(inst, ps, Asx, n, 0, 1);                            /* :761 conj(A).ps */
        zc pAx = orc_cinner(ps, Ax, n);                         /* :763 psk.dot(Ax) */
        zc ak = rho / pAx;                                      /* :764 */
        for (int i = 0; i < n; i++) {                           /* :766-768 */
            m[i] = m[i] + ak * pk[i];
            rs[i] = conj(r[i]) - conj(ak) * Asx[i];
            r[i] = r[i] - ak * Ax[i];
        }
        m4 = m4_of(m, n);                                       /* :770-771 */
        r4 = zsquare(orc_cinner(r, r, n));                      /* :773 */
        Mfp(inst, r, z, n, 0, 0);                               /* :775 */
        zc rho2 = orc_cinner(rs, z, n);                         /* :777 */
        zc bk = rho2 / rho;                                     /* :778 */
        rho = rho2;
        for (int i = 0; i < n; i++) {                           /* :781-782 */
            pk[i] = z[i] + bk * pk[i];
            ps[i] = conj(z[i]) + conj(bk) * ps[i];
        }
    }
out:
    free(r); free(rs); free(z); free(pk); free(ps); free(Ax); free(Asx);
    return ret;
}

/* ------------------------------------------------- BiCG, complex-symmetric A */
int orc_clbicg_symmetric(orc_caxfunc Afp, orc_cprogress Pfp, zc *m, const zc *B, int n,
                         const orc_cpara *param, void *inst)
{
    orc_cpara p = param ? *param : orc_cdefaults;
    int ret = ccheck_args(&p, n, m, B);
    if (ret) return ret;
    zc *r = malloc(sizeof(zc) * n), *d = malloc(sizeof(zc) * n), *Ax = malloc(sizeof(zc) * n);
    int t = 0;

    Afp(inst, m, Ax, n, 0, 0);                                  /* clcg.cpp:250 */
    for (int i = 0; i < n; i++) d[i] = r[i] = B[i] - Ax[i];     /* :252-256 */
    zc rr = orc_cdot(r, r, n);                                  /* :258-259 */
    double m4 = m4_of(m, n);                                    /* :261-265 */
    double r4 = zsquare(orc_cinner(r, r, n));                   /* :267-270 */

    if (calready_done(&p, Pfp, inst, m, r4, m4, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    while (!cloop_head(&p, Pfp, inst, m, r4, m4, n, &t, &ret)) {
        Afp(inst, d, Ax, n, 0, 0);                              /* :319 */
        zc dAx = orc_cdot(d, Ax, n);                            /* :320 */
        zc ak = rr / dAx;                                       /* :321 */
        for (int i = 0; i < n; i++) {                           /* :323-328 */
            m[i] = m[i] + ak * d[i];
            r[i] = r[i] - ak * Ax[i];
        }
        m4 = m4_of(m, n);                                       /* :330-332 */
        r4 = zsquare(orc_cinner(r, r, n));                      /* :334-335 */
        if (chas_nan(m, n)) { ret = ORC_C_NAN_VALUE; goto out; } /* :337-343 */
        zc rr2 = orc_cdot(r, r, n);                             /* :345 */
        zc bk = rr2 / rr;                                       /* :346 */
        rr = rr2;
        for (int i = 0; i < n; i++) d[i] = r[i] + bk * d[i];    /* :349-353 */
    }
out:
    free(r); free(d); free(Ax);
    return ret;
}

/* ----------------------------------------------------------------- CGS */
int orc_clcgs(orc_caxfunc Afp, orc_cprogress Pfp, zc *m, const zc *B, int n,
              const orc_cpara *param, void *inst, const zc *rbar0)
{
    orc_cpara p = param ? *param : orc_cdefaults;
    int ret = ccheck_args(&p, n, m, B);
    if (ret) return ret;
    zc *r = malloc(sizeof(zc) * n), *pk = malloc(sizeof(zc) * n), *Ax = malloc(sizeof(zc) * n),
       *u = malloc(sizeof(zc) * n), *q = malloc(sizeof(zc) * n), *w = malloc(sizeof(zc) * n);
    int t = 0;

    Afp(inst, m, Ax, n, 0, 0);                                  /* clcg.cpp:391 */
    for (int i = 0; i < n; i++) pk[i] = u[i] = r[i] = B[i] - Ax[i]; /* :393-397 */
    zc rho = orc_cinner(rbar0, r, n);                           /* :399-404 (draw made by the caller) */
    double m4 = m4_of(m, n);                                    /* :406-410 */
    double r4 = zsquare(orc_cinner(r, r, n));                   /* :412-415 */

    if (calready_done(&p, Pfp, inst, m, r4, m4, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    while (!cloop_head(&p, Pfp, inst, m, r4, m4, n, &t, &ret)) {
        Afp(inst, pk, Ax, n, 0, 0);                             /* :463 */
        zc sigma = orc_cinner(rbar0, Ax, n);                    /* :464 */
        zc ak = rho / sigma;                                    /* :465 */
        for (int i = 0; i < n; i++) {                           /* :467-472 */
            q[i] = u[i] - ak * Ax[i];
            w[i] = u[i] + q[i];
        }
        Afp(inst, w, Ax, n, 0, 0);                              /* :474 */
        for (int i = 0; i < n; i++) {                           /* :476-481 */
            m[i] = m[i] + ak * w[i];
            r[i] = r[i] - ak * Ax[i];
        }
        m4 = m4_of(m, n);                                       /* :483-485 */
        r4 = zsquare(orc_cinner(r, r, n));                      /* :487-488 */
        if (chas_nan(m, n)) { ret = ORC_C_NAN_VALUE; goto out; } /* :490-496 */
        zc rho2 = orc_cinner(rbar0, r, n);                      /* :498 */
        zc bk = rho2 / rho;                                     /* :499 */
        rho = rho2;
        for (int i = 0; i < n; i++) {                           /* :502-507 */
            u[i] = r[i] + bk * q[i];
            pk[i] = u[i] + bk * (q[i] + bk * pk[i]);
        }
    }
out:
    free(r); free(pk); free(Ax); free(u); free(q); free(w);
    return ret;
}

/* ------------------------------------------------------------ BiCGStab */
int orc_clbicgstab(orc_caxfunc Afp, orc_cprogress Pfp, zc *m, const zc *B, int n,
                   const orc_cpara *param, void *inst, const zc *rbar0)
{
    orc_cpara p = param ? *param : orc_cdefaults;
    int ret = ccheck_args(&p, n, m, B);
    if (ret) return ret;
    zc *r = malloc(sizeof(zc) * n), *pk = malloc(sizeof(zc) * n), *s = malloc(sizeof(zc) * n),
       *Ap = malloc(sizeof(zc) * n), *As = malloc(sizeof(zc) * n);
    int t = 0;

    Afp(inst, m, Ap, n, 0, 0);                                  /* clcg.cpp:548 */
    for (int i = 0; i < n; i++) pk[i] = r[i] = B[i] - Ap[i];    /* :550-554 */
    zc rho = orc_cinner(rbar0, r, n);                           /* :556-561 */
    double m4 = m4_of(m, n);                                    /* :563-567 */
    double r4 = zsquare(orc_cinner(r, r, n));                   /* :569-572 */

    if (calready_done(&p, Pfp, inst, m, r4, m4, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    while (!cloop_head(&p, Pfp, inst, m, r4, m4, n, &t, &ret)) {
        Afp(inst, pk, Ap, n, 0, 0);                             /* :620 */
        zc sigma = orc_cinner(rbar0, Ap, n);                    /* :621 */
        zc ak = rho / sigma;                                    /* :622 */
        for (int i = 0; i < n; i++) s[i] = r[i] - ak * Ap[i];   /* :624-628 */
        Afp(inst, s, As, n, 0, 0);                              /* :630 */
        zc Ass = orc_cinner(As, s, n);                          /* :631 */
        zc AsAs = orc_cinner(As, As, n);                        /* :632 */
        zc omega = Ass / AsAs;                                  /* :633 */
        for (int i = 0; i < n; i++) {                           /* :635-640 */
            m[i] = m[i] + ak * pk[i] + omega * s[i];
            r[i] = s[i] - omega * As[i];
        }
        m4 = m4_of(m, n);                                       /* :642-644 */
        r4 = zsquare(orc_cinner(r, r, n));                      /* :646-647 */
        if (chas_nan(m, n)) { ret = ORC_C_NAN_VALUE; goto out; } /* :649-655 */
        zc rho2 = orc_cinner(rbar0, r, n);                      /* :657 */
        zc bk = rho2 * ak / (rho * omega);                      /* :658 */
        rho = rho2;
        for (int i = 0; i < n; i++) pk[i] = r[i] + bk * (pk[i] - omega * Ap[i]); /* :661-665 */
    }
out:
    free(r); free(pk); free(s); free(Ap); free(As);
    return ret;
}

/* --------------------------------------------------------------- TFQMR */
int orc_cltfqmr(orc_caxfunc Afp, orc_cprogress Pfp, zc *m, const zc *B, int n,
                const orc_cpara *param, void *inst, const zc *rbar0)
{
    orc_cpara p = param ? *param : orc_cdefaults;
    int ret = ccheck_args(&p, n, m, B);
    if (ret) return ret;
    zc *pk = malloc(sizeof(zc) * n), *u = malloc(sizeof(zc) * n), *v = malloc(sizeof(zc) * n),
       *d = malloc(sizeof(zc) * n), *r = malloc(sizeof(zc) * n), *Ax = malloc(sizeof(zc) * n),
       *q = malloc(sizeof(zc) * n), *uq = malloc(sizeof(zc) * n);
    int t = 0;

    Afp(inst, m, Ax, n, 0, 0);                                  /* clcg.cpp:707 */
    for (int i = 0; i < n; i++) {                               /* :709-714 */
        pk[i] = u[i] = r[i] = B[i] - Ax[i];
        d[i] = CMPLX(0.0, 0.0);
    }
    zc rr = orc_cinner(r, r, n);                                /* :718 */
    double r4 = zsquare(rr);                                    /* :719 */
    zc rho = orc_cinner(rbar0, r, n);                           /* :721-725 */

    double theta = 0.0, omega = zmodule(rr);                    /* :727 */
    double tao = omega;                                         /* :728 */
    zc eta = CMPLX(0.0, 0.0);                                   /* :729 */
    double m4 = m4_of(m, n);                                    /* :731-735 */

    if (calready_done(&p, Pfp, inst, m, r4, m4, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    for (;;) {
        Afp(inst, pk, v, n, 0, 0);                              /* :759 */
        zc sigma = orc_cinner(rbar0, v, n);                     /* :761 */
        zc alpha = rho / sigma;                                 /* :762 */
        for (int i = 0; i < n; i++) {                           /* :764-769 */
            q[i] = u[i] - alpha * v[i];
            uq[i] = u[i] + q[i];
        }
        Afp(inst, uq, Ax, n, 0, 0);                             /* :771 */
        for (int i = 0; i < n; i++) r[i] = r[i] - alpha * Ax[i]; /* :773-777 */
        zc rr2 = orc_cinner(r, r, n);                           /* :779 */

        for (int j = 1; j <= 2; j++) {                          /* :781 */
            /* :784-805.  r4 still holds the value from the PREVIOUS outer
             * pass here (it is refreshed at :854, after this loop). */
            if (cloop_head(&p, Pfp, inst, m, r4, m4, n, &t, &ret)) goto out;

            zc sign = theta * theta * (eta / alpha);            /* :809 */
            if (j == 1) {
                omega = sqrt(zmodule(rr) * zmodule(rr2));       /* :813 */
                for (int i = 0; i < n; i++) d[i] = u[i] + sign * d[i]; /* :815-819 */
            } else {
                omega = zmodule(rr2);                           /* :823 */
                for (int i = 0; i < n; i++) d[i] = q[i] + sign * d[i]; /* :825-829 */
            }
            theta = omega / tao;                                /* :832 */
            tao = omega / sqrt(1.0 + theta * theta);            /* :833 */
            eta = (1.0 / (1.0 + theta * theta)) * alpha;        /* :834 */
            for (int i = 0; i < n; i++) m[i] = m[i] + eta * d[i]; /* :836-840 */
            m4 = m4_of(m, n);                                   /* :842-844 */
            if (chas_nan(m, n)) { ret = ORC_C_NAN_VALUE; goto out; } /* :846-852 */
        }
        rr = rr2;                                               /* :853 */
        r4 = zsquare(rr);                                       /* :854 */
        zc rho2 = orc_cinner(rbar0, r, n);                      /* :856 */
        zc bk = rho2 / rho;                                     /* :857 */
        rho = rho2;
        for (int i = 0; i < n; i++) {                           /* :860-865 */
            u[i] = r[i] + bk * q[i];
            pk[i] = u[i] + bk * (q[i] + bk * pk[i]);
        }
    }
out:
    free(pk); free(u); free(v); free(d); free(r); free(Ax); free(q); free(uq);
    return ret;
}
