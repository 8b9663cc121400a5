/*
 * lcg_oracle.c -- TEST INFRASTRUCTURE (see lcg_oracle.h).  Parity: PINNED
 * against the compiled reference (tests/test_oracle_vs_ref.py) and the
 * committed goldens (tests/golden/).
 *
 * Real fp64 solvers of liblcg's native back-end, restated.  Arithmetic is kept
 * in the reference's order -- serial left-to-right dot products
 * (algebra.cpp:154-163 and the inline loops in lcg.cpp), element-wise updates
 * with the same expression shapes -- so that results are bit-identical to the
 * reference when both are built without FMA contraction.
 */
#include "lcg_oracle.h"

#include <math.h>
#include <stdlib.h>

static const orc_para orc_defaults = {0, 1e-6, 0, 1e-6, 1.0, 0.95, 0.9, 10}; /* util.h:153 */

/* algebra.cpp:154-163: strictly sequential accumulation. */
double orc_dot(const double *a, const double *b, int n)
{
    double s = 0.0;
    for (int i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* lcg.cpp:150-155 (identical block opens every real solver). */
static int check_args(const orc_para *p, int n, const double *m, const double *B)
{
    if (n <= 0) return ORC_INVILAD_VARIABLE_SIZE;
    if (p->max_iterations < 0) return ORC_INVILAD_MAX_ITERATIONS;
    if (p->epsilon <= 0.0 || p->epsilon >= 1.0) return ORC_INVILAD_EPSILON;
    if (m == NULL || B == NULL) return ORC_INVALID_POINTER;
    return 0;
}

/* lcg.cpp:208-209: the monitored quantity.  Gradient mode is a ratio of
 * SQUARED norms; abs_diff mode is sqrt(|g|^2)/n. */
static double residual_of(const orc_para *p, double g2, double m2, int n)
{
    return p->abs_diff ? sqrt(g2) / n : g2 / m2;
}

/* lcg.cpp:186-203: test performed once before the loop.  Returns 1 when the
 * start vector already satisfies the tolerance (Pfp is told, with k = 0). */
static int already_done(const orc_para *p, orc_progress Pfp, void *inst, const double *m,
                        double g2, double m2, int n)
{
    double r;
    if (p->abs_diff && sqrt(g2) / n <= p->epsilon) r = sqrt(g2) / n;
    else if (g2 / m2 <= p->epsilon) r = g2 / m2;
    else return 0;
    if (Pfp) Pfp(inst, m, r, p, n, 0);
    return 1;
}

/* lcg.cpp:208-230: head of every iteration.  0 = carry on (t has been
 * advanced), otherwise *ret holds the status to return. */
static int loop_head(const orc_para *p, orc_progress Pfp, void *inst, const double *m,
                     double g2, double m2, int n, int *t, int *ret)
{
    double r = residual_of(p, g2, m2, n);
    if (Pfp && Pfp(inst, m, r, p, n, *t)) { *ret = ORC_STOP; return 1; }
    if (r <= p->epsilon) { *ret = ORC_CONVERGENCE; return 1; }
    if (p->max_iterations > 0 && *t + 1 > p->max_iterations) {
        *ret = ORC_REACHED_MAX_ITERATIONS; return 1;
    }
    ++*t;
    return 0;
}

/* lcg.cpp:247-253 */
static int has_nan(const double *m, int n)
{
    for (int i = 0; i < n; i++) if (m[i] != m[i]) return 1;
    return 0;
}

static double clamp1(double v) { return v < 1.0 ? 1.0 : v; } /* lcg.cpp:180,245 */

/* ------------------------------------------------------------------ CG */
int orc_lcg(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, int n,
            const orc_para *param, void *inst)
{
    orc_para p = param ? *param : orc_defaults;                 /* lcg.cpp:147 */
    int ret = check_args(&p, n, m, B);
    if (ret) return ret;

    double *g = malloc(sizeof(double) * n), *d = malloc(sizeof(double) * n),
           *Ad = malloc(sizeof(double) * n);
    int t = 0;

    Afp(inst, m, Ad, n);                                        /* :168 */
    for (int i = 0; i < n; i++) { g[i] = Ad[i] - B[i]; d[i] = -1.0 * g[i]; } /* :171-176 */
    double m2 = clamp1(orc_dot(m, m, n));                       /* :178-180 */
    double g2 = orc_dot(g, g, n);                               /* :182-183 */

    if (already_done(&p, Pfp, inst, m, g2, m2, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    while (!loop_head(&p, Pfp, inst, m, g2, m2, n, &t, &ret)) {
        Afp(inst, d, Ad, n);                                    /* :232 */
        double dAd = orc_dot(d, Ad, n);                         /* :234 */
        double ak = g2 / dAd;                                   /* :235 */
        for (int i = 0; i < n; i++) { m[i] += ak * d[i]; g[i] += ak * Ad[i]; } /* :237-242 */
        m2 = clamp1(orc_dot(m, m, n));                          /* :244-245 */
        if (has_nan(m, n)) { ret = ORC_NAN_VALUE; goto out; }   /* :247-253 */
        double g2n = orc_dot(g, g, n);                          /* :255 */
        double bk = g2n / g2;                                   /* :256 */
        g2 = g2n;
        for (int i = 0; i < n; i++) d[i] = bk * d[i] - g[i];    /* :259-263 */
    }
out:
    free(g); free(d); free(Ad);
    return ret;
}

/* ----------------------------------------------------------------- PCG */
int orc_lpcg(orc_axfunc Afp, orc_axfunc Mfp, orc_progress Pfp, double *m, const double *B,
             int n, const orc_para *param, void *inst)
{
    orc_para p = param ? *param : orc_defaults;
    int ret = check_args(&p, n, m, B);
    if (ret) return ret;

    double *r = malloc(sizeof(double) * n), *z = malloc(sizeof(double) * n),
           *d = malloc(sizeof(double) * n), *Ad = malloc(sizeof(double) * n);
    int t = 0;

    Afp(inst, m, Ad, n);                                        /* lcg.cpp:314 */
    for (int i = 0; i < n; i++) r[i] = B[i] - Ad[i];            /* :317-321 */
    Mfp(inst, r, z, n);                                         /* :323 */
    for (int i = 0; i < n; i++) d[i] = z[i];                    /* :325-329 */
    double m2 = clamp1(orc_dot(m, m, n));                       /* :331-333 */
    double r2 = orc_dot(r, r, n);                               /* :335-336 */
    double zr = orc_dot(z, r, n);                               /* :338-339 */

    if (already_done(&p, Pfp, inst, m, r2, m2, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    while (!loop_head(&p, Pfp, inst, m, r2, m2, n, &t, &ret)) {
        Afp(inst, d, Ad, n);                                    /* :387 */
        double dAd = orc_dot(d, Ad, n);                         /* :389 */
        double ak = zr / dAd;                                   /* :390 */
        for (int i = 0; i < n; i++) { m[i] += ak * d[i]; r[i] -= ak * Ad[i]; } /* :392-397 */
        Mfp(inst, r, z, n);                                     /* :399 */
        m2 = clamp1(orc_dot(m, m, n));                          /* :401-402 */
        r2 = orc_dot(r, r, n);                                  /* :404 */
        if (has_nan(m, n)) { ret = ORC_NAN_VALUE; goto out; }   /* :406-412 */
        double zrn = orc_dot(z, r, n);                          /* :414 */
        double bk = zrn / zr;                                   /* :415 */
        zr = zrn;
        for (int i = 0; i < n; i++) d[i] = z[i] + bk * d[i];    /* :418-422 */
    }
out:
    free(r); free(z); free(d); free(Ad);
    return ret;
}

/* ----------------------------------------------------------------- CGS */
int orc_lcgs(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, int n,
             const orc_para *param, void *inst)
{
    orc_para p = param ? *param : orc_defaults;
    int ret = check_args(&p, n, m, B);
    if (ret) return ret;

    double *r = malloc(sizeof(double) * n), *r0 = malloc(sizeof(double) * n),
           *pk = malloc(sizeof(double) * n), *Ax = malloc(sizeof(double) * n),
           *u = malloc(sizeof(double) * n), *q = malloc(sizeof(double) * n),
           *w = malloc(sizeof(double) * n);
    int t = 0;

    Afp(inst, m, Ax, n);                                        /* lcg.cpp:476 */
    for (int i = 0; i < n; i++) pk[i] = u[i] = r0[i] = r[i] = B[i] - Ax[i]; /* :480-484 */
    double rho = 0.0;
    for (int i = 0; i < n; i++) rho += r[i] * r0[i];            /* :486-490 */
    double m2 = clamp1(orc_dot(m, m, n));                       /* :492-494 */
    double r2 = orc_dot(r, r, n);                               /* :496-497 */

    if (already_done(&p, Pfp, inst, m, r2, m2, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    while (!loop_head(&p, Pfp, inst, m, r2, m2, n, &t, &ret)) {
        Afp(inst, pk, Ax, n);                                   /* :546 */
        double Apr = 0.0;
        for (int i = 0; i < n; i++) Apr += Ax[i] * r0[i];       /* :548-552 */
        double ak = rho / Apr;                                  /* :553 */
        for (int i = 0; i < n; i++) { q[i] = u[i] - ak * Ax[i]; w[i] = u[i] + q[i]; } /* :556-560 */
        Afp(inst, w, Ax, n);                                    /* :562 */
        for (int i = 0; i < n; i++) { m[i] += ak * w[i]; r[i] -= ak * Ax[i]; } /* :565-569 */
        m2 = clamp1(orc_dot(m, m, n));                          /* :571-572 */
        r2 = orc_dot(r, r, n);                                  /* :574 */
        if (has_nan(m, n)) { ret = ORC_NAN_VALUE; goto out; }   /* :576-582 */
        double rhon = 0.0;
        for (int i = 0; i < n; i++) rhon += r[i] * r0[i];       /* :584-588 */
        double bk = rhon / rho;                                 /* :589 */
        rho = rhon;
        for (int i = 0; i < n; i++) {                           /* :593-597 */
            u[i] = r[i] + bk * q[i];
            pk[i] = u[i] + bk * (q[i] + bk * pk[i]);
        }
    }
out:
    free(r); free(r0); free(pk); free(Ax); free(u); free(q); free(w);
    return ret;
}

/* ------------------------------------------------------------ BiCGStab */
int orc_lbicgstab(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, int n,
                  const orc_para *param, void *inst)
{
    orc_para p = param ? *param : orc_defaults;
    int ret = check_args(&p, n, m, B);
    if (ret) return ret;

    double *r = malloc(sizeof(double) * n), *r0 = malloc(sizeof(double) * n),
           *pk = malloc(sizeof(double) * n), *Ax = malloc(sizeof(double) * n),
           *s = malloc(sizeof(double) * n), *Ap = malloc(sizeof(double) * n);
    int t = 0;

    Afp(inst, m, Ax, n);                                        /* lcg.cpp:648 */
    for (int i = 0; i < n; i++) pk[i] = r0[i] = r[i] = B[i] - Ax[i]; /* :650-654 */
    double rho = 0.0;
    for (int i = 0; i < n; i++) rho += r[i] * r0[i];            /* :656-660 */
    double m2 = clamp1(orc_dot(m, m, n));                       /* :662-664 */
    double r2 = orc_dot(r, r, n);                               /* :666-667 */

    if (already_done(&p, Pfp, inst, m, r2, m2, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    while (!loop_head(&p, Pfp, inst, m, r2, m2, n, &t, &ret)) {
        Afp(inst, pk, Ap, n);                                   /* :718 */
        double Apr = 0.0;
        for (int i = 0; i < n; i++) Apr += Ap[i] * r0[i];       /* :720-724 */
        double ak = rho / Apr;                                  /* :725 */
        for (int i = 0; i < n; i++) s[i] = r[i] - ak * Ap[i];   /* :727-731 */
        Afp(inst, s, Ax, n);                                    /* :733 */
        double Ass = 0.0, AsAs = 0.0;
        for (int i = 0; i < n; i++) { Ass += Ax[i] * s[i]; AsAs += Ax[i] * Ax[i]; } /* :735-740 */
        double wk = Ass / AsAs;                                 /* :741 */
        for (int i = 0; i < n; i++) m[i] += (ak * pk[i] + wk * s[i]); /* :743-747 */
        m2 = clamp1(orc_dot(m, m, n));                          /* :749-750 */
        if (has_nan(m, n)) { ret = ORC_NAN_VALUE; goto out; }   /* :752-758 */
        for (int i = 0; i < n; i++) r[i] = s[i] - wk * Ax[i];   /* :760-764 */
        r2 = orc_dot(r, r, n);                                  /* :766 */
        double rhon = 0.0;
        for (int i = 0; i < n; i++) rhon += r[i] * r0[i];       /* :768-772 */
        double bk = (ak / wk) * rhon / rho;                     /* :773 */
        rho = rhon;
        for (int i = 0; i < n; i++) pk[i] = r[i] + bk * (pk[i] - wk * Ap[i]); /* :776-780 */
    }
out:
    free(r); free(r0); free(pk); free(Ax); free(s); free(Ap);
    return ret;
}

/* --------------------------------------------- BiCGStab with restart (lcg.cpp:812-1034)
 * Differences from lbicgstab: its own argument checks (:818-821), a second stop test in the
 * middle of the iteration when abs_diff is set -- which advances t a second time (:897-925,
 * SURVEY quirk 8) -- and a restart r0 = p = r when |r.r0| < restart_epsilon (:975-990). */
int orc_lbicgstab2(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, int n,
                   const orc_para *param, void *inst)
{
    orc_para p = param ? *param : orc_defaults;
    if (n <= 0) return ORC_INVILAD_VARIABLE_SIZE;                       /* :818 */
    if (p.max_iterations < 0) return ORC_INVILAD_MAX_ITERATIONS;
    if (p.epsilon <= 0.0) return ORC_INVILAD_EPSILON;                   /* :820 */
    if (p.restart_epsilon <= 0.0 || p.epsilon >= 1.0) return ORC_INVILAD_RESTART_EPSILON;   /* :821 */
    if (m == NULL || B == NULL) return ORC_INVALID_POINTER;
    int ret = 0, t = 0;
    double *r = malloc(sizeof(double) * n), *r0 = malloc(sizeof(double) * n),
           *pk = malloc(sizeof(double) * n), *Ax = malloc(sizeof(double) * n),
           *s = malloc(sizeof(double) * n), *Ap = malloc(sizeof(double) * n);

    Afp(inst, m, Ax, n);                                                /* :832 */
    for (int i = 0; i < n; i++) pk[i] = r0[i] = r[i] = B[i] - Ax[i];    /* :834-838 */
    double rho = 0.0;
    for (int i = 0; i < n; i++) rho += r[i] * r0[i];                    /* :840-844 */
    double m2 = clamp1(orc_dot(m, m, n));
    double r2 = orc_dot(r, r, n);
    if (already_done(&p, Pfp, inst, m, r2, m2, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    while (!loop_head(&p, Pfp, inst, m, r2, m2, n, &t, &ret)) {
        Afp(inst, pk, Ap, n);                                           /* :895 */
        double Apr = 0.0;
        for (int i = 0; i < n; i++) Apr += Ap[i] * r0[i];
        double ak = rho / Apr;
        for (int i = 0; i < n; i++) s[i] = r[i] - ak * Ap[i];           /* :904-908 */
        if (p.abs_diff) {                                               /* :910-939 */
            double res = sqrt(orc_dot(s, s, n)) / n;
            if (Pfp && Pfp(inst, m, res, &p, n, t)) { ret = ORC_STOP; goto out; }
            if (res <= p.epsilon) {
                for (int i = 0; i < n; i++) {
                    m[i] += ak * pk[i];
                    if (m[i] != m[i]) { ret = ORC_NAN_VALUE; goto out; }
                }
                ret = ORC_CONVERGENCE; goto out;
            }
            if (p.max_iterations > 0 && t + 1 > p.max_iterations) { ret = ORC_REACHED_MAX_ITERATIONS; break; }
            t++;
        }
        Afp(inst, s, Ax, n);                                            /* :941 */
        double Ass = 0.0, AsAs = 0.0;
        for (int i = 0; i < n; i++) { Ass += Ax[i] * s[i]; AsAs += Ax[i] * Ax[i]; }
        double wk = Ass / AsAs;
        for (int i = 0; i < n; i++) m[i] += ak * pk[i] + wk * s[i];     /* :951-955 */
        m2 = clamp1(orc_dot(m, m, n));
        if (has_nan(m, n)) { ret = ORC_NAN_VALUE; goto out; }
        for (int i = 0; i < n; i++) r[i] = s[i] - wk * Ax[i];           /* :968-972 */
        r2 = orc_dot(r, r, n);
        double rhon = 0.0;
        for (int i = 0; i < n; i++) rhon += r[i] * r0[i];
        if (fabs(rhon) < p.restart_epsilon) {                           /* :982-997 */
            for (int i = 0; i < n; i++) { r0[i] = r[i]; pk[i] = r[i]; }
            rhon = 0.0;
            for (int i = 0; i < n; i++) rhon += r[i] * r0[i];
            rho = rhon;
        } else {
            double bk = (ak / wk) * rhon / rho;                         /* :1001 */
            rho = rhon;
            for (int i = 0; i < n; i++) pk[i] = r[i] + bk * (pk[i] - wk * Ap[i]);
        }
    }
out:
    free(r); free(r0); free(pk); free(Ax); free(s); free(Ap);
    return ret;
}

/* algebra.cpp:50-58 with both bounds inclusive (the defaults the solvers use) */
static double set2box(double low, double hig, double a)
{
    if (a >= hig) return hig;
    if (a <= low) return low;
    return a;
}

/* --------------------------------- projected gradient, Barzilai-Borwein step (lcg.cpp:1054-1204) */
int orc_lpg(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, const double *low,
            const double *hig, int n, const orc_para *param, void *inst)
{
    orc_para p = param ? *param : orc_defaults;
    if (n <= 0) return ORC_INVILAD_VARIABLE_SIZE;                       /* :1060-1067 */
    if (p.max_iterations < 0) return ORC_INVILAD_MAX_ITERATIONS;
    if (p.epsilon <= 0.0) return ORC_INVILAD_EPSILON;
    if (p.step <= 0.0 || p.epsilon >= 1.0) return -1015;                /* LCG_INVALID_LAMBDA */
    if (m == NULL || B == NULL || low == NULL || hig == NULL) return ORC_INVALID_POINTER;
    int ret = 0, t = 0;
    double *g = malloc(sizeof(double) * n), *Ad = malloc(sizeof(double) * n), *mn = malloc(sizeof(double) * n),
           *gn = malloc(sizeof(double) * n), *s = malloc(sizeof(double) * n), *y = malloc(sizeof(double) * n);
    double alpha = p.step;

    for (int i = 0; i < n; i++) m[i] = set2box(low[i], hig[i], m[i]);   /* :1084-1088 */
    Afp(inst, m, Ad, n);
    for (int i = 0; i < n; i++) g[i] = Ad[i] - B[i];
    double m2 = clamp1(orc_dot(m, m, n));
    double g2 = orc_dot(g, g, n);
    if (already_done(&p, Pfp, inst, m, g2, m2, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }

    while (!loop_head(&p, Pfp, inst, m, g2, m2, n, &t, &ret)) {
        for (int i = 0; i < n; i++) mn[i] = set2box(low[i], hig[i], m[i] - alpha * g[i]);   /* :1151-1155 */
        Afp(inst, mn, Ad, n);
        for (int i = 0; i < n; i++) { gn[i] = Ad[i] - B[i]; s[i] = mn[i] - m[i]; y[i] = gn[i] - g[i]; }
        double ss = 0.0, sy = 0.0;
        for (int i = 0; i < n; i++) { ss += s[i] * s[i]; sy += s[i] * y[i]; }                 /* :1168-1175 */
        alpha = ss / sy;
        for (int i = 0; i < n; i++) { m[i] = mn[i]; g[i] = gn[i]; }
        m2 = clamp1(orc_dot(m, m, n));
        g2 = orc_dot(g, g, n);
    }
out:
    free(g); free(Ad); free(mn); free(gn); free(s); free(y);
    return ret;
}

/* --------------------- spectral projected gradient, non-monotone line search (lcg.cpp:1224-1446) */
int orc_lspg(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, const double *low,
             const double *hig, int n, const orc_para *param, void *inst)
{
    orc_para p = param ? *param : orc_defaults;
    if (n <= 0) return ORC_INVILAD_VARIABLE_SIZE;                       /* :1230-1241 */
    if (p.max_iterations < 0) return ORC_INVILAD_MAX_ITERATIONS;
    if (p.epsilon <= 0.0 || p.epsilon >= 1.0) return ORC_INVILAD_EPSILON;
    if (p.step <= 0.0) return -1015;                                    /* LCG_INVALID_LAMBDA */
    if (p.sigma <= 0.0 || p.sigma >= 1.0) return -1014;                 /* LCG_INVALID_SIGMA */
    if (p.beta <= 0.0 || p.beta >= 1.0) return -1013;                   /* LCG_INVALID_BETA */
    if (p.maxi_m <= 0) return -1012;                                    /* LCG_INVALID_MAXIM */
    if (m == NULL || B == NULL || low == NULL || hig == NULL) return ORC_INVALID_POINTER;
    int ret = 0, t = 0;
    double *g = malloc(sizeof(double) * n), *Ad = malloc(sizeof(double) * n), *mn = malloc(sizeof(double) * n),
           *gn = malloc(sizeof(double) * n), *s = malloc(sizeof(double) * n), *y = malloc(sizeof(double) * n),
           *d = malloc(sizeof(double) * n), *qm = malloc(sizeof(double) * p.maxi_m);
    double lambda = p.step, qk = 0.0;

    for (int i = 0; i < n; i++) m[i] = set2box(low[i], hig[i], m[i]);
    Afp(inst, m, Ad, n);
    for (int i = 0; i < n; i++) g[i] = Ad[i] - B[i];
    double m2 = clamp1(orc_dot(m, m, n));
    double g2 = orc_dot(g, g, n);
    if (already_done(&p, Pfp, inst, m, g2, m2, n)) { ret = ORC_ALREADY_OPTIMIZIED; goto out; }
    for (int i = 0; i < n; i++) qk += (0.5 * m[i] * Ad[i] - B[i] * m[i]);   /* :1297-1300 */
    qm[0] = qk;
    for (int i = 1; i < p.maxi_m; i++) qm[i] = -1e+30;

    while (!loop_head(&p, Pfp, inst, m, g2, m2, n, &t, &ret)) {
        for (int i = 0; i < n; i++) d[i] = set2box(low[i], hig[i], m[i] - lambda * g[i]) - m[i];   /* :1339-1343 */
        double ak = 1.0;
        for (int i = 0; i < n; i++) mn[i] = m[i] + ak * d[i];
        Afp(inst, mn, Ad, n);
        qk = 0.0;
        for (int i = 0; i < n; i++) qk += (0.5 * mn[i] * Ad[i] - B[i] * mn[i]);
        double amod = 0.0;
        for (int i = 0; i < n; i++) amod += p.sigma * ak * g[i] * d[i];
        double qmax = qm[0];
        for (int i = 1; i < p.maxi_m; i++) qmax = qmax > qm[i] ? qmax : qm[i];
        while (qk > qmax + amod) {                                      /* :1372-1392 */
            ak = ak * p.beta;
            for (int i = 0; i < n; i++) mn[i] = m[i] + ak * d[i];
            Afp(inst, mn, Ad, n);
            qk = 0.0;
            for (int i = 0; i < n; i++) qk += (0.5 * mn[i] * Ad[i] - B[i] * mn[i]);
            amod = 0.0;
            for (int i = 0; i < n; i++) amod += p.sigma * ak * g[i] * d[i];
        }
        qm[(t + 1) % p.maxi_m] = qk;                                    /* :1394 */
        for (int i = 0; i < n; i++) { gn[i] = Ad[i] - B[i]; s[i] = mn[i] - m[i]; y[i] = gn[i] - g[i]; }
        double ss = 0.0, sy = 0.0;
        for (int i = 0; i < n; i++) { ss += s[i] * s[i]; sy += s[i] * y[i]; }
        lambda = ss / sy;
        for (int i = 0; i < n; i++) { m[i] = mn[i]; g[i] = gn[i]; }
        m2 = clamp1(orc_dot(m, m, n));
        g2 = orc_dot(g, g, n);
    }
out:
    free(g); free(Ad); free(mn); free(gn); free(s); free(y); free(d); free(qm);
    return ret;
}

/* lcg.cpp:59-82: LCG_CG=0, LCG_CGS=2, LCG_BICGSTAB=3, LCG_BICGSTAB2=4; everything else
 * (PCG/PG/SPG handed to lcg_solver) runs CGS. */
int orc_lcg_solver(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B, int n,
                   const orc_para *param, void *inst, int solver_id)
{
    switch (solver_id) {
    case 0: return orc_lcg(Afp, Pfp, m, B, n, param, inst);
    case 3: return orc_lbicgstab(Afp, Pfp, m, B, n, param, inst);
    case 4: return orc_lbicgstab2(Afp, Pfp, m, B, n, param, inst);
    default: return orc_lcgs(Afp, Pfp, m, B, n, param, inst);
    }
}

/* lcg.cpp:121-140: LCG_SPG=6 -> lspg, everything else -> lpg */
int orc_lcg_solver_constrained(orc_axfunc Afp, orc_progress Pfp, double *m, const double *B,
                               const double *low, const double *hig, int n, const orc_para *param,
                               void *inst, int solver_id)
{
    if (solver_id == 6) return orc_lspg(Afp, Pfp, m, B, low, hig, n, param, inst);
    return orc_lpg(Afp, Pfp, m, B, low, hig, n, param, inst);
}
