// solvers_real.hip -- device-resident CG, PCG, CGS and BiCGStab (fp64).
//
// Each solver is liblcg's algorithm (lcg.cpp) re-expressed as a short chain of fused HIP
// passes per iteration.  What is fused and why (bytes are per row of N, 8-byte words):
//
//   CG   (lcg.cpp:206-264)  A.d | d.Ad | m+=a d, g+=a Ad, m.m, g.g, NaN | d = b d - g
//        reference: 13 words + serial NaN scan        here: 2 + 6 + 3 = 11 words, 3 passes
//   PCG  (lcg.cpp:361-423)  A.d | d.Ad | m+=a d, r-=a Ad | M^-1 r | m.m, r.r, z.r, NaN | d = z + b d
//        reference: 18 words                          here: 2 + 6 + 3(M) + 3 + 3 = 17 words
//        (built-in Jacobi: M^-1 r and the three dots ride in the update pass: 12 words)
//   CGS  (lcg.cpp:520-598)  A.p | Ap.r0 | q,w | A.w | m,r update + m.m, r.r, r.r0, NaN | u,p
//   BiCGStab (lcg.cpp:692-781) A.p | Ap.r0 | s | A.s | As.s, As.As | m,r update + dots | p
//
// The scalar recurrences (alpha, beta, omega, the stop rule) run in one-block kernels on
// DevState; see devcommon.hpp / driver.hpp for the mechanics.
#include <functional>

#include "driver.hpp"

namespace lcgh {

// DevState::s slots shared by the real solvers
// (S_AK .. S_G2, clamp1: devcommon.hpp)
// LCG_HIP_CG_AUTO takes the one-reduction arrangements (CG, PCG + built-in Jacobi) on one GPU below this many rows: one launch
// less per iteration against one word per row more -- measured on the 5-point Laplacian: 11.2 vs 13.9 us at 1e5 rows, 30.0 vs 31.4
// at 1e6, 106.9 vs 103.5 at 4e6 (scripts/cg_schedules.py)
constexpr int CG1_AUTO_ROWS = 1 << 20;


// ---- generic passes -------------------------------------------------------------------------
struct OpDot1 {     // acc0 = a.b
    static constexpr int NR = 1, SKIP = SKIP_DONE;
    DevState *st; const double *a, *b;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc) { acc[0] += dotp(ld<T>(a, i), ld<T>(b, i)); }
};
struct OpDot2 {     // acc0 = a.b, acc1 = a.a     (BiCGStab: As.s and As.As, lcg.cpp:735-740)
    static constexpr int NR = 2, SKIP = SKIP_DONE;
    DevState *st; const double *a, *b;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T x = ld<T>(a, i);
        acc[0] += dotp(x, ld<T>(b, i));
        acc[1] += dotp(x, x);
    }
};

// first scalar step of every body: counts the body, then forms alpha = rho / (sum)
struct FinAlpha {
    static constexpr int NR = 1;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        st->it++;
        if (st->done) return;
        st->s[S_AK] = st->s[S_RHO] / sum[0];               // lcg.cpp:235,390,553,725
    }
};
struct FinOmega {   // BiCGStab: omega = As.s / As.As (lcg.cpp:741)
    static constexpr int NR = 2;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (st->done) return;
        st->s[S_WK] = sum[0] / sum[1];
    }
};

// setup scalar step: |m|^2 (clamped), residual numerator, rho; decides "already optimised"
// (lcg.cpp:178-203: in abs_diff mode BOTH criteria are tried, in this order)
struct FinInit {
    static constexpr int NR = 3;    // m.m, g.g, rho
    __device__ void operator()(DevState *st, const double *sum) const
    {
        const double m2 = clamp1(sum[0]), g2 = sum[1];
        st->s[S_M2] = m2; st->s[S_G2] = g2; st->s[S_RHO] = sum[2];
        double r;
        bool already = false;
        if (st->abs_diff && sqrt(g2) / st->n_global <= st->eps) { r = sqrt(g2) / st->n_global; already = true; }
        else if (g2 / m2 <= st->eps) { r = g2 / m2; already = true; }
        else r = st->abs_diff ? sqrt(g2) / st->n_global : g2 / m2;
        st->residual = r;
        if (already) { st->done = 1; st->status = ST_ALREADY; }
        publish(st);
    }
};

// closing scalar step of a body.  sums: m.m, g2 (r.r or g.g), rho_new, NaN count.
// BICG selects the BiCGStab beta (lcg.cpp:773) instead of rho_new/rho (lcg.cpp:256,415,589).
template <bool BICG>
struct FinClose {
    static constexpr int NR = 4;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (!st->done) {
            st->s[S_M2] = clamp1(sum[0]);
            if (sum[3] > 0.0 || sum[0] != sum[0]) {         // lcg.cpp:247-253
                st->t++;
                st->done = 1; st->status = ST_NAN;
            } else {
                const double rho_new = sum[2];
                st->s[S_BK] = BICG ? (st->s[S_AK] / st->s[S_WK]) * rho_new / st->s[S_RHO]
                                   : rho_new / st->s[S_RHO];
                st->s[S_RHO] = rho_new;
                st->s[S_G2] = sum[1];
                st->t++;
                stop_rule(st, sum[1], st->s[S_M2]);
            }
        }
        publish(st);
    }
};

// ---- CG -------------------------------------------------------------------------------------
// The reference multiplies the initial guess before anything else (lcg.cpp:168, 314, 476, 648).  The usual guess is all zeros, and
// then so is that product: whether the guess is all zeros is decided ON THE DEVICE (every rank's count of non-zeros, summed like any
// other sum of the loop, so all ranks decide alike), the product falls through on that flag the way every kernel falls through on the
// stop flag, and the pass that consumes A.m reads zeros instead.  Same numbers as the product would have left (a sum of +-0 products).
struct OpZeroProbe {    // how many components of the guess are not zero (NaN counts)
    static constexpr int NR = 1, SKIP = SKIP_NEVER;
    DevState *st; const double *m;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc) { acc[0] += nonzeros(ld<T>(m, i)); }
    static __device__ double nonzeros(double v) { return v != 0.0 ? 1.0 : 0.0; }
    static __device__ double nonzeros(double2 v) { return (v.x != 0.0 ? 1.0 : 0.0) + (v.y != 0.0 ? 1.0 : 0.0); }
};
struct FinZeroGuess {
    static constexpr int NR = 1;
    __device__ void operator()(DevState *st, const double *sum) const { st->zero_guess = sum[0] == 0.0 ? 1 : 0; }
};

struct OpCgInit {   // g = Ad - B; d = -g; m.m, g.g (x2: rho = g.g)      lcg.cpp:171-183
    static constexpr int NR = 3, SKIP = SKIP_NEVER;
    DevState *st; const double *Ad, *B, *m; double *g, *d;
    double *clear = nullptr;    // = Ad where the loop continues A.d by recurrence (one-reduction schedule): a product that was not made leaves no zeros behind
    bool zg = false;
    __device__ void prep() { zg = st->zero_guess != 0; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T av = ldz<T>(Ad, i, zg);
        if (zg && clear) st_(clear, i, av);
        const T gv = vsub(av, ld<T>(B, i));
        const T mv = ldz<T>(m, i, zg);         // (an all-zero guess is not read a second time)
        st_(g, i, gv); st_(d, i, vneg(gv));
        acc[0] += dotp(mv, mv);
        const double gg = dotp(gv, gv);
        acc[1] += gg; acc[2] += gg;
    }
};
struct OpCgUpdate { // m += a d; g += a Ad; m.m, g.g (x2), NaN           lcg.cpp:237-255
    static constexpr int NR = 4, SKIP = SKIP_DONE;
    DevState *st; double *m, *g; const double *d, *Ad; double ak; bool nt = false;
    __device__ void prep() { ak = st->s[S_AK]; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        // (nt: the iterate and A.d are touched once per iteration -- streamed past the caches, so that d and g, which the next two
        //  kernels read, stay: stream_vectors below)
        const T mv = vadd(ldp<T>(m, i, nt), ak * ld<T>(d, i));
        const T gv = vadd(ld<T>(g, i), ak * ldp<T>(Ad, i, nt));
        stp(m, i, mv, nt); st_(g, i, gv);
        acc[0] += dotp(mv, mv);
        const double gg = dotp(gv, gv);
        acc[1] += gg; acc[2] += gg;
        acc[3] += nanflag(mv);
    }
};
struct OpCgDir {    // d = b d - g                                         lcg.cpp:259-263
    static constexpr int NR = 0, SKIP = SKIP_DIR;
    DevState *st; double *d; const double *g; double bk; bool nt = false;
    __device__ void prep() { bk = st->s[S_BK]; }
    template <class T> __device__ void apply(long i, double *)
    {
        st_(d, i, vsub(bk * ld<T>(d, i), ldp<T>(g, i, nt)));      // (g: its last use before the product; d: the product's x)
    }
};

// ---- CG, one reduction per iteration (Chronopoulos-Gear schedule of lcg.cpp:206-264) ----------
// The classic body needs d.Ad before the update and g.g after it: two reductions, two RCCL
// all-reduces per iteration once the rows are sharded.  Here A is applied to the GRADIENT,
// w = A.g, and A.d is carried by the same recurrence as d (d = b d - g  =>  Ad = b Ad - w), so
//     d.Ad = g.w - b g.g / a_prev        (g_new . A d_old = g.g / a_prev,  d_old.Ad_old = rho / a_prev)
// and g.g, g.w, m.m, NaN ride in ONE reduction.  Per row: 9 words (update, with m.m, g.g, NaN) + 2 (A.g) + 2 (g.w).
struct OpCg1UpdateSums { // d = b d - g; Ad = b Ad - w; m += a d; g += a Ad (lcg.cpp:259-263, 237-243), leaving m.m, g.g and the NaN
                         // count behind (lcg.cpp:244-255: sums 0, 1, 3); g.w -- sum 2 -- comes with the product or from the pass after it
    static constexpr int NR = 4, SKIP = SKIP_DONE;
    DevState *st; double *m, *g, *d, *Ad; const double *w; double ak, bk; bool nt = false;
    __device__ void prep() { ak = st->s[S_AK]; bk = st->s[S_BK]; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        // (nt: everything but g -- the product's x -- is next touched by this pass one product later)
        const T g0 = ld<T>(g, i);
        const T dv = vsub(bk * ldp<T>(d, i, nt), g0);
        const T sv = vsub(bk * ldp<T>(Ad, i, nt), ldp<T>(w, i, nt));
        stp(d, i, dv, nt); stp(Ad, i, sv, nt);
        const T mv = vadd(ldp<T>(m, i, nt), ak * dv);
        const T gv = vadd(g0, ak * sv);
        stp(m, i, mv, nt); st_(g, i, gv);
        acc[0] += dotp(mv, mv);
        acc[1] += dotp(gv, gv);
        acc[3] += nanflag(mv);
    }
};
struct FinCg1Start {    // a_0 = g.g / g.A.g, b_0 = 0 (d_0 = -g, lcg.cpp:171-176)
    static constexpr int NR = 1;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (st->done) return;
        st->s[S_AK] = st->s[S_RHO] / sum[0];
        st->s[S_BK] = 0.0;
    }
};
// (FinCg1Close -- the only scalar step of a body -- lives in devcommon.hpp: a sharded product's last block may run it)

// ---- PCG ------------------------------------------------------------------------------------
struct OpResidual { // r = B - Ax   (also the start of CGS/BiCGStab with extra copies)
    static constexpr int NR = 0, SKIP = SKIP_NEVER;
    DevState *st; const double *Ax, *B; double *r;
    double *clear = nullptr;    // (as in OpCgInit)
    bool zg = false;
    __device__ void prep() { zg = st->zero_guess != 0; }
    template <class T> __device__ void apply(long i, double *)
    {
        const T av = ldz<T>(Ax, i, zg);
        if (zg && clear) st_(clear, i, av);
        st_(r, i, vsub(ld<T>(B, i), av));
    }
};
struct OpPcgInit2 { // d = z; m.m, r.r, z.r                               lcg.cpp:325-339
    static constexpr int NR = 3, SKIP = SKIP_NEVER;
    DevState *st; const double *z, *m, *r; double *d;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T zv = ld<T>(z, i), mv = ld<T>(m, i), rv = ld<T>(r, i);
        st_(d, i, zv);
        acc[0] += dotp(mv, mv); acc[1] += dotp(rv, rv); acc[2] += dotp(zv, rv);
    }
};
struct OpPcgUpdate {    // m += a d; r -= a Ad                            lcg.cpp:392-397
    static constexpr int NR = 0, SKIP = SKIP_DONE;
    DevState *st; double *m, *r; const double *d, *Ad; double ak;
    __device__ void prep() { ak = st->s[S_AK]; }
    template <class T> __device__ void apply(long i, double *)
    {
        st_(m, i, vadd(ld<T>(m, i), ak * ld<T>(d, i)));
        st_(r, i, vsub(ld<T>(r, i), ak * ld<T>(Ad, i)));
    }
};
struct OpPcgDots {      // m.m, r.r, z.r, NaN                             lcg.cpp:401-414
    static constexpr int NR = 4, SKIP = SKIP_DONE;
    DevState *st; const double *m, *r, *z;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T mv = ld<T>(m, i), rv = ld<T>(r, i), zv = ld<T>(z, i);
        acc[0] += dotp(mv, mv); acc[1] += dotp(rv, rv); acc[2] += dotp(zv, rv); acc[3] += nanflag(mv);
    }
};
struct OpPcgUpdateJacobi {  // built-in Jacobi: update, z = r .* invdiag and all dots in one pass
    static constexpr int NR = 4, SKIP = SKIP_DONE;
    DevState *st; double *m, *r, *z; const double *d, *Ad, *invdiag; double ak; bool nt = false;
    __device__ void prep() { ak = st->s[S_AK]; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        // (nt: as in OpCgUpdate -- the iterate, A.d and the diagonal are touched once per iteration; z and d are read by the next kernel)
        const T mv = vadd(ldp<T>(m, i, nt), ak * ld<T>(d, i));
        const T rv = vsub(ld<T>(r, i), ak * ldp<T>(Ad, i, nt));
        const T zv = vmul(ldp<T>(invdiag, i, nt), rv);
        stp(m, i, mv, nt); st_(r, i, rv); st_(z, i, zv);
        acc[0] += dotp(mv, mv); acc[1] += dotp(rv, rv); acc[2] += dotp(zv, rv); acc[3] += nanflag(mv);
    }
};
struct OpPcgDir {       // d = z + b d                                    lcg.cpp:418-422
    static constexpr int NR = 0, SKIP = SKIP_DIR;
    DevState *st; double *d; const double *z; double bk; bool nt = false;
    __device__ void prep() { bk = st->s[S_BK]; }
    template <class T> __device__ void apply(long i, double *) { st_(d, i, vadd(ldp<T>(z, i, nt), bk * ld<T>(d, i))); }
};

// PCG with the built-in Jacobi in the one-reduction (Chronopoulos-Gear) arrangement -- what plain CG's OpCg1UpdateSums is to
// lcg.cpp:206-264, this is to lpcg (lcg.cpp:361-423): with u = M^-1 r and w = A u,
//     d = u + b d;  Ad = w + b Ad;  m += a d;  r -= a Ad;  u = invdiag r          (one pass, leaving m.m, r.r, u.r and the NaN count)
//     w = A u  carrying  w.u                                                      (the product)
//     b' = u.r / (u.r)_old;  a' = u.r / (w.u - b' u.r / a)                         (the one scalar step)
// Same iterates as lpcg in exact arithmetic (d.Ad = w.u - b u.r / a_old), same stop rule on r.r / m.m, one reduction per iteration
// and two launches instead of three.  Per row 12 words, like the fused classic form.
struct OpPcg1UpdateSums {
    static constexpr int NR = 5, SKIP = SKIP_DONE;      // sums 0 m.m, 1 r.r, 2 u.r, 3 NaN; 4 = w.u comes with the product
    DevState *st; double *m, *r, *z, *d, *Ad; const double *w, *invdiag; double ak, bk;
    __device__ void prep() { ak = st->s[S_AK]; bk = st->s[S_BK]; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T dv = vadd(ld<T>(z, i), bk * ld<T>(d, i));
        const T sv = vadd(ld<T>(w, i), bk * ld<T>(Ad, i));
        st_(d, i, dv); st_(Ad, i, sv);
        const T mv = vadd(ld<T>(m, i), ak * dv);
        const T rv = vsub(ld<T>(r, i), ak * sv);
        const T zv = vmul(ld<T>(invdiag, i), rv);
        st_(m, i, mv); st_(r, i, rv); st_(z, i, zv);
        acc[0] += dotp(mv, mv); acc[1] += dotp(rv, rv); acc[2] += dotp(zv, rv); acc[3] += nanflag(mv);
    }
};
struct FinPcg1Close {   // the only scalar step of a body (FinCg1Close with rho = u.r and the stop rule on r.r)
    static constexpr int NR = 5;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        st->it++;
        if (!st->done) {
            st->s[S_M2] = clamp1(sum[0]);
            if (sum[3] > 0.0 || sum[0] != sum[0]) {         // lcg.cpp:404-410
                st->t++;
                st->done = 1; st->status = ST_NAN;
            } else {
                const double rho_new = sum[2];
                const double bk = rho_new / st->s[S_RHO];                       // lcg.cpp:415
                st->s[S_AK] = rho_new / (sum[4] - bk * rho_new / st->s[S_AK]);  // lcg.cpp:390 with d.Ad as above
                st->s[S_BK] = bk;
                st->s[S_RHO] = rho_new;
                st->s[S_G2] = sum[1];
                st->t++;
                stop_rule(st, sum[1], st->s[S_M2]);
            }
        }
        publish(st);
    }
};

// ---- CGS / BiCGStab ---------------------------------------------------------------------------
template <bool WITH_U>
struct OpShadowInit {   // p = (u =) r0 = r = B - Ax; m.m, r.r, r.r0      lcg.cpp:480-497 / 650-667
    static constexpr int NR = 3, SKIP = SKIP_NEVER;
    DevState *st; const double *Ax, *B, *m; double *r, *r0, *p, *u; bool zg = false;
    __device__ void prep() { zg = st->zero_guess != 0; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T rv = vsub(ld<T>(B, i), ldz<T>(Ax, i, zg));
        const T mv = ldz<T>(m, i, zg);         // (an all-zero guess is not read a second time)
        st_(r, i, rv); st_(r0, i, rv); st_(p, i, rv);
        if (WITH_U) st_(u, i, rv);
        acc[0] += dotp(mv, mv);
        const double rr = dotp(rv, rv);
        acc[1] += rr; acc[2] += rr;
    }
};
struct OpCgsQW {        // q = u - a Ax; w = u + q                        lcg.cpp:556-560
    static constexpr int NR = 0, SKIP = SKIP_DONE;
    DevState *st; const double *u, *Ax; double *q, *w; double ak; bool nt = false;
    __device__ void prep() { ak = st->s[S_AK]; }
    template <class T> __device__ void apply(long i, double *)
    {
        const T uv = ld<T>(u, i);
        const T qv = vsub(uv, ak * ld<T>(Ax, i));
        st_(q, i, qv); st_(w, i, vadd(uv, qv));
    }
};
struct OpCgsUpdate {    // m += a w; r -= a Ax; m.m, r.r, r.r0, NaN       lcg.cpp:565-588
    static constexpr int NR = 4, SKIP = SKIP_DONE;
    DevState *st; double *m, *r; const double *w, *Ax, *r0; double ak; bool nt = false;
    __device__ void prep() { ak = st->s[S_AK]; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T mv = vadd(ldp<T>(m, i, nt), ak * ld<T>(w, i));       // (nt: the iterate only -- see stream_vectors)
        const T rv = vsub(ld<T>(r, i), ak * ld<T>(Ax, i));
        stp(m, i, mv, nt); st_(r, i, rv);
        acc[0] += dotp(mv, mv); acc[1] += dotp(rv, rv); acc[2] += dotp(rv, ld<T>(r0, i)); acc[3] += nanflag(mv);
    }
};
struct OpCgsDir {       // u = r + b q; p = u + b (q + b p)               lcg.cpp:593-597
    static constexpr int NR = 0, SKIP = SKIP_DIR;
    DevState *st; double *u, *p; const double *r, *q; double bk; bool nt = false;
    __device__ void prep() { bk = st->s[S_BK]; }
    template <class T> __device__ void apply(long i, double *)
    {
        const T qv = ld<T>(q, i);
        const T uv = vadd(ld<T>(r, i), bk * qv);
        st_(u, i, uv);
        st_(p, i, vadd(uv, bk * vadd(qv, bk * ld<T>(p, i))));
    }
};
struct OpBicgS {        // s = r - a Ap                                   lcg.cpp:727-731
    static constexpr int NR = 0, SKIP = SKIP_DONE;
    DevState *st; const double *r, *Ap; double *s; double ak; bool nt = false;
    __device__ void prep() { ak = st->s[S_AK]; }
    template <class T> __device__ void apply(long i, double *) { st_(s, i, vsub(ld<T>(r, i), ak * ld<T>(Ap, i))); }
};
struct OpBicgUpdate {   // m += a p + w s; r = s - w Ax; m.m, r.r, r.r0, NaN   lcg.cpp:743-772
    static constexpr int NR = 4, SKIP = SKIP_DONE;
    DevState *st; double *m, *r; const double *p, *s, *Ax, *r0; double ak, wk; bool nt = false;
    __device__ void prep() { ak = st->s[S_AK]; wk = st->s[S_WK]; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T sv = ld<T>(s, i);
        const T mv = vadd(ldp<T>(m, i, nt), vadd(ak * ld<T>(p, i), wk * sv));       // (nt: the iterate only -- see stream_vectors)
        const T rv = vsub(sv, wk * ld<T>(Ax, i));
        stp(m, i, mv, nt); st_(r, i, rv);
        acc[0] += dotp(mv, mv); acc[1] += dotp(rv, rv); acc[2] += dotp(rv, ld<T>(r0, i)); acc[3] += nanflag(mv);
    }
};
struct OpBicgDir {      // p = r + b (p - w Ap)                           lcg.cpp:776-780
    static constexpr int NR = 0, SKIP = SKIP_DIR;
    DevState *st; double *p; const double *r, *Ap; double bk, wk; bool nt = false;
    __device__ void prep() { bk = st->s[S_BK]; wk = st->s[S_WK]; }
    template <class T> __device__ void apply(long i, double *)
    {
        st_(p, i, vadd(ld<T>(r, i), bk * vsub(ld<T>(p, i), wk * ld<T>(Ap, i))));
    }
};

// ---- host drivers -------------------------------------------------------------------------------
static inline uintptr_t al(const void *p) { return (uintptr_t)p; }

// Cache policy of the vector passes.  On a system whose product streams far more than the 256 MB Infinity Cache holds, the passes
// read and write what the NEXT kernels do not need (the iterate, A.d after the update, g in the direction pass, the Jacobi diagonal)
// with non-temporal accesses, so that what they do need -- the product's x above all -- is still cached when they run.  Same-box A/B
// (LCG_HIP_NT_VECTORS=0 / 1 forces either; 10M rows): headline CG +1.4..3 % (six pairs, two boxes), row-random band +2 %, 27-point
// stencil +1.5 %, BiCGStab / CGS (iterate only: with more the passes got slower) +1 %, 5-point Laplacian CG +1..2 %, PCG + Jacobi +3..5 %.
// The effect is NOT monotonic in the size (5-point Laplacians: 1M..3M rows -2..-5 %, 4M +6..13 % -- the matrix then fits the cache
// beside two vectors --, 6M classic CG -4 %), so the automatic rule only takes the regime that was measured to be safe: vectors that
// cannot stay cached (five of them > 128 MB) AND a product that streams >= 512 MB.
static bool stream_vectors(long n, lcg_axfunc_ptr Afp, void *inst)
{
    static const int env = [] { const char *e = std::getenv("LCG_HIP_NT_VECTORS"); return e ? atoi(e) : -1; }();
    if (env >= 0) return env != 0;
    if (n * 40L <= (128L << 20)) return false;
    if (Afp == lcg_hip_csr_ax && inst != nullptr) {
        const lcg_hip_csr *A = static_cast<const lcg_hip_csr *>(inst);
        const long nnz = A->distributed ? (long)A->loc.nnz + (long)A->rem.nnz : (long)A->main.nnz;
        return nnz * 12L >= (512L << 20);
    }
    return n * 40L > (256L << 20);       // a product of the caller's own: nothing is known about its stream
}

struct RealCommon {
    Ctx &c; Driver drv; lcg_para para; void *inst; lcg_axfunc_ptr Afp; lcg_progress_ptr Pfp;
    double *m; int n;
    RealCommon(Ctx &c_, int n_, const lcg_para &p, void *inst_, lcg_axfunc_ptr A, lcg_progress_ptr P, double *m_)
        : c(c_), drv(c_, n_, false, p.max_iterations, p.epsilon, p.abs_diff), para(p), inst(inst_),
          Afp(A), Pfp(P), m(m_), n(n_) { drv.user_cb = A != lcg_hip_csr_ax; }
    int ax(const double *x, double *y) { return drv.timed_ax([&] { Afp(inst, x, y, n); }); }
    // y = A.m for the initial guess (lcg.cpp:168, 314, 476, 648).  With the built-in product: the zero-guess probe in front (above),
    // the product honouring its verdict, and no event pair around a launch that may be empty (lcg_hip_last_ax_mean_us is about products).
    int ax_setup(const double *m0, double *y)
    {
        if (Afp != lcg_hip_csr_ax || inst == nullptr || !zero_guess_probe()) return ax(m0, y);
        int rc = drv.vec(OpZeroProbe{c.state, m0}, (uintptr_t)m0);
        if (!rc) rc = drv.scal(FinZeroGuess{});
        if (rc) return rc;
        c.ax_skip = &c.state->zero_guess;
        Afp(inst, m0, y, n);
        c.ax_skip = nullptr;
        return c.ax_rc;
    }
    static bool zero_guess_probe()
    {
        static const bool on = [] { const char *e = lab_env("LCG_HIP_ZERO_GUESS"); return !e || atoi(e) != 0; }();     // (A/B runs)
        return on;
    }
    // y = A.x followed by the sums y.u (and y.y): with the built-in product on a handle this process holds whole, the sums ride
    // in the product's epilogue (csr.hip: k_spmv_lds1d) and reach the next scalar step as its sum `row` (y.y: row + 1) -- *fused
    // says so; otherwise the product is made as always and the caller runs its own reducing pass
    int ax_dot(const double *x, double *y, const double *u, bool yy, int row, bool *fused)
    {
        int f = 0, slots = 0;
        // (sharded rows: csr_ax_dot hands over to comm.hip, which always makes the product and answers 1 when it carried the sum, 2 when not)
        const bool builtin = Afp == lcg_hip_csr_ax && inst != nullptr;
        int rc = drv.timed_ax([&] {
            if (builtin) f = csr_ax_dot(static_cast<lcg_hip_csr *>(inst), x, y, u, yy ? 1 : 0, c.ax_partials, &slots, c.stream, &c.state->done);
            if (f == 0) Afp(inst, x, y, n);
            else if (f < 0 && !c.ax_rc) c.ax_rc = f;
        });
        *fused = f == 1;
        if (f == 1) {
            PartCount &pc = drv.pcnt;
            pc.axp = c.ax_partials; pc.ax_n = slots; pc.ax_row = row; pc.ax_yy = yy ? 1 : 0;
            pc.g[row] = 0; if (yy) pc.g[row + 1] = 0;
        }
        return rc;
    }
    int run_loop(const std::function<int()> &body)
    {
        auto pfp = [&](double resid, int t) -> int { return Pfp(inst, m, resid, &para, n, t); };
        return drv.run(body, Pfp != nullptr, pfp, LCG_REACHED_MAX_ITERATIONS, LCG_NAN_VALUE);
    }
};

static int check_args(const lcg_para &p, int n, const double *m, const double *B)
{   // lcg.cpp:150-155
    if (n <= 0) return LCG_INVILAD_VARIABLE_SIZE;
    if (p.max_iterations < 0) return LCG_INVILAD_MAX_ITERATIONS;
    if (p.epsilon <= 0.0 || p.epsilon >= 1.0) return LCG_INVILAD_EPSILON;
    if (m == nullptr || B == nullptr) return LCG_INVALID_POINTER;
    return 0;
}

double global_rows(Ctx &c, int n);
double global_rows_of(Ctx &c, int n, const void *afp, const void *inst);   // comm.hip   // comm.hip: n summed over ranks (n itself when single)

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

static int solve_cg(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, int n,
                    const lcg_para *param, void *inst, double *Gk, double *Dk, double *ADk, int mem)
{
    const lcg_para p = param ? *param : lcg_hip_default_parameters();
    TRY(check_args(p, n, m, B));
    TRY(ensure_init());
    Ctx &c = ctx();
    HostBridge hb; TRY(hb.open(mem, m, B, sizeof(double) * n, c.stream));
    Workspace ws; double *g, *d, *Ad;
    TRY(ws.get(g, Gk, sizeof(double) * n)); TRY(ws.get(d, Dk, sizeof(double) * n)); TRY(ws.get(Ad, ADk, sizeof(double) * n));
    // AUTO: the one-reduction schedule when the rows are sharded (one all-reduce per iteration) and on one GPU for systems so
    // small that a launch costs more than a word per row (< 2^20 rows): two launches per iteration instead of three
    // (a callback of the caller's own keeps the reference's recurrence and with it the reference's sequence of callback calls:
    //  the one-reduction arrangement makes one product more before the first stop test)
    const bool one_reduction = c.cg_schedule == LCG_HIP_CG_ONE_REDUCTION ||
                               (c.cg_schedule == LCG_HIP_CG_AUTO && (comm_active() || (n < CG1_AUTO_ROWS && Afp == lcg_hip_csr_ax)));
    double *w = nullptr;
    if (one_reduction) TRY(ws.get(w, nullptr, sizeof(double) * n));
    // roles by weight (driver.hpp: Placement): what the loop's product writes (A.d; w = A.g in the one-reduction arrangement), what it
    // reads, the rest
    if (one_reduction) TRY(Placement::run(c, n, (const void *)Afp, inst, B, ws, {&w, &g, &d, &Ad}, 1));
    else TRY(Placement::run(c, n, (const void *)Afp, inst, B, ws, {&Ad, &d, &g}, 1));
    RealCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    const bool nt = stream_vectors(n, Afp, inst);      // (n: the rows THIS process holds)

    TRY(k.ax_setup(m, Ad));                                                      // lcg.cpp:168
    TRY(k.drv.vec(OpCgInit{st, Ad, B, m, g, d, one_reduction ? Ad : nullptr}, al(Ad) | al(B) | al(m) | al(g) | al(d)));
    TRY(k.drv.scal(FinInit{}));
    const uintptr_t a_upd = al(m) | al(g) | al(d) | al(Ad);
    if (one_reduction) {
        bool fused; TRY(k.ax_dot(g, w, g, false, 0, &fused));
        if (!fused) TRY(k.drv.vec(OpDot1{st, g, w}, al(g) | al(w)));
        int rc;
        // The product that closes a body only serves the NEXT body's step length: the body the iteration cap ends the solve with goes
        // without it (K iterations = K + 1 products, as in the reference's loop: lcg.cpp:168, 232).
        // (Its closing step then finds NO partial sums for g.w -- neither the previous body's nor an unwritten table row: the count of
        //  that row is zero, the sum exactly 0, and the a_k, b_k the step leaves behind are a fixed function of the state; nothing
        //  reads them after the cap.)
        auto last_body = [&]() {
            const bool last = p.max_iterations > 0 && k.drv.enq + 1 >= p.max_iterations;
            if (last) { k.drv.pcnt.ax_n = 0; k.drv.pcnt.ax_row = -1; k.drv.pcnt.g[2] = 0; }
            return last;
        };
        if (fused && Pfp == nullptr && !comm_active()) {
            // The product carries g.w, the update pass the other sums of the body: a body is TWO launches, `[step] update + sums |
            // A.g + g.w`; the scalar step that closes body k rides in the update of body k+1, the last one is closed by the tail.
            bool first = true;
            k.drv.tail = [&]() -> int { return first ? k.drv.scal(FinCg1Start{}) : k.drv.scal(FinCg1Close{}); };
            rc = k.run_loop([&]() -> int {
                if (first) { TRY(k.drv.vecf(FinCg1Start{}, OpCg1UpdateSums{st, m, g, d, Ad, w, 0.0, 0.0, nt}, a_upd | al(w))); first = false; }
                else TRY(k.drv.vecf(FinCg1Close{}, OpCg1UpdateSums{st, m, g, d, Ad, w, 0.0, 0.0, nt}, a_upd | al(w)));
                if (last_body()) return 0;
                bool f; TRY(k.ax_dot(g, w, g, false, 2, &f));
                if (!f) { c.err = "A.x stopped carrying its dot in the middle of a solve"; return LCG_HIP_E_ARG; }
                return 0;
            });
            k.drv.tail = nullptr;
        } else if (fused) {
            // with a progress callback the state is read after every body: the same passes with the scalar step as its own
            // kernel at the end of the body (the same arithmetic: bit-identical iterates).  Sharded rows take this form too: the step
            // is where the ranks' sums meet (update + m.m, g.g, NaN | product, pushes, remote part + g.w | reduce, exchange, step)
            TRY(k.drv.scal(FinCg1Start{}));
            rc = k.run_loop([&]() -> int {
                TRY(k.drv.vec(OpCg1UpdateSums{st, m, g, d, Ad, w, 0.0, 0.0, nt}, a_upd | al(w)));
                if (!last_body()) {
                    bool f; TRY(k.ax_dot(g, w, g, false, 2, &f));
                    if (!f) { c.err = "A.x stopped carrying its dot in the middle of a solve"; return LCG_HIP_E_ARG; }
                }
                TRY(k.drv.scal(FinCg1Close{}));
                return 0;
            });
        } else if (Pfp == nullptr && !comm_active()) {
            // One GPU, no progress callback: the scalar step that closes body k rides in the first pass of body k+1
            // (FinCg1Start in front of the first one), so a body is three launches: update | A.g | dots.  The last body is
            // closed by the tail.  Same arithmetic in the same order as the four-launch form below.
            bool first = true;
            k.drv.tail = [&]() -> int { return first ? k.drv.scal(FinCg1Start{}) : k.drv.scal(FinCg1Close{}); };
            rc = k.run_loop([&]() -> int {
                if (first) { TRY(k.drv.vecf(FinCg1Start{}, OpCg1UpdateSums{st, m, g, d, Ad, w, 0.0, 0.0, nt}, a_upd | al(w))); first = false; }
                else TRY(k.drv.vecf(FinCg1Close{}, OpCg1UpdateSums{st, m, g, d, Ad, w, 0.0, 0.0, nt}, a_upd | al(w)));
                if (last_body()) return 0;
                TRY(k.ax(g, w));
                TRY(k.drv.vec_rows(OpDot1{st, g, w}, 2, al(g) | al(w)));
                return 0;
            });
            k.drv.tail = nullptr;
        } else {
            // sharded rows (and callbacks that are not the built-in product): m.m, g.g and the NaN count ride in the update pass,
            // which has m and g in registers anyway; the pass after the product only takes g.w (two words per row instead of three)
            TRY(k.drv.scal(FinCg1Start{}));
            rc = k.run_loop([&]() -> int {
                TRY(k.drv.vec(OpCg1UpdateSums{st, m, g, d, Ad, w, 0.0, 0.0, nt}, a_upd | al(w)));
                if (!last_body()) {
                    bool f; TRY(k.ax_dot(g, w, g, false, 2, &f));   // (a product that cannot carry the sum is made all the same)
                    if (!f) TRY(k.drv.vec_rows(OpDot1{st, g, w}, 2, al(g) | al(w)));
                }
                TRY(k.drv.scal(FinCg1Close{}));
                return 0;
            });
        }
        int rc2 = hb.close(c.stream);
        return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
    }
    int rc = k.run_loop([&]() -> int {
        bool f; TRY(k.ax_dot(d, Ad, d, false, 0, &f));                          // :232
        if (!f) TRY(k.drv.vec(OpDot1{st, d, Ad}, al(d) | al(Ad)));               // :234
        TRY(k.drv.vecf(FinAlpha{}, OpCgUpdate{st, m, g, d, Ad, 0.0, nt}, a_upd));    // :235, :237-255
        TRY(k.drv.vecf(FinClose<false>{}, OpCgDir{st, d, g, 0.0, nt}, al(d) | al(g)));   // :244-257, :259-263
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

static int solve_pcg(lcg_axfunc_ptr Afp, lcg_axfunc_ptr Mfp, lcg_progress_ptr Pfp, double *m, const double *B,
                     int n, const lcg_para *param, void *inst, int mem)
{
    const lcg_para p = param ? *param : lcg_hip_default_parameters();
    TRY(check_args(p, n, m, B));
    if (Mfp == nullptr) return LCG_NULL_PRECONDITION_MATRIX;
    TRY(ensure_init());
    Ctx &c = ctx();
    HostBridge hb; TRY(hb.open(mem, m, B, sizeof(double) * n, c.stream));
    Workspace ws; double *r, *z, *d, *Ad;
    TRY(ws.get(r, nullptr, sizeof(double) * n)); TRY(ws.get(z, nullptr, sizeof(double) * n));
    TRY(ws.get(d, nullptr, sizeof(double) * n)); TRY(ws.get(Ad, nullptr, sizeof(double) * n));
    TRY(Placement::run(c, n, (const void *)Afp, inst, B, ws, {&Ad, &d, &r, &z}, 1));      // (roles by weight: A.d, the product's x, the rest)
    RealCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    const bool nt = stream_vectors(n, Afp, inst);      // (n: the rows THIS process holds)
    // built-in Jacobi on a handle that owns its reciprocal diagonal: fold M^-1 into the update
    const double *invdiag = nullptr;
    if (Mfp != lcg_hip_jacobi_mx) k.drv.user_cb = true;
    if (Mfp == lcg_hip_jacobi_mx && inst) {
        const lcg_hip_csr *A = static_cast<const lcg_hip_csr *>(inst);
        if (!A->is_complex && A->n_rows == n) invdiag = A->invdiag;
    }

    TRY(k.ax_setup(m, Ad));                                                      // lcg.cpp:314
    TRY(k.drv.vec(OpResidual{st, Ad, B, r, Ad}, al(Ad) | al(B) | al(r)));        // :317-321
    TRY(k.drv.checked_mx([&] { Mfp(inst, r, z, n); }));                          // :323
    TRY(k.drv.vec(OpPcgInit2{st, z, m, r, d}, al(z) | al(m) | al(r) | al(d)));   // :325-339
    TRY(k.drv.scal(FinInit{}));
    const uintptr_t a_all = al(m) | al(r) | al(z) | al(d) | al(Ad) | al(invdiag);
    const bool one_reduction = invdiag != nullptr &&
                               (c.cg_schedule == LCG_HIP_CG_ONE_REDUCTION ||
                                (c.cg_schedule == LCG_HIP_CG_AUTO && (comm_active() || (n < CG1_AUTO_ROWS && !k.drv.user_cb))));
    if (one_reduction) {
        double *w; TRY(ws.get(w, nullptr, sizeof(double) * n));
        {   // w = A u_0 and w.u_0 (sum 0): a_0 = u.r / w.u, b_0 = 0
            bool f; TRY(k.ax_dot(z, w, z, false, 0, &f));
            if (!f) TRY(k.drv.vec(OpDot1{st, z, w}, al(z) | al(w)));
        }
        auto product = [&]() -> int {       // w = A u carrying w.u as sum 4 (or a pass of its own where the product cannot)
            bool f; TRY(k.ax_dot(z, w, z, false, 4, &f));
            if (!f) TRY(k.drv.vec_rows(OpDot1{st, z, w}, 4, al(z) | al(w)));
            return 0;
        };
        auto last_body = [&]() {       // (as in solve_cg: the capped body's closing step finds an empty row for w.u, not a stale one)
            const bool last = p.max_iterations > 0 && k.drv.enq + 1 >= p.max_iterations;
            if (last) { k.drv.pcnt.ax_n = 0; k.drv.pcnt.ax_row = -1; k.drv.pcnt.g[4] = 0; }
            return last;
        };
        const OpPcg1UpdateSums upd{st, m, r, z, d, Ad, w, invdiag, 0.0, 0.0};
        int rc;
        if (Pfp == nullptr && !comm_active()) {
            // the step that closes body k rides in the update of body k + 1 (FinCg1Start -- a_0, b_0 -- in front of the first): TWO
            // launches per body; the last body is closed by the tail
            bool first = true;
            k.drv.tail = [&]() -> int { return first ? k.drv.scal(FinCg1Start{}) : k.drv.scal(FinPcg1Close{}); };
            rc = k.run_loop([&]() -> int {
                if (first) { TRY(k.drv.vecf(FinCg1Start{}, upd, a_all | al(w))); first = false; }
                else TRY(k.drv.vecf(FinPcg1Close{}, upd, a_all | al(w)));
                if (last_body()) return 0;
                return product();
            });
            k.drv.tail = nullptr;
        } else {
            // progress callback (the state is read after every body) or sharded rows (the step is where the ranks' sums meet)
            TRY(k.drv.scal(FinCg1Start{}));
            rc = k.run_loop([&]() -> int {
                TRY(k.drv.vec(upd, a_all | al(w)));
                if (!last_body()) TRY(product());
                TRY(k.drv.scal(FinPcg1Close{}));
                return 0;
            });
        }
        int rc2 = hb.close(c.stream);
        return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
    }
    int rc = k.run_loop([&]() -> int {
        bool f; TRY(k.ax_dot(d, Ad, d, false, 0, &f));                          // :387
        if (!f) TRY(k.drv.vec(OpDot1{st, d, Ad}, al(d) | al(Ad)));               // :389
        if (invdiag) {
            TRY(k.drv.vecf(FinAlpha{}, OpPcgUpdateJacobi{st, m, r, z, d, Ad, invdiag, 0.0, nt}, a_all));   // :390, :392-414
        } else {
            TRY(k.drv.vecf(FinAlpha{}, OpPcgUpdate{st, m, r, d, Ad, 0.0}, a_all));   // :390, :392-397
            TRY(k.drv.checked_mx([&] { Mfp(inst, r, z, n); }));                  // :399
            TRY(k.drv.vec(OpPcgDots{st, m, r, z}, a_all));                       // :401-414
        }
        TRY(k.drv.vecf(FinClose<false>{}, OpPcgDir{st, d, z, 0.0, nt && invdiag != nullptr}, al(d) | al(z)));  // :415-416, :418-422
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

static int solve_cgs(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, int n,
                     const lcg_para *param, void *inst, double *RK, double *R0T, double *PK, double *AX,
                     double *UK, double *QK, double *WK, int mem)
{
    const lcg_para p = param ? *param : lcg_hip_default_parameters();
    TRY(check_args(p, n, m, B));
    TRY(ensure_init());
    Ctx &c = ctx();
    HostBridge hb; TRY(hb.open(mem, m, B, sizeof(double) * n, c.stream));
    Workspace ws; double *r, *r0, *pk, *Ax, *u, *q, *w;
    const size_t nb = sizeof(double) * n;
    TRY(ws.get(r, RK, nb)); TRY(ws.get(r0, R0T, nb)); TRY(ws.get(pk, PK, nb)); TRY(ws.get(Ax, AX, nb));
    TRY(ws.get(u, UK, nb)); TRY(ws.get(q, QK, nb)); TRY(ws.get(w, WK, nb));
    TRY(Placement::run(c, n, (const void *)Afp, inst, B, ws, {&Ax, &pk, &w, &u, &q, &r, &r0}, 1));        // (both products write Ax; they read p and w)
    RealCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    const bool nt = stream_vectors(n, Afp, inst);      // (n: the rows THIS process holds)
    const uintptr_t a_all = al(m) | al(B) | al(r) | al(r0) | al(pk) | al(Ax) | al(u) | al(q) | al(w);

    TRY(k.ax_setup(m, Ax));                                                      // lcg.cpp:476
    TRY(k.drv.vec(OpShadowInit<true>{st, Ax, B, m, r, r0, pk, u}, a_all));       // :480-497
    TRY(k.drv.scal(FinInit{}));
    int rc = k.run_loop([&]() -> int {
        bool f; TRY(k.ax_dot(pk, Ax, r0, false, 0, &f));                        // :546
        if (!f) TRY(k.drv.vec(OpDot1{st, Ax, r0}, a_all));                       // :548-552
        TRY(k.drv.vecf(FinAlpha{}, OpCgsQW{st, u, Ax, q, w, 0.0, nt}, a_all));       // :553, :556-560
        TRY(k.ax(w, Ax));                                                        // :562
        TRY(k.drv.vec(OpCgsUpdate{st, m, r, w, Ax, r0, 0.0, nt}, a_all));            // :565-588
        TRY(k.drv.vecf(FinClose<false>{}, OpCgsDir{st, u, pk, r, q, 0.0, nt}, a_all));   // :589-590, :593-597
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

static int solve_bicgstab(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, int n,
                          const lcg_para *param, void *inst, int mem)
{
    const lcg_para p = param ? *param : lcg_hip_default_parameters();
    TRY(check_args(p, n, m, B));
    TRY(ensure_init());
    Ctx &c = ctx();
    HostBridge hb; TRY(hb.open(mem, m, B, sizeof(double) * n, c.stream));
    Workspace ws; double *r, *r0, *pk, *Ax, *s, *Ap;
    const size_t nb = sizeof(double) * n;
    TRY(ws.get(r, nullptr, nb)); TRY(ws.get(r0, nullptr, nb)); TRY(ws.get(pk, nullptr, nb));
    TRY(ws.get(Ax, nullptr, nb)); TRY(ws.get(s, nullptr, nb)); TRY(ws.get(Ap, nullptr, nb));
    TRY(Placement::run(c, n, (const void *)Afp, inst, B, ws, {&Ap, &Ax, &pk, &s, &r, &r0}, 2));           // (A.p and A.s; p and s are read)
    RealCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    const bool nt = stream_vectors(n, Afp, inst);      // (n: the rows THIS process holds)
    const uintptr_t a_all = al(m) | al(B) | al(r) | al(r0) | al(pk) | al(Ax) | al(s) | al(Ap);

    TRY(k.ax_setup(m, Ax));                                                      // lcg.cpp:648
    TRY(k.drv.vec(OpShadowInit<false>{st, Ax, B, m, r, r0, pk, nullptr}, a_all)); // :650-667
    TRY(k.drv.scal(FinInit{}));
    int rc = k.run_loop([&]() -> int {
        bool f; TRY(k.ax_dot(pk, Ap, r0, false, 0, &f));                        // :718
        if (!f) TRY(k.drv.vec(OpDot1{st, Ap, r0}, a_all));                       // :720-724
        TRY(k.drv.vecf(FinAlpha{}, OpBicgS{st, r, Ap, s, 0.0, nt}, a_all));          // :725, :727-731
        TRY(k.ax_dot(s, Ax, s, true, 0, &f));                                    // :733
        if (!f) TRY(k.drv.vec(OpDot2{st, Ax, s}, a_all));                        // :735-740
        TRY(k.drv.vecf(FinOmega{}, OpBicgUpdate{st, m, r, pk, s, Ax, r0, 0.0, 0.0, nt}, a_all));   // :741, :743-772
        TRY(k.drv.vecf(FinClose<true>{}, OpBicgDir{st, pk, r, Ap, 0.0, 0.0, nt}, a_all));          // :773-774, :776-780
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

// ---- BiCGStab with restart (lcg.cpp:812-1034) ---------------------------------------------------
// Two things on top of lbicgstab.  (1) When abs_diff is set there is a second stop test in the
// middle of the iteration, on |s|/N, and t advances a second time (:910-939): the iteration is
// enqueued as two counted halves.  If that test fires, m takes the half step m += a p first.
// (2) When |r.r0| < restart_epsilon the shadow residual and the direction restart from r (:982-997).
enum { S_RESTART = 8, S_MID = 9 };

struct OpBicgS2 {       // s = r - a Ap; s.s                                  lcg.cpp:904-912
    static constexpr int NR = 1, SKIP = SKIP_DONE;
    DevState *st; const double *r, *Ap; double *s; double ak;
    __device__ void prep() { ak = st->s[S_AK]; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T sv = vsub(ld<T>(r, i), ak * ld<T>(Ap, i));
        st_(s, i, sv);
        acc[0] += dotp(sv, sv);
    }
};
struct FinMid {         // the mid-iteration stop test (abs_diff only)
    static constexpr int NR = 1;
    int abs_diff;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (!abs_diff) return;
        if (!st->done) {
            st->t++;                                        // the head's t++ of this half
            const double res = sqrt(sum[0]) / st->n_global; // :912-913
            st->residual = res;
            if (res <= st->eps) { st->done = 1; st->status = ST_CONVERGED; st->s[S_MID] = 1.0; }
        }
        publish(st);
    }
};
struct OpMidFinish {    // m += a p when the mid test fired; NaN                lcg.cpp:922-930
    static constexpr int NR = 1, SKIP = SKIP_NEVER;
    DevState *st; double *m; const double *p; double ak; bool active;
    __device__ void prep() { active = st->s[S_MID] == 1.0; ak = st->s[S_AK]; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        if (!active) return;
        const T mv = vadd(ld<T>(m, i), ak * ld<T>(p, i));
        st_(m, i, mv);
        acc[0] += nanflag(mv);
    }
};
struct FinMidNan {
    static constexpr int NR = 1;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (st->s[S_MID] != 1.0) return;
        st->s[S_MID] = 2.0;                                 // applied
        if (sum[0] > 0.0) st->status = ST_NAN;
        publish(st);
    }
};
struct FinOmega2 {      // omega; first scalar step of the second half when halves are counted
    static constexpr int NR = 2;
    int counts_body;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (counts_body) st->it++;
        if (st->done) return;
        st->s[S_WK] = sum[0] / sum[1];                      // :949
    }
};
struct FinClose2 {      // sums: m.m, r.r, r.r0, NaN                            lcg.cpp:956-1003
    static constexpr int NR = 4;
    double restart_eps;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (!st->done) {
            st->s[S_M2] = clamp1(sum[0]);
            if (sum[3] > 0.0 || sum[0] != sum[0]) { st->t++; st->done = 1; st->status = ST_NAN; }
            else {
                const double rho_new = sum[2];
                if (fabs(rho_new) < restart_eps) {          // restart: r0 = p = r, so r.r0 = r.r
                    st->s[S_RESTART] = 1.0;
                    st->s[S_RHO] = sum[1];
                } else {
                    st->s[S_RESTART] = 0.0;
                    st->s[S_BK] = (st->s[S_AK] / st->s[S_WK]) * rho_new / st->s[S_RHO];
                    st->s[S_RHO] = rho_new;
                }
                st->s[S_G2] = sum[1];
                st->t++;
                stop_rule(st, sum[1], st->s[S_M2]);
            }
        }
        publish(st);
    }
};
struct OpBicgDir2 {     // p = r + b (p - w Ap), or the restart r0 = p = r       lcg.cpp:984-1009
    static constexpr int NR = 0, SKIP = SKIP_DIR;
    DevState *st; double *p, *r0; const double *r, *Ap; double bk, wk; bool restart;
    __device__ void prep() { bk = st->s[S_BK]; wk = st->s[S_WK]; restart = st->s[S_RESTART] != 0.0; }
    template <class T> __device__ void apply(long i, double *)
    {
        const T rv = ld<T>(r, i);
        if (restart) { st_(r0, i, rv); st_(p, i, rv); }
        else st_(p, i, vadd(rv, bk * vsub(ld<T>(p, i), wk * ld<T>(Ap, i))));
    }
};

static int solve_bicgstab2(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, int n,
                           const lcg_para *param, void *inst, int mem)
{
    const lcg_para p = param ? *param : lcg_hip_default_parameters();
    if (n <= 0) return LCG_INVILAD_VARIABLE_SIZE;                               // lcg.cpp:818-823
    if (p.max_iterations < 0) return LCG_INVILAD_MAX_ITERATIONS;
    if (p.epsilon <= 0.0) return LCG_INVILAD_EPSILON;
    if (p.restart_epsilon <= 0.0 || p.epsilon >= 1.0) return LCG_INVILAD_RESTART_EPSILON;
    if (m == nullptr || B == nullptr) return LCG_INVALID_POINTER;
    TRY(ensure_init());
    Ctx &c = ctx();
    HostBridge hb; TRY(hb.open(mem, m, B, sizeof(double) * n, c.stream));
    Workspace ws; double *r, *r0, *pk, *Ax, *s, *Ap;
    const size_t nb = sizeof(double) * n;
    TRY(ws.get(r, nullptr, nb)); TRY(ws.get(r0, nullptr, nb)); TRY(ws.get(pk, nullptr, nb));
    TRY(ws.get(Ax, nullptr, nb)); TRY(ws.get(s, nullptr, nb)); TRY(ws.get(Ap, nullptr, nb));
    TRY(Placement::run(c, n, (const void *)Afp, inst, B, ws, {&Ap, &Ax, &pk, &s, &r, &r0}, 2));           // (A.p and A.s; p and s are read)
    RealCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    const uintptr_t a_all = al(m) | al(B) | al(r) | al(r0) | al(pk) | al(Ax) | al(s) | al(Ap);
    const int halves = p.abs_diff ? 1 : 0;

    TRY(k.ax(m, Ax));                                                            // :832
    TRY(k.drv.vec(OpShadowInit<false>{st, Ax, B, m, r, r0, pk, nullptr}, a_all));
    TRY(k.drv.scal(FinInit{}));
    auto first_half = [&]() -> int {
        TRY(k.ax(pk, Ap));                                                       // :895
        TRY(k.drv.vec(OpDot1{st, Ap, r0}, a_all));
        TRY(k.drv.scal(FinAlpha{}));
        TRY(k.drv.vec(OpBicgS2{st, r, Ap, s, 0.0}, a_all));                      // :904-912
        TRY(k.drv.scal(FinMid{halves}));
        if (halves) {
            TRY(k.drv.vec(OpMidFinish{st, m, pk, 0.0, false}, a_all));           // :922-930
            TRY(k.drv.scal(FinMidNan{}));
        }
        return 0;
    };
    auto second_half = [&]() -> int {
        TRY(k.ax(s, Ax));                                                        // :941
        TRY(k.drv.vec(OpDot2{st, Ax, s}, a_all));
        TRY(k.drv.scal(FinOmega2{halves}));
        TRY(k.drv.vec(OpBicgUpdate{st, m, r, pk, s, Ax, r0, 0.0, 0.0}, a_all));  // :951-980
        TRY(k.drv.scal(FinClose2{p.restart_epsilon}));
        TRY(k.drv.vec(OpBicgDir2{st, pk, r0, r, Ap, 0.0, 0.0, false}, a_all));   // :984-1009
        return 0;
    };
    int step = 0;
    int rc = k.run_loop([&]() -> int {
        if (!halves) { TRY(first_half()); return second_half(); }
        return (step++ & 1) ? second_half() : first_half();
    });
    if (rc == LCG_CONVERGENCE || rc == LCG_NAN_VALUE) {      // a NaN found by the mid-iteration half step
        DevState h; TRY(k.drv.read_state(h));
        if (h.status == ST_NAN) rc = LCG_NAN_VALUE;
    }
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

} // namespace lcgh

using namespace lcgh;

extern "C" {

int lcg_hip_solver(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, int n,
                   const lcg_para *param, void *instance, int solver_id, int mem)
{
    switch (solver_id) {                                                        // lcg.cpp:59-82
    case LCG_CG: return solve_cg(Afp, Pfp, m, B, n, param, instance, nullptr, nullptr, nullptr, mem);
    case LCG_BICGSTAB: return solve_bicgstab(Afp, Pfp, m, B, n, param, instance, mem);
    case LCG_BICGSTAB2: return solve_bicgstab2(Afp, Pfp, m, B, n, param, instance, mem);
    case LCG_CGS:
    default: return solve_cgs(Afp, Pfp, m, B, n, param, instance, nullptr, nullptr, nullptr, nullptr,
                              nullptr, nullptr, nullptr, mem);
    }
}

int lcg_hip_solver_preconditioned(lcg_axfunc_ptr Afp, lcg_axfunc_ptr Mfp, lcg_progress_ptr Pfp, double *m,
                                  const double *B, int n, const lcg_para *param, void *instance,
                                  int solver_id, int mem)
{
    (void)solver_id;                                                            // lcg.cpp:87-91
    return solve_pcg(Afp, Mfp, Pfp, m, B, n, param, instance, mem);
}

int lcg_hip_lcg(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, int n,
                const lcg_para *param, void *instance, double *Gk, double *Dk, double *ADk, int mem)
{
    return solve_cg(Afp, Pfp, m, B, n, param, instance, Gk, Dk, ADk, mem);
}

int lcg_hip_lcgs(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, int n,
                 const lcg_para *param, void *instance, double *RK, double *R0T, double *PK, double *AX,
                 double *UK, double *QK, double *WK, int mem)
{
    return solve_cgs(Afp, Pfp, m, B, n, param, instance, RK, R0T, PK, AX, UK, QK, WK, mem);
}

} // extern "C"
