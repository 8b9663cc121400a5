// solvers_cplx.hip -- device-resident complex (c128) BiCG-symmetric, CGS, BiCGStab and TFQMR.
//
// liblcg's clcg.cpp loops re-expressed as fused HIP passes; one complex element (16 B) per lane
// per access.  Inner products follow the reference's two flavours: clcg_dot (no conjugate,
// lcg_complex.cpp:143-154) and clcg_inner (conjugated first argument, :156-167).  <v,v> under
// clcg_inner has an exactly-zero imaginary part, so |<v,v>|^2 -- the 4th-power quantities of
// the stop rule, clcg.cpp:262-270,295-296 -- is carried as (sum |v_i|^2)^2.
//
// Deliberate deviations: the shadow residual rbar0 is drawn from a caller-set
// seed (or supplied explicitly) instead of srand(time(0)); TFQMR returns
// LCG_REACHED_MAX_ITERATIONS at the iteration cap instead of spinning (clcg.cpp:800-804).
#include <cstdlib>
#include <functional>

#include "driver.hpp"

namespace lcgh {

// DevState::s slots (complex values take two)
enum { C_AK = 0, C_BK = 2, C_WK = 4, C_RHO = 6, C_RR = 8, C_M4 = 10, C_R4 = 11,
       T_THETA = 12, T_TAO = 13, T_ETA = 14, T_SIGN = 16, T_RR2 = 18 };

__device__ __forceinline__ double2 lds2(const DevState *st, int i) { return make_double2(st->s[i], st->s[i + 1]); }
__device__ __forceinline__ void sts2(DevState *st, int i, double2 v) { st->s[i] = v.x; st->s[i + 1] = v.y; }
__device__ __forceinline__ double clamp1c(double v) { return v < 1.0 ? 1.0 : v; }
__device__ __forceinline__ double2 L(const double *p, long i) { return reinterpret_cast<const double2 *>(p)[i]; }
__device__ __forceinline__ void S(double *p, long i, double2 v) { reinterpret_cast<double2 *>(p)[i] = v; }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 cneg(double2 a) { return make_double2(-a.x, -a.y); }
__device__ __forceinline__ void acc_inner(double *acc, double2 a, double2 b)   // conj(a)*b
{
    acc[0] += a.x * b.x + a.y * b.y;
    acc[1] += a.x * b.y - a.y * b.x;
}
__device__ __forceinline__ void acc_dot(double *acc, double2 a, double2 b)     // a*b
{
    acc[0] += a.x * b.x - a.y * b.y;
    acc[1] += a.x * b.y + a.y * b.x;
}
__device__ __forceinline__ double cnan(double2 a) { return (a.x != a.x || a.y != a.y) ? 1.0 : 0.0; }

// ---- shared passes ---------------------------------------------------------------------------
template <bool CONJ>
struct OpZDot {     // acc[0..1] = <a,b> (CONJ) or a.b
    static constexpr int NR = 2, SKIP = SKIP_DONE;
    DevState *st; const double *a, *b;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        if (CONJ) acc_inner(acc, L(a, i), L(b, i)); else acc_dot(acc, L(a, i), L(b, i));
    }
};

// alpha = rho / sum  (first scalar step of a body: counts it)
struct FinZAlpha {
    static constexpr int NR = 2;
    int rho_slot;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        st->it++;
        if (st->done) return;
        sts2(st, C_AK, cdiv(lds2(st, rho_slot), make_double2(sum[0], sum[1])));
    }
};

// setup: sums = |m|^2, |r|^2, rho.re, rho.im [, rr.re, rr.im]
template <bool WITH_RR>
struct FinZInit {
    static constexpr int NR = WITH_RR ? 6 : 4;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        const double m4 = clamp1c(sum[0] * sum[0]), r4 = sum[1] * sum[1];
        st->s[C_M4] = m4; st->s[C_R4] = r4;
        sts2(st, C_RHO, make_double2(sum[2], sum[3]));
        if (WITH_RR) sts2(st, C_RR, make_double2(sum[4], sum[5]));
        double r; bool already = false;                    // clcg.cpp:273-290
        if (st->abs_diff && sqrt(r4) / st->n_global <= st->eps) { r = sqrt(r4) / st->n_global; already = true; }
        else if (r4 / m4 <= st->eps) { r = r4 / m4; already = true; }
        else r = st->abs_diff ? sqrt(r4) / st->n_global : r4 / m4;
        st->residual = r;
        if (already) { st->done = 1; st->status = ST_ALREADY; }
        publish(st);
    }
};

// closing step: sums = |m|^2, |r|^2, rhoNew.re, rhoNew.im, NaN.   KIND 0: beta = new/old on
// slot RHO (CGS); 1: same on slot RR (BiCG-sym); 2: BiCGStab beta (clcg.cpp:658)
template <int KIND>
struct FinZClose {
    static constexpr int NR = 5;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (!st->done) {
            const double m4 = clamp1c(sum[0] * sum[0]), r4 = sum[1] * sum[1];
            st->s[C_M4] = m4; st->s[C_R4] = r4;
            if (sum[4] > 0.0 || sum[0] != sum[0]) { st->t++; st->done = 1; st->status = ST_NAN; }
            else {
                const int slot = KIND == 1 ? C_RR : C_RHO;
                const double2 nw = make_double2(sum[2], sum[3]), old = lds2(st, slot);
                double2 bk;
                if (KIND == 2) bk = cdiv(cmul(nw, lds2(st, C_AK)), cmul(old, lds2(st, C_WK)));
                else bk = cdiv(nw, old);
                sts2(st, C_BK, bk);
                sts2(st, slot, nw);
                st->t++;
                stop_rule(st, r4, m4);
            }
        }
        publish(st);
    }
};

// ---- BiCG for complex-symmetric A (clcg.cpp:228-364) ------------------------------------------
struct OpSymInit {  // d = r = B - Ax; |m|^2, |r|^2, (unused rho), r.r
    static constexpr int NR = 6, SKIP = SKIP_NEVER;
    DevState *st; const double *Ax, *B, *m; double *r, *d;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 rv = csub(L(B, i), L(Ax, i)), mv = L(m, i);
        S(r, i, rv); S(d, i, rv);
        acc[0] += cnorm(mv); acc[1] += cnorm(rv);
        acc_dot(acc + 4, rv, rv);
    }
};
struct OpSymUpdate {    // m += a d; r -= a Ax; |m|^2, |r|^2, r.r, NaN   (clcg.cpp:323-345)
    static constexpr int NR = 5, SKIP = SKIP_DONE;
    DevState *st; double *m, *r; const double *d, *Ax; double2 ak;
    __device__ void prep() { ak = lds2(st, C_AK); }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 mv = cfma(ak, L(d, i), L(m, i));
        const double2 rv = cfma(cneg(ak), L(Ax, i), L(r, i));
        S(m, i, mv); S(r, i, rv);
        acc[0] += cnorm(mv); acc[1] += cnorm(rv);
        acc_dot(acc + 2, rv, rv);
        acc[4] += cnan(mv);
    }
};
struct OpZXpay {        // d = r + b d       (clcg.cpp:349-353)
    static constexpr int NR = 0, SKIP = SKIP_DIR;
    DevState *st; double *d; const double *r; double2 bk;
    __device__ void prep() { bk = lds2(st, C_BK); }
    template <class T> __device__ void apply(long i, double *) { S(d, i, cfma(bk, L(d, i), L(r, i))); }
};

// ---- BiCG (clcg.cpp:77-226): needs A^H.x --------------------------------------------------------
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }
struct OpZBicgInit {     // d1 = r1 = B - Ax; d2 = r2 = conj(r1); |m|^2, |r1|^2, <r2,r1>   (clcg.cpp:101-120)
    static constexpr int NR = 4, SKIP = SKIP_NEVER;
    DevState *st; const double *Ax, *B, *m; double *r1, *r2, *d1, *d2;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 rv = csub(L(B, i), L(Ax, i)), mv = L(m, i), rc = cconj(rv);
        S(r1, i, rv); S(d1, i, rv); S(r2, i, rc); S(d2, i, rc);
        acc[0] += cnorm(mv); acc[1] += cnorm(rv);
        acc_inner(acc + 2, rc, rv);
    }
};
struct OpZBicgUpd1 {     // m += a d1; r1 -= a Ax                               (clcg.cpp:173-178)
    static constexpr int NR = 0, SKIP = SKIP_DONE;
    DevState *st; double *m, *r1; const double *d1, *Ax; double2 ak;
    __device__ void prep() { ak = lds2(st, C_AK); }
    template <class T> __device__ void apply(long i, double *)
    {
        S(m, i, cfma(ak, L(d1, i), L(m, i)));
        S(r1, i, cfma(cneg(ak), L(Ax, i), L(r1, i)));
    }
};
struct OpZBicgUpd2 {     // r2 -= conj(a) A^H d2; |m|^2, |r1|^2, <r2,r1>, NaN  (clcg.cpp:180-203)
    static constexpr int NR = 5, SKIP = SKIP_DONE;
    DevState *st; double *r2; const double *AHd, *m, *r1; double2 akc;
    __device__ void prep() { akc = cconj(lds2(st, C_AK)); }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 r2v = cfma(cneg(akc), L(AHd, i), L(r2, i));
        const double2 mv = L(m, i), r1v = L(r1, i);
        S(r2, i, r2v);
        acc[0] += cnorm(mv); acc[1] += cnorm(r1v);
        acc_inner(acc + 2, r2v, r1v);
        acc[4] += cnan(mv);
    }
};
struct OpZBicgDirPair {     // d1 = r1 + b d1; d2 = r2 + conj(b) d2                (clcg.cpp:207-212)
    static constexpr int NR = 0, SKIP = SKIP_DIR;
    DevState *st; double *d1, *d2; const double *r1, *r2; double2 bk;
    __device__ void prep() { bk = lds2(st, C_BK); }
    template <class T> __device__ void apply(long i, double *)
    {
        S(d1, i, cfma(bk, L(d1, i), L(r1, i)));
        S(d2, i, cfma(cconj(bk), L(d2, i), L(r2, i)));
    }
};

// ---- PCG for complex-symmetric A (reference: CUDA back-end only, clcg_cuda.cu:403-558) -------------
// Unconjugated products (cublasZdotu) and the REAL solvers' stop rule |r|^2 / max(|m|^2,1)
// (clcg_cuda.cu:459,472), not the 4th power of clcg.cpp.  No NaN scan in the reference loop.
struct OpZResid {       // r = B - Ax                                          clcg_cuda.cu:442-443
    static constexpr int NR = 0, SKIP = SKIP_NEVER;
    DevState *st; const double *Ax, *B; double *r;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *) { S(r, i, csub(L(B, i), L(Ax, i))); }
};
struct OpZPcgDots {     // |m|^2, |r|^2, r.s                                   clcg_cuda.cu:448-457, 507-516
    static constexpr int NR = 4, SKIP = SKIP_DONE;
    DevState *st; const double *m, *r, *s;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 rv = L(r, i);
        acc[0] += cnorm(L(m, i)); acc[1] += cnorm(rv);
        acc_dot(acc + 2, rv, L(s, i));
    }
};
struct OpZPcgUpd {      // m += a d; r -= a Ax                                 clcg_cuda.cu:504-505
    static constexpr int NR = 0, SKIP = SKIP_DONE;
    DevState *st; double *m, *r; const double *d, *Ax; double2 ak;
    __device__ void prep() { ak = lds2(st, C_AK); }
    template <class T> __device__ void apply(long i, double *)
    {
        S(m, i, cfma(ak, L(d, i), L(m, i)));
        S(r, i, cfma(cneg(ak), L(Ax, i), L(r, i)));
    }
};
struct OpZPcgUpdJacobi {    // built-in Jacobi: update, s = r .* invdiag and the three sums in one pass
    static constexpr int NR = 4, SKIP = SKIP_DONE;
    DevState *st; double *m, *r, *s; const double *d, *Ax, *inv; double2 ak;
    __device__ void prep() { ak = lds2(st, C_AK); }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 mv = cfma(ak, L(d, i), L(m, i));
        const double2 rv = cfma(cneg(ak), L(Ax, i), L(r, i));
        const double2 sv = cmul(L(inv, i), rv);
        S(m, i, mv); S(r, i, rv); S(s, i, sv);
        acc[0] += cnorm(mv); acc[1] += cnorm(rv);
        acc_dot(acc + 2, rv, sv);
    }
};
template <bool INIT>
struct FinZPcg {        // sums: |m|^2, |r|^2, (r.s).re, (r.s).im
    static constexpr int NR = 4;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (INIT || !st->done) {
            const double m2 = clamp1c(sum[0]), r2 = sum[1];
            const double2 nw = make_double2(sum[2], sum[3]);
            if (INIT) {
                const double r = st->abs_diff ? sqrt(r2) / st->n_global : r2 / m2;     // clcg_cuda.cu:456-466
                st->residual = r;
                if (r <= st->eps) { st->done = 1; st->status = ST_ALREADY; }
            } else {
                sts2(st, C_BK, cdiv(nw, lds2(st, C_RHO)));                               // :517
                st->t++;
                stop_rule(st, r2, m2);
            }
            sts2(st, C_RHO, nw);
        }
        publish(st);
    }
};

// ---- preconditioned BiCG (reference: Eigen back-end only, clcg_eigen.cpp:685-802) --------------------------------------
// Eigen's a.dot(b) conjugates a; the stop rule is the 4th-power one of the CPU loops (std::norm of the inner products).
// The shadow residual is recomputed from the OLD residual every iteration (rsk = conj(rk) - conj(ak) Asx, :767), as written.
struct OpZPbInit {      // p = z; ps = conj(z); rs = conj(r); |m|^2, |r|^2, <rs,z>          clcg_eigen.cpp:707-716
    static constexpr int NR = 4, SKIP = SKIP_NEVER;
    DevState *st; const double *z, *r, *m; double *p, *ps, *rs;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 zv = L(z, i), rv = L(r, i), rc = cconj(rv);
        S(p, i, zv); S(ps, i, cconj(zv)); S(rs, i, rc);
        acc[0] += cnorm(L(m, i)); acc[1] += cnorm(rv);
        acc_inner(acc + 2, rc, zv);
    }
};
struct OpZPbUpd {       // m += a p; rs = conj(r) - conj(a) Asx; r -= a Ax                    clcg_eigen.cpp:766-768
    static constexpr int NR = 0, SKIP = SKIP_DONE;
    DevState *st; double *m, *r, *rs; const double *p, *Ax, *Asx; double2 ak;
    __device__ void prep() { ak = lds2(st, C_AK); }
    template <class T> __device__ void apply(long i, double *)
    {
        const double2 rv = L(r, i);
        S(m, i, cfma(ak, L(p, i), L(m, i)));
        S(rs, i, cfma(cneg(cconj(ak)), L(Asx, i), cconj(rv)));
        S(r, i, cfma(cneg(ak), L(Ax, i), rv));
    }
};
struct OpZPbDots {      // |m|^2, |r|^2, <rs,z>, NaN                                           clcg_eigen.cpp:770-777
    static constexpr int NR = 5, SKIP = SKIP_DONE;
    DevState *st; const double *m, *r, *rs, *z;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 mv = L(m, i);
        acc[0] += cnorm(mv); acc[1] += cnorm(L(r, i));
        acc_inner(acc + 2, L(rs, i), L(z, i));
        acc[4] += cnan(mv);
    }
};
struct OpZPbUpdJacobi { // built-in Jacobi: the update, z = r .* invdiag and the sums in one pass
    static constexpr int NR = 5, SKIP = SKIP_DONE;
    DevState *st; double *m, *r, *rs, *z; const double *p, *Ax, *Asx, *inv; double2 ak;
    __device__ void prep() { ak = lds2(st, C_AK); }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 r0 = L(r, i);
        const double2 mv = cfma(ak, L(p, i), L(m, i));
        const double2 rsv = cfma(cneg(cconj(ak)), L(Asx, i), cconj(r0));
        const double2 rv = cfma(cneg(ak), L(Ax, i), r0);
        const double2 zv = cmul(L(inv, i), rv);
        S(m, i, mv); S(rs, i, rsv); S(r, i, rv); S(z, i, zv);
        acc[0] += cnorm(mv); acc[1] += cnorm(rv);
        acc_inner(acc + 2, rsv, zv);
        acc[4] += cnan(mv);
    }
};
struct OpZPbDir {       // p = z + b p; ps = conj(z) + conj(b) ps                               clcg_eigen.cpp:781-782
    static constexpr int NR = 0, SKIP = SKIP_DIR;
    DevState *st; double *p, *ps; const double *z; double2 bk;
    __device__ void prep() { bk = lds2(st, C_BK); }
    template <class T> __device__ void apply(long i, double *)
    {
        const double2 zv = L(z, i);
        S(p, i, cfma(bk, L(p, i), zv));
        S(ps, i, cfma(cconj(bk), L(ps, i), cconj(zv)));
    }
};

// ---- CGS / BiCGStab / TFQMR shared ----------------------------------------------------------------
template <int MODE>   // 0 CGS: p = u = r; 1 BiCGStab: p = r; 2 TFQMR: p = u = r, d = 0
struct OpZShadowInit {  // r = B - Ax ...; |m|^2, |r|^2, <rbar0, r>
    static constexpr int NR = 4, SKIP = SKIP_NEVER;
    DevState *st; const double *Ax, *B, *m, *rb; double *r, *p, *u, *d;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 rv = csub(L(B, i), L(Ax, i)), mv = L(m, i);
        S(r, i, rv); S(p, i, rv);
        if (MODE != 1) S(u, i, rv);
        if (MODE == 2) S(d, i, make_double2(0.0, 0.0));
        acc[0] += cnorm(mv); acc[1] += cnorm(rv);
        acc_inner(acc + 2, L(rb, i), rv);
    }
};
struct OpZQW {          // q = u - a Ax; w = u + q          (clcg.cpp:467-472, 764-769)
    static constexpr int NR = 0, SKIP = SKIP_DONE;
    DevState *st; const double *u, *Ax; double *q, *w; double2 ak;
    __device__ void prep() { ak = lds2(st, C_AK); }
    template <class T> __device__ void apply(long i, double *)
    {
        const double2 uv = L(u, i);
        const double2 qv = cfma(cneg(ak), L(Ax, i), uv);
        S(q, i, qv); S(w, i, cadd(uv, qv));
    }
};
struct OpZCgsUpdate {   // m += a w; r -= a Ax; |m|^2, |r|^2, <rbar0,r>, NaN   (clcg.cpp:476-498)
    static constexpr int NR = 5, SKIP = SKIP_DONE;
    DevState *st; double *m, *r; const double *w, *Ax, *rb; double2 ak;
    __device__ void prep() { ak = lds2(st, C_AK); }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 mv = cfma(ak, L(w, i), L(m, i));
        const double2 rv = cfma(cneg(ak), L(Ax, i), L(r, i));
        S(m, i, mv); S(r, i, rv);
        acc[0] += cnorm(mv); acc[1] += cnorm(rv);
        acc_inner(acc + 2, L(rb, i), rv);
        acc[4] += cnan(mv);
    }
};
struct OpZUP {          // u = r + b q; p = u + b (q + b p)      (clcg.cpp:502-507, 860-865)
    static constexpr int NR = 0, SKIP = SKIP_DIR;
    DevState *st; double *u, *p; const double *r, *q; double2 bk;
    __device__ void prep() { bk = lds2(st, C_BK); }
    template <class T> __device__ void apply(long i, double *)
    {
        const double2 qv = L(q, i);
        const double2 uv = cfma(bk, qv, L(r, i));
        S(u, i, uv);
        S(p, i, cfma(bk, cfma(bk, L(p, i), qv), uv));
    }
};
struct OpZS {           // s = r - a Ap                            (clcg.cpp:624-628)
    static constexpr int NR = 0, SKIP = SKIP_DONE;
    DevState *st; const double *r, *Ap; double *s; double2 ak;
    __device__ void prep() { ak = lds2(st, C_AK); }
    template <class T> __device__ void apply(long i, double *) { S(s, i, cfma(cneg(ak), L(Ap, i), L(r, i))); }
};
struct OpZOmegaDots {   // <As,s> (2), <As,As> (1)                 (clcg.cpp:631-632)
    static constexpr int NR = 3, SKIP = SKIP_DONE;
    DevState *st; const double *As, *s;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 a = L(As, i);
        acc_inner(acc, a, L(s, i));
        acc[2] += cnorm(a);
    }
};
struct FinZOmega {
    static constexpr int NR = 3;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (st->done) return;
        sts2(st, C_WK, cdiv(make_double2(sum[0], sum[1]), make_double2(sum[2], 0.0)));   // clcg.cpp:633
    }
};
struct OpZBicgUpdate {  // m += a p + w s; r = s - w As; sums            (clcg.cpp:635-657)
    static constexpr int NR = 5, SKIP = SKIP_DONE;
    DevState *st; double *m, *r; const double *p, *s, *As, *rb; double2 ak, wk;
    __device__ void prep() { ak = lds2(st, C_AK); wk = lds2(st, C_WK); }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 sv = L(s, i);
        const double2 mv = cfma(wk, sv, cfma(ak, L(p, i), L(m, i)));
        const double2 rv = cfma(cneg(wk), L(As, i), sv);
        S(m, i, mv); S(r, i, rv);
        acc[0] += cnorm(mv); acc[1] += cnorm(rv);
        acc_inner(acc + 2, L(rb, i), rv);
        acc[4] += cnan(mv);
    }
};
struct OpZBicgDir {     // p = r + b (p - w Ap)                           (clcg.cpp:661-665)
    static constexpr int NR = 0, SKIP = SKIP_DIR;
    DevState *st; double *p; const double *r, *Ap; double2 bk, wk;
    __device__ void prep() { bk = lds2(st, C_BK); wk = lds2(st, C_WK); }
    template <class T> __device__ void apply(long i, double *)
    {
        S(p, i, cfma(bk, cfma(cneg(wk), L(Ap, i), L(p, i)), L(r, i)));
    }
};

// ---- TFQMR (clcg.cpp:681-881) -----------------------------------------------------------------------
struct FinTfInit {      // sums: |m|^2, |r|^2, rho.re, rho.im
    static constexpr int NR = 4;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        sts2(st, C_RR, make_double2(sum[1], 0.0));          // <r,r>            :718
        st->s[T_THETA] = 0.0;
        st->s[T_TAO] = sqrt(sum[1] * sum[1]);               // omega = |<r,r>|   :727-728
        sts2(st, T_ETA, make_double2(0.0, 0.0));
        FinZInit<false> base;
        base(st, sum);
    }
};
struct OpTfR {          // r -= alpha Ax; |r|^2                              (clcg.cpp:773-779)
    static constexpr int NR = 1, SKIP = SKIP_DONE;
    DevState *st; double *r; const double *Ax; double2 ak;
    __device__ void prep() { ak = lds2(st, C_AK); }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 rv = cfma(cneg(ak), L(Ax, i), L(r, i));
        S(r, i, rv);
        acc[0] += cnorm(rv);
    }
};
// head of inner step J (clcg.cpp:806-834): t++, sign, omega, theta, tao, eta.  For J == 1 the
// reduction delivers <r,r> of the freshly updated r; for J == 2 there is nothing to reduce and
// this is the first scalar step of the body (so it counts the body).
template <int J>
struct FinTfHead {
    static constexpr int NR = J == 1 ? 1 : 0;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (J == 2) st->it++;
        if (st->done) return;
        if (J == 1) sts2(st, T_RR2, make_double2(sum[0], 0.0));
        st->t++;
        const double2 alpha = lds2(st, C_AK), eta0 = lds2(st, T_ETA);
        const double th0 = st->s[T_THETA];
        const double2 ea = cdiv(eta0, alpha);
        sts2(st, T_SIGN, make_double2(th0 * th0 * ea.x, th0 * th0 * ea.y));     // :809
        const double rr2 = sqrt(cnorm(lds2(st, T_RR2)));
        const double omega = J == 1 ? sqrt(sqrt(cnorm(lds2(st, C_RR))) * rr2) : rr2;   // :813 / :823
        const double theta = omega / st->s[T_TAO];                             // :832
        st->s[T_TAO] = omega / sqrt(1.0 + theta * theta);                       // :833
        st->s[T_THETA] = theta;
        const double f = 1.0 / (1.0 + theta * theta);
        sts2(st, T_ETA, make_double2(f * alpha.x, f * alpha.y));               // :834
    }
};
struct OpTfDM {         // d = (u|q) + sign d; m += eta d; |m|^2, NaN          (clcg.cpp:815-852)
    static constexpr int NR = 2, SKIP = SKIP_DONE;
    DevState *st; double *d, *m; const double *src; double2 sign, eta;
    __device__ void prep() { sign = lds2(st, T_SIGN); eta = lds2(st, T_ETA); }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 dv = cfma(sign, L(d, i), L(src, i));
        const double2 mv = cfma(eta, dv, L(m, i));
        S(d, i, dv); S(m, i, mv);
        acc[0] += cnorm(mv); acc[1] += cnan(mv);
    }
};
template <int J>
struct FinTfStepClose { // m4, NaN; after J == 1 the next head (J == 2) tests with the OLD r4
    static constexpr int NR = 2;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (!st->done) {
            st->s[C_M4] = clamp1c(sum[0] * sum[0]);
            if (sum[1] > 0.0 || sum[0] != sum[0]) { st->done = 1; st->status = ST_NAN; }
            else if (J == 1) stop_rule(st, st->s[C_R4], st->s[C_M4]);
        }
        if (J == 1) publish(st);
    }
};
struct FinTfTail {      // sums: <rbar0, r>.  rr = rr2, r4, beta, rho; stop rule for the next pass
    static constexpr int NR = 2;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        if (!st->done) {
            const double2 rr2 = lds2(st, T_RR2);
            sts2(st, C_RR, rr2);                                                // :853
            st->s[C_R4] = cnorm(rr2);                                           // :854
            const double2 nw = make_double2(sum[0], sum[1]);
            sts2(st, C_BK, cdiv(nw, lds2(st, C_RHO)));                          // :857
            sts2(st, C_RHO, nw);
            stop_rule(st, st->s[C_R4], st->s[C_M4]);
        }
        publish(st);
    }
};

// ---- host drivers ---------------------------------------------------------------------------------
static int ccheck_args(const clcg_para &p, int n, const double *m, const double *B)
{   // clcg.cpp:235-240 and twins
    if (n <= 0) return CLCG_INVILAD_VARIABLE_SIZE;
    if (p.max_iterations < 0) return CLCG_INVILAD_MAX_ITERATIONS;
    if (p.epsilon <= 0.0 || p.epsilon >= 1.0) return CLCG_INVILAD_EPSILON;
    if (m == nullptr || B == nullptr) return CLCG_INVALID_POINTER;
    return 0;
}

double global_rows(Ctx &c, int n);
double global_rows_of(Ctx &c, int n, const void *afp, const void *inst);   // comm.hip

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

struct CplxCommon {
    Ctx &c; Driver drv; clcg_para para; void *inst; clcg_hip_axfunc_ptr Afp; clcg_hip_progress_ptr Pfp;
    double *m; int n;
    CplxCommon(Ctx &c_, int n_, const clcg_para &p, void *inst_, clcg_hip_axfunc_ptr A, clcg_hip_progress_ptr P, double *m_)
        : c(c_), drv(c_, n_, true, p.max_iterations, p.epsilon, p.abs_diff), para(p), inst(inst_), Afp(A), Pfp(P), m(m_), n(n_)
    { drv.user_cb = A != clcg_hip_csr_ax; }
    int ax(const double *x, double *y) { return drv.timed_ax([&] { Afp(inst, x, y, n, 0, 0); }); }
    int axop(const double *x, double *y, int layout, int conj) { return drv.timed_ax([&] { Afp(inst, x, y, n, layout, conj); }); }
    int run_loop(const std::function<int()> &body)
    {
        auto pfp = [&](double resid, int t) -> int { return Pfp(inst, m, resid, &para, n, t); };
        // the complex loops hand back the REAL enum's iteration-cap code (clcg.cpp:126,164 ...)
        return drv.run(body, Pfp != nullptr, pfp, LCG_REACHED_MAX_ITERATIONS, CLCG_NAN_VALUE);
    }
};

// rbar0 in [1,2] + 0i: lcg_complex.cpp:118-127 with an explicit seed; or the caller's vector
static int make_shadow(Ctx &c, int n, double *dev)
{
    std::vector<double> h(2 * (size_t)n);
    if (c.shadow_vec.size() == h.size()) { h = c.shadow_vec; c.shadow_vec.clear(); }
    else {
        std::srand(c.shadow_seed);
        for (int i = 0; i < n; i++) {
            h[2 * i] = (2.0 - 1.0) * std::rand() * 1.0 / RAND_MAX + 1.0;
            h[2 * i + 1] = (0.0 - 0.0) * std::rand() * 1.0 / RAND_MAX + 0.0;
        }
    }
    HIPCHK(hipMemcpyAsync(dev, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return 0;
}

static int solve_cbicg(clcg_hip_axfunc_ptr Afp, clcg_hip_progress_ptr Pfp, double *m, const double *B, int n,
                       const clcg_para *param, void *inst, int mem)
{
    const clcg_para p = param ? *param : clcg_hip_default_parameters();
    TRY(ccheck_args(p, n, m, B));
    TRY(ensure_init());
    Ctx &c = ctx();
    const size_t nb = sizeof(double) * 2 * (size_t)n;
    HostBridge hb; TRY(hb.open(mem, m, B, nb, c.stream));
    Workspace ws; double *r1, *r2, *d1, *d2, *Ax;
    TRY(ws.get(r1, nullptr, nb)); TRY(ws.get(r2, nullptr, nb)); TRY(ws.get(d1, nullptr, nb));
    TRY(ws.get(d2, nullptr, nb)); TRY(ws.get(Ax, nullptr, nb));
    CplxCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;

    TRY(k.ax(m, Ax));                                                   // clcg.cpp:99
    TRY(k.drv.vec(OpZBicgInit{st, Ax, B, m, r1, r2, d1, d2}));           // :101-120
    TRY(k.drv.scal(FinZInit<false>{}));
    int rc = k.run_loop([&]() -> int {
        TRY(k.ax(d1, Ax));                                              // :169
        TRY(k.drv.vec(OpZDot<true>{st, d2, Ax}));                       // :170
        TRY(k.drv.vecf(FinZAlpha{C_RHO}, OpZBicgUpd1{st, m, r1, d1, Ax, {}}));   // :171 | :173-178
        TRY(k.axop(d2, Ax, 1, 1));                                      // :187  A^H.d2
        TRY(k.drv.vec(OpZBicgUpd2{st, r2, Ax, m, r1, {}}));              // :180-203
        TRY(k.drv.vecf(FinZClose<0>{}, OpZBicgDirPair{st, d1, d2, r1, r2, {}}));   // :204-205 | :207-212
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

static int solve_cpcg(clcg_hip_axfunc_ptr Afp, clcg_hip_axfunc_ptr Mfp, clcg_hip_progress_ptr Pfp, double *m,
                      const double *B, int n, const clcg_para *param, void *inst, int mem)
{
    const clcg_para p = param ? *param : clcg_hip_default_parameters();
    TRY(ccheck_args(p, n, m, B));
    if (Mfp == nullptr) return LCG_NULL_PRECONDITION_MATRIX;
    TRY(ensure_init());
    Ctx &c = ctx();
    const size_t nb = sizeof(double) * 2 * (size_t)n;
    HostBridge hb; TRY(hb.open(mem, m, B, nb, c.stream));
    Workspace ws; double *r, *d, *s, *Ax;
    TRY(ws.get(r, nullptr, nb)); TRY(ws.get(d, nullptr, nb)); TRY(ws.get(s, nullptr, nb)); TRY(ws.get(Ax, nullptr, nb));
    CplxCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    const double *inv = nullptr;        // built-in Jacobi on a complex handle: fold M^-1 into the update
    if (Mfp != clcg_hip_jacobi_mx) k.drv.user_cb = true;
    if (Mfp == clcg_hip_jacobi_mx && inst) {
        const lcg_hip_csr *A = static_cast<const lcg_hip_csr *>(inst);
        if (A->is_complex && A->n_rows == n) inv = A->invdiag;
    }

    TRY(k.ax(m, Ax));                                                   // clcg_cuda.cu:441
    TRY(k.drv.vec(OpZResid{st, Ax, B, r}));                             // :442-443
    TRY(k.drv.checked_mx([&] { Mfp(inst, r, d, n, 0, 0); }));           // :445
    TRY(k.drv.vec(OpZPcgDots{st, m, r, d}));                            // :448-457
    TRY(k.drv.scal(FinZPcg<true>{}));
    int rc = k.run_loop([&]() -> int {
        TRY(k.ax(d, Ax));                                               // :500
        TRY(k.drv.vec(OpZDot<false>{st, d, Ax}));                       // :501
        if (inv) {
            TRY(k.drv.vecf(FinZAlpha{C_RHO}, OpZPcgUpdJacobi{st, m, r, s, d, Ax, inv, {}}));   // :502 | :504-516
        } else {
            TRY(k.drv.vecf(FinZAlpha{C_RHO}, OpZPcgUpd{st, m, r, d, Ax, {}}));             // :502 | :504-505
            TRY(k.drv.checked_mx([&] { Mfp(inst, r, s, n, 0, 0); }));   // :513
            TRY(k.drv.vec(OpZPcgDots{st, m, r, s}));                    // :507-516
        }
        TRY(k.drv.vecf(FinZPcg<false>{}, OpZXpay{st, d, s, {}}));   // :517 | :519-520  d = s + b d
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

static int solve_cpbicg(clcg_hip_axfunc_ptr Afp, clcg_hip_axfunc_ptr Mfp, clcg_hip_progress_ptr Pfp, double *m,
                        const double *B, int n, const clcg_para *param, void *inst, int mem)
{
    const clcg_para p = param ? *param : clcg_hip_default_parameters();
    TRY(ccheck_args(p, n, m, B));                                       // clcg_eigen.cpp:693-697
    if (Mfp == nullptr) return LCG_NULL_PRECONDITION_MATRIX;
    TRY(ensure_init());
    Ctx &c = ctx();
    const size_t nb = sizeof(double) * 2 * (size_t)n;
    HostBridge hb; TRY(hb.open(mem, m, B, nb, c.stream));
    Workspace ws; double *r, *rs, *z, *pk, *ps, *Ax, *Asx;
    TRY(ws.get(r, nullptr, nb)); TRY(ws.get(rs, nullptr, nb)); TRY(ws.get(z, nullptr, nb)); TRY(ws.get(pk, nullptr, nb));
    TRY(ws.get(ps, nullptr, nb)); TRY(ws.get(Ax, nullptr, nb)); TRY(ws.get(Asx, nullptr, nb));
    CplxCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    const double *inv = nullptr;        // built-in Jacobi on a complex handle: fold M^-1 into the update
    if (Mfp != clcg_hip_jacobi_mx) k.drv.user_cb = true;
    if (Mfp == clcg_hip_jacobi_mx && inst) {
        const lcg_hip_csr *A = static_cast<const lcg_hip_csr *>(inst);
        if (A->is_complex && A->n_rows == n) inv = A->invdiag;
    }

    TRY(k.ax(m, Ax));                                                   // :702
    TRY(k.drv.vec(OpZResid{st, Ax, B, r}));                             // :704
    TRY(k.drv.checked_mx([&] { Mfp(inst, r, z, n, 0, 0); }));           // :705
    TRY(k.drv.vec(OpZPbInit{st, z, r, m, pk, ps, rs}));                 // :707-716
    TRY(k.drv.scal(FinZInit<false>{}));                                 // :719-736
    int rc = k.run_loop([&]() -> int {
        TRY(k.ax(pk, Ax));                                              // :760
        TRY(k.axop(ps, Asx, 0, 1));                                     // :761  conj(A).ps
        TRY(k.drv.vec(OpZDot<true>{st, ps, Ax}));                       // :763
        if (inv) {
            TRY(k.drv.vecf(FinZAlpha{C_RHO}, OpZPbUpdJacobi{st, m, r, rs, z, pk, Ax, Asx, inv, {}}));     // :764 | :766-777
        } else {
            TRY(k.drv.vecf(FinZAlpha{C_RHO}, OpZPbUpd{st, m, r, rs, pk, Ax, Asx, {}}));                   // :764 | :766-768
            TRY(k.drv.checked_mx([&] { Mfp(inst, r, z, n, 0, 0); }));   // :775
            TRY(k.drv.vec(OpZPbDots{st, m, r, rs, z}));                 // :770-777
        }
        TRY(k.drv.vecf(FinZClose<0>{}, OpZPbDir{st, pk, ps, z, {}}));   // :778-779 | :781-782
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

static int solve_bicg_sym(clcg_hip_axfunc_ptr Afp, clcg_hip_progress_ptr Pfp, double *m, const double *B, int n,
                          const clcg_para *param, void *inst, int mem)
{
    const clcg_para p = param ? *param : clcg_hip_default_parameters();
    TRY(ccheck_args(p, n, m, B));
    TRY(ensure_init());
    Ctx &c = ctx();
    const size_t nb = sizeof(double) * 2 * (size_t)n;
    HostBridge hb; TRY(hb.open(mem, m, B, nb, c.stream));
    Workspace ws; double *r, *d, *Ax;
    TRY(ws.get(r, nullptr, nb)); TRY(ws.get(d, nullptr, nb)); TRY(ws.get(Ax, nullptr, nb));
    CplxCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;

    TRY(k.ax(m, Ax));                                                   // clcg.cpp:250
    TRY(k.drv.vec(OpSymInit{st, Ax, B, m, r, d}));                      // :252-270
    TRY(k.drv.scal(FinZInit<true>{}));
    int rc = k.run_loop([&]() -> int {
        TRY(k.ax(d, Ax));                                               // :319
        TRY(k.drv.vec(OpZDot<false>{st, d, Ax}));                       // :320
        TRY(k.drv.vecf(FinZAlpha{C_RR}, OpSymUpdate{st, m, r, d, Ax, {}}));   // :321 | :323-345
        TRY(k.drv.vecf(FinZClose<1>{}, OpZXpay{st, d, r, {}}));   // :346-347 | :349-353
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

static int solve_ccgs(clcg_hip_axfunc_ptr Afp, clcg_hip_progress_ptr Pfp, double *m, const double *B, int n,
                      const clcg_para *param, void *inst, int mem)
{
    const clcg_para p = param ? *param : clcg_hip_default_parameters();
    TRY(ccheck_args(p, n, m, B));
    TRY(ensure_init());
    Ctx &c = ctx();
    const size_t nb = sizeof(double) * 2 * (size_t)n;
    HostBridge hb; TRY(hb.open(mem, m, B, nb, c.stream));
    Workspace ws; double *r, *rb, *pk, *Ax, *u, *q, *w;
    TRY(ws.get(r, nullptr, nb)); TRY(ws.get(rb, nullptr, nb)); TRY(ws.get(pk, nullptr, nb)); TRY(ws.get(Ax, nullptr, nb));
    TRY(ws.get(u, nullptr, nb)); TRY(ws.get(q, nullptr, nb)); TRY(ws.get(w, nullptr, nb));
    CplxCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    TRY(make_shadow(c, n, rb));                                         // clcg.cpp:399-404

    TRY(k.ax(m, Ax));                                                   // :391
    TRY(k.drv.vec(OpZShadowInit<0>{st, Ax, B, m, rb, r, pk, u, nullptr}));  // :393-415
    TRY(k.drv.scal(FinZInit<false>{}));
    int rc = k.run_loop([&]() -> int {
        TRY(k.ax(pk, Ax));                                              // :463
        TRY(k.drv.vec(OpZDot<true>{st, rb, Ax}));                       // :464
        TRY(k.drv.vecf(FinZAlpha{C_RHO}, OpZQW{st, u, Ax, q, w, {}}));   // :465 | :467-472
        TRY(k.ax(w, Ax));                                               // :474
        TRY(k.drv.vec(OpZCgsUpdate{st, m, r, w, Ax, rb, {}}));          // :476-498
        TRY(k.drv.vecf(FinZClose<0>{}, OpZUP{st, u, pk, r, q, {}}));   // :499-500 | :502-507
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

static int solve_cbicgstab(clcg_hip_axfunc_ptr Afp, clcg_hip_progress_ptr Pfp, double *m, const double *B, int n,
                           const clcg_para *param, void *inst, int mem)
{
    const clcg_para p = param ? *param : clcg_hip_default_parameters();
    TRY(ccheck_args(p, n, m, B));
    TRY(ensure_init());
    Ctx &c = ctx();
    const size_t nb = sizeof(double) * 2 * (size_t)n;
    HostBridge hb; TRY(hb.open(mem, m, B, nb, c.stream));
    Workspace ws; double *r, *rb, *pk, *s, *Ap, *As;
    TRY(ws.get(r, nullptr, nb)); TRY(ws.get(rb, nullptr, nb)); TRY(ws.get(pk, nullptr, nb));
    TRY(ws.get(s, nullptr, nb)); TRY(ws.get(Ap, nullptr, nb)); TRY(ws.get(As, nullptr, nb));
    CplxCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    TRY(make_shadow(c, n, rb));                                         // clcg.cpp:556-561

    TRY(k.ax(m, Ap));                                                   // :548
    TRY(k.drv.vec(OpZShadowInit<1>{st, Ap, B, m, rb, r, pk, nullptr, nullptr}));   // :550-572
    TRY(k.drv.scal(FinZInit<false>{}));
    int rc = k.run_loop([&]() -> int {
        TRY(k.ax(pk, Ap));                                              // :620
        TRY(k.drv.vec(OpZDot<true>{st, rb, Ap}));                       // :621
        TRY(k.drv.vecf(FinZAlpha{C_RHO}, OpZS{st, r, Ap, s, {}}));   // :622 | :624-628
        TRY(k.ax(s, As));                                               // :630
        TRY(k.drv.vec(OpZOmegaDots{st, As, s}));                        // :631-632
        TRY(k.drv.vecf(FinZOmega{}, OpZBicgUpdate{st, m, r, pk, s, As, rb, {}, {}}));   // :633 | :635-657
        TRY(k.drv.vecf(FinZClose<2>{}, OpZBicgDir{st, pk, r, Ap, {}, {}}));   // :658-659 | :661-665
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

static int solve_tfqmr(clcg_hip_axfunc_ptr Afp, clcg_hip_progress_ptr Pfp, double *m, const double *B, int n,
                       const clcg_para *param, void *inst, int mem)
{
    const clcg_para p = param ? *param : clcg_hip_default_parameters();
    TRY(ccheck_args(p, n, m, B));
    TRY(ensure_init());
    Ctx &c = ctx();
    const size_t nb = sizeof(double) * 2 * (size_t)n;
    HostBridge hb; TRY(hb.open(mem, m, B, nb, c.stream));
    Workspace ws; double *pk, *u, *v, *d, *rb, *r, *Ax, *q, *uq;
    TRY(ws.get(pk, nullptr, nb)); TRY(ws.get(u, nullptr, nb)); TRY(ws.get(v, nullptr, nb)); TRY(ws.get(d, nullptr, nb));
    TRY(ws.get(rb, nullptr, nb)); TRY(ws.get(r, nullptr, nb)); TRY(ws.get(Ax, nullptr, nb)); TRY(ws.get(q, nullptr, nb));
    TRY(ws.get(uq, nullptr, nb));
    CplxCommon k(c, n, p, inst, Afp, Pfp, m);
    TRY(k.drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    TRY(make_shadow(c, n, rb));                                         // clcg.cpp:721-725

    TRY(k.ax(m, Ax));                                                   // :707
    TRY(k.drv.vec(OpZShadowInit<2>{st, Ax, B, m, rb, r, pk, u, d}));    // :709-735
    TRY(k.drv.scal(FinTfInit{}));
    int step = 0;   // counted iterations enqueued: odd = (pass head + inner step 1), even = (inner step 2 + pass tail)
    int rc = k.run_loop([&]() -> int {
        step++;
        if (step & 1) {
            TRY(k.ax(pk, v));                                           // :759
            TRY(k.drv.vec(OpZDot<true>{st, rb, v}));                    // :761
            TRY(k.drv.vecf(FinZAlpha{C_RHO}, OpZQW{st, u, v, q, uq, {}}));   // :762 | :764-769
            TRY(k.ax(uq, Ax));                                          // :771
            TRY(k.drv.vec(OpTfR{st, r, Ax, {}}));                       // :773-779
            TRY(k.drv.vecf(FinTfHead<1>{}, OpTfDM{st, d, m, u, {}, {}}));   // :806-834 (j = 1) | :815-840
            TRY(k.drv.scal(FinTfStepClose<1>{}));                       // :842-852
        } else {
            TRY(k.drv.vecf(FinTfHead<2>{}, OpTfDM{st, d, m, q, {}, {}}));   // :806-834 (j = 2) | :825-840
            TRY(k.drv.vecf(FinTfStepClose<2>{}, OpZDot<true>{st, rb, r}));   // :842-852 | :856
            TRY(k.drv.vecf(FinTfTail{}, OpZUP{st, u, pk, r, q, {}}));   // :853-858 | :860-865
        }
        return 0;
    });
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

} // namespace lcgh

using namespace lcgh;

extern "C" int clcg_hip_solver(clcg_hip_axfunc_ptr Afp, clcg_hip_progress_ptr Pfp, double *m, const double *B, int n,
                               const clcg_para *param, void *instance, int solver_id, int mem)
{
    switch (solver_id) {                                                // clcg.cpp:46-74
    case CLCG_BICG: return solve_cbicg(Afp, Pfp, m, B, n, param, instance, mem);
    case CLCG_BICG_SYM: return solve_bicg_sym(Afp, Pfp, m, B, n, param, instance, mem);
    case CLCG_BICGSTAB: return solve_cbicgstab(Afp, Pfp, m, B, n, param, instance, mem);
    case CLCG_TFQMR: return solve_tfqmr(Afp, Pfp, m, B, n, param, instance, mem);
    case CLCG_CGS:
    default: return solve_ccgs(Afp, Pfp, m, B, n, param, instance, mem);
    }
}

// clcg_solver_preconditioned_cuda (clcg_cuda.h:105-108) -> clpcg (clcg_cuda.cu:403-558);
// clcg_solver_preconditioned_eigen (clcg_eigen.h:87-92, clcg_eigen.cpp:78-93): CLCG_PBICG -> clpbicg, CLCG_PCG -> clpcg.
// Here: CLCG_PBICG runs clpbicg, every other id clpcg (the CUDA entry's only loop and this entry's default).
extern "C" int clcg_hip_solver_preconditioned(clcg_hip_axfunc_ptr Afp, clcg_hip_axfunc_ptr Mfp, clcg_hip_progress_ptr Pfp,
                                              double *m, const double *B, int n, const clcg_para *param,
                                              void *instance, int solver_id, int mem)
{
    if (solver_id == CLCG_PBICG) return solve_cpbicg(Afp, Mfp, Pfp, m, B, n, param, instance, mem);
    return solve_cpcg(Afp, Mfp, Pfp, m, B, n, param, instance, mem);
}

// complex Jacobi z = x ./ diag (reciprocal multiply) with the complex callback signature:
// clcg_vecDvecZ_element_wise as used by sample10.cu:117
extern "C" void clcg_hip_jacobi_mx(void *instance, const double *x, double *z, const int n, int layout, int conjugate)
{
    (void)layout; (void)conjugate;
    const int rc = jacobi_launch(static_cast<lcg_hip_csr *>(instance), x, z, n, ctx().stream);
    if (rc && !ctx().ax_rc) ctx().ax_rc = rc;          // void callback: parked for the loop (driver.hpp: checked_mx)
}
