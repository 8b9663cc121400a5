// solvers_box.hip -- box-constrained solvers of lcg_solver_constrained() (lcg.h:111-113):
// projected gradient with Barzilai-Borwein steps (lpg, lcg.cpp:1054-1204) and spectral projected
// gradient with a non-monotone line search (lspg, lcg.cpp:1224-1446), plus the box projection
// itself (lcg_set2box, algebra.cpp:50-58; lcg_set2box_cuda, algebra_cuda.cu:26-38).
//
// lpg is fully device resident (two fused passes + A.x + one scalar step per iteration).
// lspg's line search `while (q(m + a d) > max(q history) + sigma a g.d)` (lcg.cpp:1372-1392) is a
// data-dependent loop around A.x: the host reads the two sums once per trial step.
#include <algorithm>
#include <cstdlib>
#include <functional>

#include "driver.hpp"

namespace lcgh {

enum { B_ALPHA = 0, B_M2 = 4, B_G2 = 5 };   // DevState::s slots (B_M2/B_G2 mirror S_M2/S_G2)

__device__ __forceinline__ double box1(double lo, double hi, double a)
{   // algebra.cpp:50-58 with both bounds inclusive
    return a >= hi ? hi : (a <= lo ? lo : a);
}
__device__ __forceinline__ double box(double lo, double hi, double a) { return box1(lo, hi, a); }
__device__ __forceinline__ double2 box(double2 lo, double2 hi, double2 a)
{
    return make_double2(box1(lo.x, hi.x, a.x), box1(lo.y, hi.y, a.y));
}

struct OpClamp {        // a = clamp(a)                                        lcg.cpp:1084-1088
    static constexpr int NR = 0, SKIP = SKIP_NEVER;
    DevState *st; double *a; const double *low, *hig;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *) { st_(a, i, box(ld<T>(low, i), ld<T>(hig, i), ld<T>(a, i))); }
};
struct OpBoxInit {      // g = Ad - B; m.m, g.g, q = sum(0.5 m Ad - B m)      lcg.cpp:1092-1104, 1297-1300
    static constexpr int NR = 3, SKIP = SKIP_NEVER;
    DevState *st; const double *Ad, *B, *m; double *g;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T a = ld<T>(Ad, i), b = ld<T>(B, i), mv = ld<T>(m, i);
        st_(g, i, vsub(a, b));
        acc[0] += dotp(mv, mv);
        const T gv = vsub(a, b);
        acc[1] += dotp(gv, gv);
        acc[2] += dotp(mv, vsub(0.5 * a, b));
    }
};
struct FinBoxInit {
    static constexpr int NR = 3;
    double step;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        const double m2 = sum[0] < 1.0 ? 1.0 : sum[0], g2 = sum[1];
        st->s[B_M2] = m2; st->s[B_G2] = g2; st->s[B_ALPHA] = step; st->s[8] = sum[2];
        double r; bool already = false;                     // lcg.cpp:1110-1127
        if (st->abs_diff && sqrt(g2) / st->n_global <= st->eps) { r = sqrt(g2) / st->n_global; already = true; }
        else if (g2 / m2 <= st->eps) { r = g2 / m2; already = true; }
        else r = st->abs_diff ? sqrt(g2) / st->n_global : g2 / m2;
        st->residual = r;
        if (already) { st->done = 1; st->status = ST_ALREADY; }
        publish(st);
    }
};
struct OpPgStep {       // mn = clamp(m - a g)                                 lcg.cpp:1151-1155
    static constexpr int NR = 0, SKIP = SKIP_DONE;
    DevState *st; const double *m, *g, *low, *hig; double *mn; double alpha;
    __device__ void prep() { alpha = st->s[B_ALPHA]; }
    template <class T> __device__ void apply(long i, double *)
    {
        st_(mn, i, box(ld<T>(low, i), ld<T>(hig, i), vsub(ld<T>(m, i), alpha * ld<T>(g, i))));
    }
};
struct OpBoxUpdate {    // gn = Ad - B; s = mn - m; y = gn - g; s.s, s.y; m = mn; g = gn; m.m, g.g   lcg.cpp:1159-1190
    static constexpr int NR = 4, SKIP = SKIP_DONE;
    DevState *st; const double *Ad, *B, *mn; double *m, *g;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T gn = vsub(ld<T>(Ad, i), ld<T>(B, i)), mnv = ld<T>(mn, i);
        const T s = vsub(mnv, ld<T>(m, i)), y = vsub(gn, ld<T>(g, i));
        acc[0] += dotp(s, s); acc[1] += dotp(s, y);
        st_(m, i, mnv); st_(g, i, gn);
        acc[2] += dotp(mnv, mnv); acc[3] += dotp(gn, gn);
    }
};
struct FinBoxClose {    // step = s.s / s.y; stop rule
    static constexpr int NR = 4;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        st->it++;
        if (!st->done) {
            st->s[B_ALPHA] = sum[0] / sum[1];               // lcg.cpp:1176 / 1418
            st->s[B_M2] = sum[2] < 1.0 ? 1.0 : sum[2];
            st->s[B_G2] = sum[3];
            st->t++;
            stop_rule(st, sum[3], st->s[B_M2]);
        }
        publish(st);
    }
};
// ---- SPG extras
struct OpSpgDir {       // d = clamp(m - lambda g) - m; mn = m + d             lcg.cpp:1339-1349
    static constexpr int NR = 1, SKIP = SKIP_DONE;      // acc0 = g.d
    DevState *st; const double *m, *g, *low, *hig; double *d, *mn; double lambda;
    __device__ void prep() { lambda = st->s[B_ALPHA]; }
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T mv = ld<T>(m, i), gv = ld<T>(g, i);
        const T dv = vsub(box(ld<T>(low, i), ld<T>(hig, i), vsub(mv, lambda * gv)), mv);
        st_(d, i, dv); st_(mn, i, vadd(mv, dv));
        acc[0] += dotp(gv, dv);
    }
};
struct OpSpgTrial {     // mn = m + a d                                        lcg.cpp:1375-1379
    static constexpr int NR = 0, SKIP = SKIP_DONE;
    DevState *st; const double *m, *d; double *mn; double ak;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *) { st_(mn, i, vadd(ld<T>(m, i), ak * ld<T>(d, i))); }
};
struct OpSpgQ {         // q = sum(0.5 mn Ad - B mn)                           lcg.cpp:1353-1357
    static constexpr int NR = 1, SKIP = SKIP_DONE;
    DevState *st; const double *mn, *Ad, *B;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        acc[0] += dotp(ld<T>(mn, i), vsub(0.5 * ld<T>(Ad, i), ld<T>(B, i)));
    }
};
template <int SLOT> struct FinStore1 {   // park one reduced sum in DevState::s[SLOT]
    static constexpr int NR = 1;
    __device__ void operator()(DevState *st, const double *sum) const { st->s[SLOT] = sum[0]; }
};

static inline uintptr_t al(const void *p) { return (uintptr_t)p; }
double global_rows(Ctx &c, int n);
double global_rows_of(Ctx &c, int n, const void *afp, const void *inst);   // comm.hip
#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

static int solve_pg(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, const double *low,
                    const double *hig, int n, const lcg_para *param, void *inst, int mem)
{
    const lcg_para p = param ? *param : lcg_hip_default_parameters();
    if (n <= 0) return LCG_INVILAD_VARIABLE_SIZE;                               // lcg.cpp:1060-1067
    if (p.max_iterations < 0) return LCG_INVILAD_MAX_ITERATIONS;
    if (p.epsilon <= 0.0) return LCG_INVILAD_EPSILON;
    if (p.step <= 0.0 || p.epsilon >= 1.0) return LCG_INVALID_LAMBDA;
    if (!m || !B || !low || !hig) return LCG_INVALID_POINTER;
    TRY(ensure_init());
    Ctx &c = ctx();
    const size_t nb = sizeof(double) * n;
    HostBridge hb; TRY(hb.open(mem, m, B, nb, c.stream));
    Workspace ws; double *g, *Ad, *mn, *dlow = nullptr, *dhig = nullptr;
    TRY(ws.get(g, nullptr, nb)); TRY(ws.get(Ad, nullptr, nb)); TRY(ws.get(mn, nullptr, nb));
    if (mem == LCG_HIP_MEM_HOST) {
        TRY(ws.get(dlow, nullptr, nb)); TRY(ws.get(dhig, nullptr, nb));
        HIPCHK(hipMemcpyAsync(dlow, low, nb, hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipMemcpyAsync(dhig, hig, nb, hipMemcpyHostToDevice, c.stream));
        low = dlow; hig = dhig;
    }
    Driver drv(c, n, false, p.max_iterations, p.epsilon, p.abs_diff);
    drv.user_cb = Afp != lcg_hip_csr_ax;
    TRY(drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    const uintptr_t a_all = al(m) | al(B) | al(g) | al(Ad) | al(mn) | al(low) | al(hig);
    lcg_para para = p;
    auto ax = [&](const double *x, double *y) { return drv.timed_ax([&] { Afp(inst, x, y, n); }); };

    TRY(drv.vec(OpClamp{st, m, low, hig}, a_all));                               // :1084-1088
    TRY(ax(m, Ad));
    TRY(drv.vec(OpBoxInit{st, Ad, B, m, g}, a_all));
    TRY(drv.scal(FinBoxInit{p.step}));
    auto pfp = [&](double resid, int t) -> int { return Pfp(inst, m, resid, &para, n, t); };
    int rc = drv.run([&]() -> int {
        TRY(drv.vec(OpPgStep{st, m, g, low, hig, mn, 0.0}, a_all));              // :1151-1155
        TRY(ax(mn, Ad));                                                         // :1157
        TRY(drv.vec(OpBoxUpdate{st, Ad, B, mn, m, g}, a_all));                   // :1159-1190
        TRY(drv.scal(FinBoxClose{}));
        return 0;
    }, Pfp != nullptr, pfp, LCG_REACHED_MAX_ITERATIONS, LCG_NAN_VALUE);
    int rc2 = hb.close(c.stream);
    return rc <= -2000 ? rc : (rc2 ? rc2 : rc);
}

static int solve_spg(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, const double *low,
                     const double *hig, int n, const lcg_para *param, void *inst, int mem)
{
    const lcg_para p = param ? *param : lcg_hip_default_parameters();
    if (n <= 0) return LCG_INVILAD_VARIABLE_SIZE;                               // lcg.cpp:1230-1241
    if (p.max_iterations < 0) return LCG_INVILAD_MAX_ITERATIONS;
    if (p.epsilon <= 0.0 || p.epsilon >= 1.0) return LCG_INVILAD_EPSILON;
    if (p.step <= 0.0) return LCG_INVALID_LAMBDA;
    if (p.sigma <= 0.0 || p.sigma >= 1.0) return LCG_INVALID_SIGMA;
    if (p.beta <= 0.0 || p.beta >= 1.0) return LCG_INVALID_BETA;
    if (p.maxi_m <= 0) return LCG_INVALID_MAXIM;
    if (!m || !B || !low || !hig) return LCG_INVALID_POINTER;
    TRY(ensure_init());
    Ctx &c = ctx();
    const size_t nb = sizeof(double) * n;
    HostBridge hb; TRY(hb.open(mem, m, B, nb, c.stream));
    Workspace ws; double *g, *Ad, *mn, *d, *dlow = nullptr, *dhig = nullptr;
    TRY(ws.get(g, nullptr, nb)); TRY(ws.get(Ad, nullptr, nb)); TRY(ws.get(mn, nullptr, nb)); TRY(ws.get(d, nullptr, nb));
    if (mem == LCG_HIP_MEM_HOST) {
        TRY(ws.get(dlow, nullptr, nb)); TRY(ws.get(dhig, nullptr, nb));
        HIPCHK(hipMemcpyAsync(dlow, low, nb, hipMemcpyHostToDevice, c.stream));
        HIPCHK(hipMemcpyAsync(dhig, hig, nb, hipMemcpyHostToDevice, c.stream));
        low = dlow; hig = dhig;
    }
    Driver drv(c, n, false, p.max_iterations, p.epsilon, p.abs_diff);
    drv.user_cb = Afp != lcg_hip_csr_ax;
    TRY(drv.init_state(global_rows_of(c, n, (const void *)Afp, inst)));
    DevState *st = c.state;
    const uintptr_t a_all = al(m) | al(B) | al(g) | al(Ad) | al(mn) | al(d) | al(low) | al(hig);
    lcg_para para = p;
    auto ax = [&](const double *x, double *y) { return drv.timed_ax([&] { Afp(inst, x, y, n); }); };

    TRY(drv.vec(OpClamp{st, m, low, hig}, a_all));
    TRY(ax(m, Ad));
    TRY(drv.vec(OpBoxInit{st, Ad, B, m, g}, a_all));
    TRY(drv.scal(FinBoxInit{p.step}));
    DevState h;
    TRY(drv.read_state(h));
    auto leave = [&](int code) { drv.finish(h); int rc2 = hb.close(c.stream); return rc2 ? rc2 : code; };
    if (h.status == ST_ALREADY) {
        if (Pfp) Pfp(inst, m, h.residual, &para, n, 0);
        return leave(LCG_ALREADY_OPTIMIZIED);
    }
    std::vector<double> qm((size_t)p.maxi_m, -1e+30);                           // :1301-1305
    qm[0] = h.s[8];
    for (;;) {                                                                   // host-driven: see file header
        if (Pfp && Pfp(inst, m, h.residual, &para, n, h.t)) return leave(LCG_STOP);
        if (h.residual <= p.epsilon) return leave(LCG_CONVERGENCE);
        if (p.max_iterations > 0 && h.t + 1 > p.max_iterations) return leave(LCG_REACHED_MAX_ITERATIONS);
        const int t = h.t + 1;
        TRY(drv.vec(OpSpgDir{st, m, g, low, hig, d, mn, 0.0}, a_all));           // :1339-1349
        TRY(drv.scal(FinStore1<9>{}));                                           // g.d
        double ak = 1.0;
        const double qmax = *std::max_element(qm.begin(), qm.end());             // :1366-1370
        for (;;) {
            TRY(ax(mn, Ad));                                                     // :1351 / :1381
            TRY(drv.vec(OpSpgQ{st, mn, Ad, B}, a_all));
            TRY(drv.scal(FinStore1<8>{}));
            TRY(drv.read_state(h));
            const double qk = h.s[8], amod = p.sigma * ak * h.s[9];             // :1359-1364
            if (!(qk > qmax + amod)) { qm[(size_t)((t + 1) % p.maxi_m)] = qk; break; }   // :1372, :1394
            ak *= p.beta;                                                        // :1374
            TRY(drv.vec(OpSpgTrial{st, m, d, mn, ak}, a_all));
        }
        TRY(drv.vec(OpBoxUpdate{st, Ad, B, mn, m, g}, a_all));                   // :1396-1427
        TRY(drv.scal(FinBoxClose{}));
        TRY(drv.read_state(h));
    }
}

} // namespace lcgh

using namespace lcgh;

extern "C" {

int lcg_hip_solver_constrained(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, const double *low,
                               const double *hig, int n, const lcg_para *param, void *instance, int solver_id, int mem)
{
    if (solver_id == LCG_SPG) return solve_spg(Afp, Pfp, m, B, low, hig, n, param, instance, mem);   // lcg.cpp:121-140
    return solve_pg(Afp, Pfp, m, B, low, hig, n, param, instance, mem);
}

int lcg_hip_set2box(int n, const double *low, const double *hig, double *a)
{
    if (n <= 0 || !low || !hig || !a) return LCG_HIP_E_ARG;
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    const bool v2 = ((al(low) | al(hig) | al(a)) & 15) == 0;
    const int g = grid_for(v2 ? (n + 1) / 2 : n);
    OpClamp op{nullptr, a, low, hig};
    if (v2) hipLaunchKernelGGL((k_vec<OpClamp, true>), dim3(g), dim3(VB), 0, c.stream, op, (long)n, c.partials);
    else hipLaunchKernelGGL((k_vec<OpClamp, false>), dim3(g), dim3(VB), 0, c.stream, op, (long)n, c.partials);
    HIPCHK(hipGetLastError());
    return 0;
}

} // extern "C"
