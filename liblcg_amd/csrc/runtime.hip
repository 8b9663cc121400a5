// runtime.hip -- library context (stream, reduction workspace, device state), error text,
// and the stand-alone BLAS-1 entry points of the C ABI.
#include <cmath>
#include <cstdarg>

#include "devcommon.hpp"

namespace lcgh {

Ctx &ctx()
{
    static Ctx c;   // one context per process, like the reference (no re-entrancy promised)
    return c;
}

int fail(hipError_t e, const char *what, const char *file, int line)
{
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    ctx().err = buf;
    (void)hipGetLastError();
    return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? LCG_HIP_E_NO_DEVICE : LCG_HIP_E_RUNTIME;
}

int ensure_init()
{
    Ctx &c = ctx();
    if (c.inited) return 0;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        c.err = "no HIP device: liblcg_hip has no CPU fallback";
        return LCG_HIP_E_NO_DEVICE;
    }
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    c.device = dev;
    // "Blocking" streams: they order themselves against the legacy default stream, which is where a
    // host program that does not think about streams (a plain hipMemset, PyTorch's default stream)
    // puts its work -- so `y.zero_(); lcg_hip_spmv(A, x, y)` cannot race.  The two streams still
    // run concurrently with each other (gather/compute overlap).
    HIPCHK(hipStreamCreateWithFlags(&c.own_stream, hipStreamDefault));
    HIPCHK(hipStreamCreateWithFlags(&c.comm_stream, hipStreamDefault));
    if (!c.stream) c.stream = c.own_stream;
    // fork/join of the two streams inside A.x: same device on both sides, so no system-scope fence
    // (with it each hand-off cost ~11 us of cache write-back on the critical path)
    HIPCHK(hipEventCreateWithFlags(&c.ev_a, hipEventDisableTiming | hipEventDisableSystemFence));
    HIPCHK(hipEventCreateWithFlags(&c.ev_b, hipEventDisableTiming | hipEventDisableSystemFence));
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipMalloc(&c.partials_pair[i], sizeof(double) * MAXR * MAXG));
        HIPCHK(hipMalloc(&c.state_pair[i], sizeof(DevState)));
        HIPCHK(hipMemset(c.state_pair[i], 0, sizeof(DevState)));
    }
    c.partials = c.partials_pair[0];
    HIPCHK(hipMalloc(&c.ax_partials, sizeof(double) * 2 * AXP_CAP));
    c.state = c.state_pair[0];
    HIPCHK(hipHostMalloc((void **)&c.hstat, sizeof(HostStatus), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void **)&c.hstat_dev, c.hstat, 0));
    HIPCHK(hipHostMalloc((void **)&c.scratch_host, sizeof(double) * 64, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void **)&c.state_stage, sizeof(DevState), hipHostMallocDefault));
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipHostMalloc((void **)&c.snap[i], sizeof(DevState), hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipHostGetDevicePointer((void **)&c.snap_dev[i], c.snap[i], 0));
        HIPCHK(hipEventCreateWithFlags(&c.snap_ev[i], hipEventDisableTiming));
    }
    if (hipMemGetInfo(&c.mem_free_at_init, &c.mem_total) != hipSuccess) { (void)hipGetLastError(); c.mem_free_at_init = c.mem_total = 0; }
    if (const char *e = std::getenv("LCG_HIP_PLACE")) { const int v = atoi(e); if (v >= -1 && v <= 1 && c.place_mode == -1) c.place_mode = v; }
    c.inited = true;
    return 0;
}

// ---- stand-alone reductions -------------------------------------------------------------------
struct OpDotPlain {
    static constexpr int NR = 1, SKIP = SKIP_NEVER;
    DevState *st; const double *a, *b;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc) { acc[0] += dotp(ld<T>(a, i), ld<T>(b, i)); }
};
struct OpDot2Plain {    // acc0 = a.b, acc1 = a.a
    static constexpr int NR = 2, SKIP = SKIP_NEVER;
    DevState *st; const double *a, *b;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const T x = ld<T>(a, i);
        acc[0] += dotp(x, ld<T>(b, i)); acc[1] += dotp(x, x);
    }
};
template <bool CONJ>
struct OpCDot {      // complex: sum a_i b_i (CONJ = false) or sum conj(a_i) b_i
    static constexpr int NR = 2, SKIP = SKIP_NEVER;
    DevState *st; const double *a, *b;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *acc)
    {
        const double2 x = ld<double2>(a, i), y = ld<double2>(b, i);
        if (CONJ) { acc[0] += x.x * y.x + x.y * y.y; acc[1] += x.x * y.y - x.y * y.x; }
        else { acc[0] += x.x * y.x - x.y * y.y; acc[1] += x.x * y.y + x.y * y.x; }
    }
};
struct OpAxpy {
    static constexpr int NR = 0, SKIP = SKIP_NEVER;
    DevState *st; double alpha; const double *x; double *y;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *) { st_(y, i, vadd(ld<T>(y, i), alpha * ld<T>(x, i))); }
};
struct OpScal {
    static constexpr int NR = 0, SKIP = SKIP_NEVER;
    DevState *st; double alpha; double *x;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *) { st_(x, i, alpha * ld<T>(x, i)); }
};
struct OpCAxpy {
    static constexpr int NR = 0, SKIP = SKIP_NEVER;
    DevState *st; double2 alpha; const double *x; double *y;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *) { st_(y, i, cfma(alpha, ld<double2>(x, i), ld<double2>(y, i))); }
};
template <int NRR> struct FinCopy {
    static constexpr int NR = NRR;
    double *out;    // device-visible (pinned host) destination
    __device__ void operator()(DevState *, const double *sum) const
    {
        for (int r = 0; r < NRR; r++) out[r] = sum[r];
    }
};

template <class Op, int NRR>
static int reduce_to_host(Op op, long n, bool cplx, uintptr_t align_or, double *result)
{
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    const bool v2 = !cplx && (align_or & 15) == 0;
    const int g = grid_for(cplx ? n : (v2 ? (n + 1) / 2 : n));
    if (v2) hipLaunchKernelGGL((k_vec<Op, true>), dim3(g), dim3(VB), 0, c.stream, op, n, c.partials);
    else hipLaunchKernelGGL((k_vec<Op, false>), dim3(g), dim3(VB), 0, c.stream, op, n, c.partials);
    double *dst = nullptr;
    HIPCHK(hipMalloc(&dst, sizeof(double) * NRR));
    FinCopy<NRR> fin{dst};
    PartCount pc; pc.all(g);
    hipLaunchKernelGGL((k_scal<FinCopy<NRR>>), dim3(1), dim3(VB), 0, c.stream, fin, c.partials, pc, c.state, SC_FUSED, XgBox());
    if (comm_active()) { rc = comm_allreduce(dst, NRR, c.stream); if (rc) { hipFree(dst); return rc; } }
    hipError_t e = hipMemcpyAsync(c.scratch_host, dst, sizeof(double) * NRR, hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    hipFree(dst);
    if (e != hipSuccess) return fail(e, "reduction", __FILE__, __LINE__);
    for (int r = 0; r < NRR; r++) result[r] = c.scratch_host[r];
    return 0;
}

template <class Op> static int plain_vec(Op op, long n, bool cplx, uintptr_t align_or)
{
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    const bool v2 = !cplx && (align_or & 15) == 0;
    const int g = grid_for(cplx ? n : (v2 ? (n + 1) / 2 : n));
    if (v2) hipLaunchKernelGGL((k_vec<Op, true>), dim3(g), dim3(VB), 0, c.stream, op, n, c.partials);
    else hipLaunchKernelGGL((k_vec<Op, false>), dim3(g), dim3(VB), 0, c.stream, op, n, c.partials);
    HIPCHK(hipGetLastError());
    return 0;
}

} // namespace lcgh

using namespace lcgh;

extern "C" {

int lcg_hip_init(int device)
{
    if (device >= 0) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return fail(e, "hipSetDevice", __FILE__, __LINE__);
    }
    return ensure_init();
}

int lcg_hip_set_stream(void *s)
{
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    c.stream = s ? static_cast<hipStream_t>(s) : c.own_stream;
    return 0;
}

void *lcg_hip_get_stream(void) { return ensure_init() ? nullptr : ctx().stream; }

int lcg_hip_synchronize(void)
{
    int rc = ensure_init(); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx().stream));
    // a product over the direct exchange cannot return its own failure (it only enqueues): a timed-out
    // wait for a neighbour leaves the failure flag up, and this is where a caller learns of it
    if (lcg_hip_p2p_status() < 0) {
        ctx().err = "direct exchange: a neighbour's data or sums did not arrive in time (results since then are invalid)";
        return LCG_HIP_E_COMM;
    }
    return 0;
}

int lcg_hip_memcpy(void *dst, const void *src, uint64_t bytes, int kind)
{
    int rc = ensure_init(); if (rc) return rc;
    const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, k, ctx().stream));
    HIPCHK(hipStreamSynchronize(ctx().stream));
    return 0;
}

const char *lcg_hip_last_error(void) { return ctx().err.c_str(); }

lcg_para lcg_hip_default_parameters(void)
{
    lcg_para p = {0, 1e-6, 0, 1e-6, 1.0, 0.95, 0.9, 10};   // util.h:153
    return p;
}
clcg_para clcg_hip_default_parameters(void)
{
    clcg_para p = {0, 1e-6, 0};                             // util.h:278
    return p;
}

int lcg_hip_last_iterations(void) { return ctx().last_iters; }
double lcg_hip_last_residual(void) { return ctx().last_residual; }
double lcg_hip_last_ax_mean_us(void)
{
    Ctx &c = ctx();
    if (c.prof_pending >= 2) {      // read the events of the last solve now, not inside its timed region
        hipStreamSynchronize(c.stream);
        double tot = 0.0;
        for (int i = 0; i + 1 < c.prof_pending; i += 2) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, c.prof_ev[i], c.prof_ev[i + 1]);
            tot += ms;
        }
        c.last_ax_mean_us = 1e3 * tot / (c.prof_pending / 2);
        c.prof_pending = 0;
    }
    return c.last_ax_mean_us;
}

int lcg_hip_trim(void)
{
    Ctx &c = ctx();
    if (!c.inited) return 0;
    (void)hipDeviceSynchronize();
    // (slots of an arena go together: the arena is given back when none of its slots is in use)
    auto arena_busy = [&](void *a) { for (auto &s : c.scratch) if (s.arena == a && s.busy) return true; return false; };
    for (auto it = c.scratch.begin(); it != c.scratch.end();) {
        if (it->busy || (it->arena && arena_busy(it->arena))) { ++it; continue; }
        if (!it->arena || it->p == it->arena) (void)hipFree(it->p);
        c.released_bytes += it->bytes;
        it = c.scratch.erase(it);
    }
    c.forget_places();       // (addresses may come back as other memory)
    return 0;
}
int lcg_hip_pool_info(int *vectors, int64_t *bytes, int *arena_slots)
{
    Ctx &c = ctx();
    int64_t b = 0; int a = 0;
    for (auto &s : c.scratch) { b += (int64_t)s.bytes; a += s.arena != nullptr; }
    if (vectors) *vectors = (int)c.scratch.size();
    if (bytes) *bytes = b;
    if (arena_slots) *arena_slots = a;
    return 0;
}
int lcg_hip_pool_add_arena_for_test(uint64_t slot_bytes, int slots)
{
    if (slot_bytes == 0 || slots <= 0) return LCG_HIP_E_ARG;
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    const size_t slot = ((size_t)slot_bytes + 255) & ~(size_t)255;
    void *base = nullptr;
    HIPCHK(hipMalloc(&base, slot * (size_t)slots));
    for (int i = 0; i < slots; i++) c.scratch.push_back({reinterpret_cast<double *>(static_cast<char *>(base) + (size_t)i * slot), slot, false, base});
    return 0;
}
int lcg_hip_placement_tune_for_test(uint64_t stream_min_bytes, uint64_t chunk_bytes, int max_chunks, double wall_ms, uint64_t hold_max_bytes,
                                    int force_find_at, int allow_shared)
{
    int rc = ensure_init(); if (rc) return rc;
    Ctx::PlaceTune t;       // zeros / negatives: the production value
    if (stream_min_bytes) t.stream_min = (size_t)stream_min_bytes;
    if (chunk_bytes) t.chunk = ((size_t)chunk_bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (max_chunks > 0) t.max_chunks = max_chunks;
    if (wall_ms > 0.0) t.wall_ms = wall_ms;
    if (hold_max_bytes) t.hold_max = (size_t)hold_max_bytes;
    t.force_find_at = force_find_at;
    if (allow_shared) t.shared_min = ~(size_t)0;
    if (allow_shared == 1) { t.released_max = ~(size_t)0; t.predict = false; }    // (a test walks many times in one process; 2 keeps the fresh-allocator rule)
    ctx().place_tune = t;
    return 0;
}
int lcg_hip_last_placement_walk(int *chunks, double *wall_ms, int64_t *held_bytes, int *found, const char **ended)
{
    Ctx &c = ctx();
    if (chunks) *chunks = c.walk_chunks;
    if (wall_ms) *wall_ms = c.walk_ms;
    if (held_bytes) *held_bytes = (int64_t)c.walk_held;
    if (found) *found = c.walk_found;
    if (ended) *ended = c.walk_end;
    return c.walks_made;
}
int lcg_hip_set_placement(int mode)
{
    if (mode < -1 || mode > 1) return LCG_HIP_E_ARG;
    ctx().place_mode = mode;
    return 0;
}
int lcg_hip_last_placement(int *timed, int *moved, double *us_as_allocated, double *us_as_placed)
{
    Ctx &c = ctx();
    if (timed) *timed = c.place_timed;
    if (moved) *moved = c.place_moved;
    if (us_as_allocated) *us_as_allocated = c.place_us_first;
    if (us_as_placed) *us_as_placed = c.place_us_chosen;
    return 0;
}
int lcg_hip_last_ax_calls(void) { return ctx().last_ax_calls; }
int lcg_hip_last_launches(int *vector_passes, int *scalar_steps, int *rank_reductions, int *products)
{
    Ctx &c = ctx();
    if (vector_passes) *vector_passes = c.cnt_vec;
    if (scalar_steps) *scalar_steps = c.cnt_scal;
    if (rank_reductions) *rank_reductions = c.cnt_allreduce;
    if (products) *products = c.cnt_ax;
    return 0;
}
int lcg_hip_last_finisher_steps(void) { return 0; }      // (the experiment it counted is retired: lcg_hip.h)

int lcg_hip_set_cg_schedule(int schedule)
{
    if (schedule < LCG_HIP_CG_AUTO || schedule > LCG_HIP_CG_ONE_REDUCTION) return LCG_HIP_E_ARG;
    ctx().cg_schedule = schedule;      // no device needed: a host-side choice
    return 0;
}

int lcg_hip_set_profiling(int on)
{
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    c.profile = on != 0;
    c.profile_every = on > 1 ? on : 1;
    c.ax_seq = 0;
    if (c.profile && c.prof_ev.empty()) {
        c.prof_ev.resize(2 * 4096);
        // default flags on purpose: a hipEventDisableSystemFence event is not a marker of its own, it
        // rides on the previous command, and the elapsed time then starts with THAT kernel (A.x read
        // 35 us long -- the direction update before it -- against rocprofv3's kernel trace)
        for (auto &ev : c.prof_ev) HIPCHK(hipEventCreate(&ev));
    }
    c.prof_used = 0; c.prof_pending = 0;
    return 0;
}

int lcg_hip_set_shadow_seed(unsigned seed) { ctx().shadow_seed = seed; return 0; }
int lcg_hip_set_shadow_vector(const double *v, int n)
{
    if (!v || n <= 0) { ctx().shadow_vec.clear(); return 0; }
    ctx().shadow_vec.assign(v, v + 2 * (size_t)n);
    return 0;
}

int lcg_hip_dot(int n, const double *a, const double *b, double *result)
{
    if (n <= 0 || !a || !b || !result) return LCG_HIP_E_ARG;
    return reduce_to_host<OpDotPlain, 1>(OpDotPlain{nullptr, a, b}, n, false, (uintptr_t)a | (uintptr_t)b, result);
}
int lcg_hip_spmv_dot(lcg_hip_csr_t A, const double *x, double *y, const double *u, double *result2)
{
    if (!A || !x || !y || !u || A->is_complex) return LCG_HIP_E_ARG;
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    int slots = 0;
    const int f = csr_ax_dot(A, x, y, u, 1, c.ax_partials, &slots, c.stream, nullptr);
    if (f < 0) return f;
    if (f == 0) {       // this matrix / kernel family keeps product and reduction apart
        rc = lcg_hip_spmv(A, x, y); if (rc) return rc;
        if (!result2) return 0;
        return reduce_to_host<OpDot2Plain, 2>(OpDot2Plain{nullptr, y, u}, A->n_rows, false, (uintptr_t)y | (uintptr_t)u, result2);
    }
    if (!result2) return 0;
    PartCount pc; pc.all(0); pc.axp = c.ax_partials; pc.ax_n = slots; pc.ax_row = 0; pc.ax_yy = 1;
    double *dst = nullptr;
    HIPCHK(hipMalloc(&dst, sizeof(double) * 2));
    FinCopy<2> fin{dst};
    hipLaunchKernelGGL((k_scal<FinCopy<2>>), dim3(1), dim3(VB), 0, c.stream, fin, c.partials, pc, c.state, SC_FUSED, XgBox());
    hipError_t e = hipMemcpyAsync(c.scratch_host, dst, sizeof(double) * 2, hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    hipFree(dst);
    if (e != hipSuccess) return fail(e, "spmv_dot", __FILE__, __LINE__);
    result2[0] = c.scratch_host[0]; result2[1] = c.scratch_host[1];
    return 0;
}
int lcg_hip_nrm2(int n, const double *a, double *result)
{
    int rc = lcg_hip_dot(n, a, a, result);
    if (!rc) *result = std::sqrt(*result);
    return rc;
}
int clcg_hip_dot(int n, const double *a, const double *b, double *result2)
{
    if (n <= 0 || !a || !b || !result2) return LCG_HIP_E_ARG;
    return reduce_to_host<OpCDot<false>, 2>(OpCDot<false>{nullptr, a, b}, n, true, 0, result2);
}
int clcg_hip_inner(int n, const double *a, const double *b, double *result2)
{
    if (n <= 0 || !a || !b || !result2) return LCG_HIP_E_ARG;
    return reduce_to_host<OpCDot<true>, 2>(OpCDot<true>{nullptr, a, b}, n, true, 0, result2);
}
int lcg_hip_axpy(int n, double alpha, const double *x, double *y)
{
    if (n <= 0 || !x || !y) return LCG_HIP_E_ARG;
    return plain_vec(OpAxpy{nullptr, alpha, x, y}, n, false, (uintptr_t)x | (uintptr_t)y);
}
int lcg_hip_scal(int n, double alpha, double *x)
{
    if (n <= 0 || !x) return LCG_HIP_E_ARG;
    return plain_vec(OpScal{nullptr, alpha, x}, n, false, (uintptr_t)x);
}
int clcg_hip_axpy(int n, const double *alpha2, const double *x, double *y)
{
    if (n <= 0 || !alpha2 || !x || !y) return LCG_HIP_E_ARG;
    return plain_vec(OpCAxpy{nullptr, make_double2(alpha2[0], alpha2[1]), x, y}, n, true, 0);
}

} // extern "C"
