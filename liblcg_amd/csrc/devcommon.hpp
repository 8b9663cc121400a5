// devcommon.hpp -- device-side building blocks shared by the real and complex kernels.
//
// Shape of the BLAS-1 side of the iteration (all HBM-bound, no MFMA):
//   * k_vec<Op>: one fused element-wise pass over the vectors an iteration step touches,
//     16 B per lane per access, grid-stride over <= MAXG blocks, with up to MAXR running sums
//     reduced wave (shuffle) -> block (LDS) -> one partial per block in a fixed table.
//   * k_scal<Fin>: one block that sums the partials in a fixed order (bit-reproducible, no
//     atomics), then a single lane runs the solver's scalar recurrence on DevState.  When
//     the rows are sharded over ranks the same kernel runs in two halves around an RCCL
//     all-reduce of DevState::red.
// Coefficients (alpha, beta, ...) are read by the vector kernels from DevState, so the host
// never waits for a scalar inside the loop.
#pragma once

#include <cstdlib>

#include "internal.hpp"

namespace lcgh {

// Sum over the 64 lanes of a wavefront with DPP moves (row shifts inside the rows of 16, then the two row broadcasts): six
// steps of two v_mov_dpp + one v_add_f64.  __shfl_down goes through the LDS crossbar (ds_bpermute, two per double and step):
// ~1 us per reducing kernel with four running sums -- a fifth of a kernel on the launch-bound systems.  The total is valid in
// the LAST lane (WSUM_LANE); lanes a step has no source for add +0.0.  Fixed order: the same bits every time.
constexpr int WSUM_LANE = 63;
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_get(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_get<0x111, 0xf>(v);    // row_shr:1
    v += dpp_get<0x112, 0xf>(v);    // row_shr:2
    v += dpp_get<0x114, 0xf>(v);    // row_shr:4
    v += dpp_get<0x118, 0xf>(v);    // row_shr:8: lane 15 of every row holds the row's sum
    v += dpp_get<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
    v += dpp_get<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return v;
}

// Reduce NR per-thread accumulators over the block; thread r (< NR) ends up with sum r and
// writes it to partials[r*MAXG + blockIdx.x].
template <int NR>
__device__ __forceinline__ void block_reduce_store(double *acc, double *partials)
{
    __shared__ double sh[NR][VB / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < NR; r++) {
        double v = wave_sum(acc[r]);
        if (lane == WSUM_LANE) sh[r][w] = v;
    }
    __syncthreads();
    if (threadIdx.x < NR) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < VB / 64; k++) v += sh[threadIdx.x][k];
        partials[threadIdx.x * MAXG + blockIdx.x] = v;
    }
}

// skip rules shared by every kernel of the loop
//  SKIP_DONE: no-op once the stop flag is set
//  SKIP_DIR : the direction update that closes an iteration still runs in the iteration that
//             converged (as the reference does before it re-tests at the loop head), but not
//             after a NaN stop and not in later, already-void iterations
enum { SKIP_NEVER = 0, SKIP_DONE = 1, SKIP_DIR = 2 };

__device__ __forceinline__ bool should_skip(const DevState *st, int mode)
{
    if (mode == SKIP_NEVER) return false;
    const int done = st->done;
    if (mode == SKIP_DONE) return done != 0;
    if (!done) return false;
    return st->status == ST_NAN || st->it != st->t;
}

// ---- generic fused vector pass ----------------------------------------------------------
// Op provides:  static constexpr int NR, SKIP;  DevState *st;  void prep();  (loads scalars)
//               template<class T> void apply(long i, double *acc);   T = double2 (two reals /
//               one complex) or double (scalar tail, real only)
template <class Op, bool VEC2>
__device__ __forceinline__ void vec_body(Op &op, long n, double *partials)
{
    constexpr int NRA = Op::NR > 0 ? Op::NR : 1;
    double acc[NRA];
#pragma unroll
    for (int r = 0; r < NRA; r++) acc[r] = 0.0;
    op.prep();
    const long stride = (long)gridDim.x * VB;
    const long tid = (long)blockIdx.x * VB + threadIdx.x;
    if (VEC2) {
        const long n2 = n >> 1;
        for (long i = tid; i < n2; i += stride) op.template apply<double2>(i, acc);
        if ((n & 1) && tid == 0) op.template apply<double>(n - 1, acc);
    } else {
        for (long i = tid; i < n; i += stride) op.template apply<double>(i, acc);
    }
    if (Op::NR > 0) block_reduce_store<NRA>(acc, partials);
}

template <class Op, bool VEC2>
__global__ __launch_bounds__(VB) void k_vec(Op op, long n, double *partials)
{
    if (should_skip(op.st, Op::SKIP)) return;
    vec_body<Op, VEC2>(op, n, partials);
}

template <class T> __device__ __forceinline__ T ld(const double *p, long i);
template <> __device__ __forceinline__ double ld<double>(const double *p, long i) { return p[i]; }
template <> __device__ __forceinline__ double2 ld<double2>(const double *p, long i)
{
    return reinterpret_cast<const double2 *>(p)[i];
}
__device__ __forceinline__ void st_(double *p, long i, double v) { p[i] = v; }
__device__ __forceinline__ void st_(double *p, long i, double2 v) { reinterpret_cast<double2 *>(p)[i] = v; }
// non-temporal twins: vectors an iteration touches once (the iterate m, the product A.d after the update) should not push the vectors
// the NEXT kernels need (d = the product's x, g) out of the L2 / Infinity Cache
typedef double v2d_nt __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ T ldnt(const double *p, long i);
template <> __device__ __forceinline__ double ldnt<double>(const double *p, long i) { return __builtin_nontemporal_load(p + i); }
template <> __device__ __forceinline__ double2 ldnt<double2>(const double *p, long i)
{
    const v2d_nt v = __builtin_nontemporal_load(reinterpret_cast<const v2d_nt *>(p) + i);
    return make_double2(v.x, v.y);
}
__device__ __forceinline__ void stnt(double *p, long i, double v) { __builtin_nontemporal_store(v, p + i); }
__device__ __forceinline__ void stnt(double *p, long i, double2 v)
{
    v2d_nt w; w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<v2d_nt *>(p) + i);
}
// (nt is uniform per launch: the host sets it for systems whose vectors cannot stay in the caches anyway -- solvers_real.hip: stream_vectors)
template <class T> __device__ __forceinline__ T ldp(const double *p, long i, bool nt) { return nt ? ldnt<T>(p, i) : ld<T>(p, i); }
__device__ __forceinline__ void stp(double *p, long i, double v, bool nt) { if (nt) stnt(p, i, v); else st_(p, i, v); }
__device__ __forceinline__ void stp(double *p, long i, double2 v, bool nt) { if (nt) stnt(p, i, v); else st_(p, i, v); }

// y of the row-block product, by cache policy (internal.hpp: y_store_policy)
__device__ __forceinline__ void store_y(double *p, double v, int how)
{
    if (how == 0) *p = v;
    else if (how == 1) __builtin_nontemporal_store(v, p);
    else if (how == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// real helpers on 1 or 2 packed values
__device__ __forceinline__ double dotp(double a, double b) { return a * b; }
__device__ __forceinline__ double dotp(double2 a, double2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ double nanflag(double a) { return a != a ? 1.0 : 0.0; }
__device__ __forceinline__ double nanflag(double2 a) { return (a.x != a.x || a.y != a.y) ? 1.0 : 0.0; }
__device__ __forceinline__ double2 operator*(double s, double2 v) { return make_double2(s * v.x, s * v.y); }
__device__ __forceinline__ double2 vadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 vsub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double vadd(double a, double b) { return a + b; }
__device__ __forceinline__ double vsub(double a, double b) { return a - b; }
__device__ __forceinline__ double2 vmul(double2 a, double2 b) { return make_double2(a.x * b.x, a.y * b.y); }
__device__ __forceinline__ double vmul(double a, double b) { return a * b; }
__device__ __forceinline__ double2 vneg(double2 a) { return make_double2(-a.x, -a.y); }
__device__ __forceinline__ double vneg(double a) { return -a; }
// a vector that counts as all zeros when `zero` is set (the product of an all-zero initial guess, which was never made)
template <class T> __device__ __forceinline__ T ldz(const double *p, long i, bool zero);
template <> __device__ __forceinline__ double ldz<double>(const double *p, long i, bool zero) { return zero ? 0.0 : p[i]; }
template <> __device__ __forceinline__ double2 ldz<double2>(const double *p, long i, bool zero)
{
    return zero ? make_double2(0.0, 0.0) : reinterpret_cast<const double2 *>(p)[i];
}

// complex helpers (double2 = re, im)
__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cfma(double2 a, double2 b, double2 c)   // a*b + c
{
    return make_double2(fma(a.x, b.x, fma(-a.y, b.y, c.x)), fma(a.x, b.y, fma(a.y, b.x, c.y)));
}
// robust complex division (Smith's algorithm): avoids the overflow of |b|^2
__device__ __forceinline__ double2 cdiv(double2 a, double2 b)
{
    if (fabs(b.x) >= fabs(b.y)) {
        const double r = b.y / b.x, d = b.x + b.y * r;
        return make_double2((a.x + a.y * r) / d, (a.y - a.x * r) / d);
    }
    const double r = b.x / b.y, d = b.x * r + b.y;
    return make_double2((a.x * r + a.y) / d, (a.y * r - a.x) / d);
}
__device__ __forceinline__ double cnorm(double2 a) { return a.x * a.x + a.y * a.y; }

// value ops shared by the A.x kernels (csr.hip, comm.hip)
__device__ __forceinline__ double vzero(double) { return 0.0; }
__device__ __forceinline__ double2 vzero(double2) { return make_double2(0.0, 0.0); }
__device__ __forceinline__ double mac(double a, double x, double acc) { return fma(a, x, acc); }
__device__ __forceinline__ double2 mac(double2 a, double2 x, double2 acc) { return cfma(a, x, acc); }
__device__ __forceinline__ double shfl_down_v(double v, int off, int w) { return __shfl_down(v, off, w); }
__device__ __forceinline__ double2 shfl_down_v(double2 v, int off, int w)
{
    return make_double2(__shfl_down(v.x, off, w), __shfl_down(v.y, off, w));
}

// ---- scalar step ----------------------------------------------------------------------------
// mode 0: reduce partials and run fin (single GPU)   1: reduce only -> st->red
// mode 2: fin only, sums taken from st->red (after the all-reduce)
// mode 3: reduce, exchange the sums with every peer through the mailboxes, run fin -- ONE kernel
//         where the RCCL path needs reduce | ncclAllReduce | fin
enum { SC_FUSED = 0, SC_REDUCE = 1, SC_FIN = 2, SC_XGMI = 3 };

// Direct all-reduce (sum) of NR doubles held in LDS by one block per rank.  Lane q < P writes this
// rank's sums and then the call's sequence number into slot [parity][me] of peer q's mailbox
// (release, system scope), waits for slot [parity][q] of its own mailbox to show the same number
// (acquire) and fetches peer q's sums.  The totals are then added in RANK order, so every rank
// obtains the same bits -- which the lock-step loop (driver.hpp) relies on.  Two parities suffice:
// a rank can finish call k+1 only after every peer has entered k+1, i.e. finished reading call k.
// A contribution that does not arrive within timeout_ticks raises *fail; the call returns false.
// 16-byte packets, one global_store_dwordx4 / global_load_dwordx4 each, system scope, uncached memory: {value low word, tag, value high
// word, tag} with tag = the low 32 bits of the call's sequence number.  The value needs no flag behind it -- the sender fires its
// packets and goes on, the receiver polls the packets themselves -- and a packet is accepted only when BOTH tags are the call's:
// an aligned 16-byte access has been one transaction on every gfx950 link observed, but that is not an architectural promise, and
// over xGMI a packet torn between its 8-byte halves (new tag beside the value of call k - 2 of the same parity) would hand one rank
// other bits than its peers and take the lock-step loop apart.  With a tag in each naturally aligned 8-byte half -- the
// unit that IS written whole -- a torn packet shows one stale tag and is polled again.  (A slot is rewritten every second call, so a
// stale tag differs from the expected one by 2: the 32-bit tag cannot be mistaken before 2^32 calls.)
typedef unsigned int xg_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void xg_store_packet(double *slot, double v, unsigned long long k)
{
    const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
    xg_v4u w; w.x = (unsigned)vb; w.y = (unsigned)k; w.z = (unsigned)(vb >> 32); w.w = (unsigned)k;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(slot), "v"(w) : "memory");
}
// true when the packet in the slot is call k's, whole
__device__ __forceinline__ bool xg_load_packet(const double *slot, double *v, unsigned long long k)
{
    xg_v4u w;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(slot) : "memory");
    *v = __longlong_as_double((long long)((unsigned long long)w.x | ((unsigned long long)w.z << 32)));
    return w.y == (unsigned)k && w.w == (unsigned)k;
}

// The sequence number of the exchange a one-block kernel is about to make and the failure flag, requested by thread 0 at the
// kernel's START (beside the partial sums the kernel loads first) instead of in front of the exchange: one memory round trip less
// on the chain of a scalar step.
struct XgTicket { unsigned long long k = 0; int bad = 0; };
__device__ __forceinline__ XgTicket xg_begin(const XgBox &xb)
{
    XgTicket t;
    if (threadIdx.x == 0) { t.k = *xb.seq + 1; t.bad = *xb.fail; }
    return t;
}

template <int NR>
__device__ __forceinline__ bool xg_exchange(const XgBox &xb, const double *sums, double (*got)[XG_MAXP], XgTicket tk)
{
    static_assert(2 * NR <= XG_SLOT, "a slot holds one 16-byte packet per sum");
    __shared__ unsigned long long sq;
    __shared__ int bad;
    if (threadIdx.x == 0) { sq = tk.k; *xb.seq = tk.k; bad = tk.bad; }
    __syncthreads();
    if (bad) return false;      // an earlier exchange failed: the peers are gone, do not wait again
    const unsigned long long k = sq;
    const int par = (int)(k & 1);
    if ((int)threadIdx.x < xb.P) {
        const int q = threadIdx.x;
        double *dst = xb.peers[q] + (size_t)(par * xb.P + xb.me) * XG_SLOT;
#pragma unroll
        for (int r = 0; r < NR; r++) xg_store_packet(dst + 2 * r, sums[r], k);
        const double *src = xb.mine + (size_t)(par * xb.P + q) * XG_SLOT;
        const long long t0 = wall_clock64();
        bool ok = true;
        double tmp[NR];
        for (;;) {
            bool all = true;
#pragma unroll
            for (int r = 0; r < NR; r++) all = xg_load_packet(src + 2 * r, &tmp[r], k) && all;
            if (all) break;
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > xb.timeout_ticks) { ok = false; break; }
        }
        if (!ok) { bad = 1; *xb.fail = 1; }
#pragma unroll
        for (int r = 0; r < NR; r++) got[r][q] = tmp[r];
    }
    __syncthreads();
    return bad == 0;
}

template <int NR>
__device__ __forceinline__ bool xg_allreduce(const XgBox &xb, double *sums, XgTicket tk)
{
    __shared__ double got[NR][XG_MAXP];
    if (!xg_exchange<NR>(xb, sums, got, tk)) return false;
    if ((int)threadIdx.x < NR) {
        double v = 0.0;
        for (int q = 0; q < xb.P; q++) v += got[threadIdx.x][q];
        sums[threadIdx.x] = v;
    }
    __syncthreads();
    return true;
}

// ---- direct neighbour exchange (dist mode 2) ----------------------------------------------------
// Block b of the pushing blocks copies its chunk of a segment of x into the neighbour's landing zone
// (peer-mapped memory) with write-through stores, waits until they are acknowledged and takes a
// ticket; the block that takes the last ticket raises this rank's flag word at every neighbour.
__device__ __forceinline__ void push_block(const PushPlan &pp, int b)
{
    int s = 0;
    while (s + 1 < pp.nseg && b >= pp.first_block[s + 1]) s++;
    if (pp.nseg > 0 && threadIdx.x < VB) {     // (workgroups wider than VB threads -- k_tile_spmv2 -- push with their first VB)
        const long off = (long)(b - pp.first_block[s]) * PUSH_CHUNK;
        const long cnt = min((long)PUSH_CHUNK, pp.count[s] - off);
        const double *src = pp.src[s] + off;
        double *dst = pp.dst[s] + off;
        // write-through stores at system scope: nothing of the neighbour's buffer stays dirty in this
        // GPU's caches, so "visible over there" only needs the stores to be acknowledged (vmcnt) --
        // a release fence would also write back every dirty line of the product running beside us
        // (measured: +12 us on the 100 us local product)
        constexpr int PER = PUSH_CHUNK / VB;
        double v[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) { const long i = threadIdx.x + (long)q * VB; v[q] = src[i < cnt ? i : 0]; }
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const long i = threadIdx.x + (long)q * VB;
            if (i < cnt) __hip_atomic_store(dst + i, v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // all of this lane's stores acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        // The ticket is relaxed on purpose (see above): every block's stores were acknowledged before its
        // ticket, so whoever draws the last ticket knows all of x has landed.  The flags that follow are
        // RELEASE stores at system scope: one lane per A.x pays for the ordering the memory model asks for
        // (the per-block release fences measured above stay out).
        const unsigned t = __hip_atomic_fetch_add(pp.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)pp.nblocks - 1u) {
            __hip_atomic_store(pp.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int f = 0; f < pp.nflag; f++)
                __hip_atomic_store(pp.flag[f], pp.seq, f == 0 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Every block of the receiving kernel: wait until all neighbours have raised their flag to this
// call's number (they only grow).  No cache maintenance here: flags and landing zone are uncached
// memory read with system-scope loads, and a block reads the landing zone only after it has seen the
// flags (an acquire fence per block cost 150 us on a 4000-block grid: it invalidates the whole L2).
// ACQ: the polling lanes close their wait with ONE acquire load of the flag at system scope (relaxed polls, then
// the acquire -- polling with acquire loads is 2-3x slower per hop).  k_recv (a few dozen blocks) does; the
// opt-in landing-direct product (thousands of blocks, every one of them waiting) stays relaxed, as measured.
template <bool ACQ = false>
__device__ __forceinline__ bool wait_flags(const WaitPlan &wp)
{
    __shared__ int wbad;
    if (threadIdx.x == 0) wbad = *wp.fail;
    __syncthreads();
    if (wbad) return false;
    if ((int)threadIdx.x < wp.n) {
        const long long t0 = wall_clock64();
        bool ok = true;
        while (__hip_atomic_load(wp.flag[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < wp.seq) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > wp.timeout_ticks) { wbad = 1; *wp.fail = 1; ok = false; break; }
        }
        if (ACQ && ok) (void)__hip_atomic_load(wp.flag[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    return wbad == 0;
}

// A receiving block in the tail of a product's grid (PushPlan::nrecv): k_recv's work (comm.hip) without its launch.  Relaxed
// polls and system-scope loads of the uncached landing zone, like the landing-direct product: an acquire here would invalidate
// the XCD's L2 under the product blocks still running beside it.
__device__ __forceinline__ void recv_block(const PushPlan &pp, int b)
{
    if (!wait_flags<false>(pp.wp)) {
        if (pp.rst && b == 0 && threadIdx.x == 0) { pp.rst->done = 1; pp.rst->status = ST_COMM; }
        return;
    }
    if (pp.rnseg == 0 || threadIdx.x >= VB) return;     // (workgroups wider than VB threads -- k_tile_spmv2 -- copy with their first VB)
    int s = 0;
    while (s + 1 < pp.rnseg && b >= pp.rfirst[s + 1]) s++;
    const long off = (long)(b - pp.rfirst[s]) * PUSH_CHUNK;
    const long cnt = min((long)PUSH_CHUNK, pp.rcount[s] - off);
    const double *src = pp.rsrc[s] + off;
    double *dst = pp.rdst[s] + off;
    constexpr int PER = PUSH_CHUNK / VB;
    double v[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const long i = threadIdx.x + (long)q * VB;
        v[q] = __hip_atomic_load(const_cast<double *>(src) + (i < cnt ? i : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const long i = threadIdx.x + (long)q * VB;
        if (i < cnt) dst[i] = v[q];
    }
}

__device__ __forceinline__ void publish(DevState *st)
{
    HostStatus *h = st->host;
    if (!h) return;         // a block's private copy of the state (k_vecf): only block 0 mirrors to the host
    // the kernel cannot retire before these PCIe writes are acknowledged (a few us): between stops
    // the mirror is refreshed only every (pub_mask + 1)-th body -- enough for the host's pacing
    if (!st->done && (st->it & st->pub_mask)) return;
    // posted writes over PCIe; no fence: the host only uses `it` to pace itself and `done`
    // to stop enqueuing early, and re-reads DevState with a real copy before it returns
    // (a system-scope fence here costs ~10 us per iteration)
    h->residual = st->residual;
    h->t = st->t;
    h->done = st->done;
    h->status = st->status;
    h->it = st->it;
}

// stop rule evaluated for the NEXT loop head (lcg.cpp:208-222): g2, m2 as the solver defines them
__device__ __forceinline__ void stop_rule(DevState *st, double g2, double m2)
{
    const double r = st->abs_diff ? sqrt(g2) / st->n_global : g2 / m2;
    st->residual = r;
    if (r <= st->eps) { st->done = 1; st->status = ST_CONVERGED; }
}

// Sum the G partials of each of NRA running sums (table row r = partials + r*MAXG) into sums[r] (LDS), every thread of the
// block taking part.  All NRA * MAXG/VB loads of a lane are issued before the first add (a dependent load-add chain made
// the one-block kernel take 16 us); fixed order => the same bits wherever and however often this runs.
template <int NRA>
__device__ __forceinline__ void reduce_partials(const double *partials, const PartCount &pc, double *sums)
{
    constexpr int PER = MAXG / VB;
    double acc[NRA], v[NRA][PER];
#pragma unroll
    for (int r = 0; r < NRA; r++)
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int j = threadIdx.x + q * VB;
            const int G = pc.g[r];
            const double x = partials[r * MAXG + (j < G ? j : 0)];     // branch-free: select after the load
            v[r][q] = j < G ? x : 0.0;
        }
#pragma unroll
    for (int r = 0; r < NRA; r++) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < PER; q++) t += v[r][q];
        acc[r] = t;
    }
    if (pc.ax_row >= 0) {       // sums carried by an A.x kernel: one partial per workgroup of that kernel, batches of 8 loads in flight
        double t0 = 0.0, t1 = 0.0;
        for (int j0 = 0; j0 < pc.ax_n; j0 += 8 * VB) {
            double a[8], b[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int j = j0 + threadIdx.x + q * VB;
                const int jj = j < pc.ax_n ? j : 0;
                a[q] = pc.axp[jj];
                b[q] = pc.ax_yy ? pc.axp[AXP_CAP + jj] : 0.0;
                if (j >= pc.ax_n) { a[q] = 0.0; b[q] = 0.0; }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) { t0 += a[q]; t1 += b[q]; }
        }
#pragma unroll
        for (int r = 0; r < NRA; r++) acc[r] += r == pc.ax_row ? t0 : (r == pc.ax_row + 1 && pc.ax_yy ? t1 : 0.0);
    }
    __shared__ double sh[NRA][VB / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < NRA; r++) {
        double t = wave_sum(acc[r]);
        if (lane == WSUM_LANE) sh[r][w] = t;
    }
    __syncthreads();
    if (threadIdx.x < NRA) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < VB / 64; k++) t += sh[threadIdx.x][k];
        sums[threadIdx.x] = t;
    }
    __syncthreads();
}

template <class Fin>
__global__ __launch_bounds__(VB) void k_scal(Fin fin, const double *partials, PartCount G, DevState *st, int mode, XgBox xb, DevState *snap = nullptr)
{   // snap (host-mapped pinned memory, or null): the state this step leaves, for the sharded loop's host (driver.hpp: run_lockstep)
    constexpr int NRA = Fin::NR > 0 ? Fin::NR : 1;
    constexpr int SW = (int)(sizeof(DevState) / 8);
    __shared__ double sums[NRA];
    __shared__ DevState L;
    XgTicket tk;
    if (mode == SC_XGMI && Fin::NR > 0) tk = xg_begin(xb);
    // the state the step works on is requested NOW, beside the partial sums, and the step runs on the copy in LDS: its loads would
    // otherwise be one more dependent round trip at the end of the chain (reduce -> exchange -> step)
    double sv = 0.0;
    if (mode != SC_REDUCE && (int)threadIdx.x < SW) sv = reinterpret_cast<const double *>(st)[threadIdx.x];
    if (mode != SC_FIN && Fin::NR > 0) {
        reduce_partials<NRA>(partials, G, sums);
        if (mode == SC_REDUCE && threadIdx.x < NRA) st->red[threadIdx.x] = sums[threadIdx.x];
    } else if (Fin::NR > 0) {
        if (threadIdx.x < NRA) sums[threadIdx.x] = st->red[threadIdx.x];
        __syncthreads();
    }
    if (mode == SC_XGMI && Fin::NR > 0) {
        if (!xg_allreduce<NRA>(xb, sums, tk)) {
            // every rank sees the failure of this or a later exchange and stops the same way
            if (threadIdx.x == 0) { st->done = 1; st->status = ST_COMM; if (snap) { snap->done = 1; snap->status = ST_COMM; } }
            return;
        }
    }
    if (mode == SC_REDUCE) return;
    if ((int)threadIdx.x < SW) reinterpret_cast<double *>(&L)[threadIdx.x] = sv;
    __syncthreads();
    if (threadIdx.x == 0) fin(&L, sums);
    __syncthreads();
    if ((int)threadIdx.x < SW) {
        const double w = reinterpret_cast<const double *>(&L)[threadIdx.x];
        reinterpret_cast<double *>(st)[threadIdx.x] = w;
        if (snap) reinterpret_cast<double *>(snap)[threadIdx.x] = w;     // posted writes; the host reads behind an event on this stream
    }
}

// ---- the body-closing step of plain CG's one-reduction schedule (solvers_real.hip) -------------------------------------------
enum { S_AK = 0, S_BK, S_WK, S_RHO /* g.g | z.r | r.r0 */, S_M2, S_G2 /* residual numerator */ };   // DevState::s slots of the real solvers
__device__ __forceinline__ double clamp1(double v) { return v < 1.0 ? 1.0 : v; }
struct FinCg1Close {    // the only scalar step of a body: counts it, closes it, prepares the next
    static constexpr int NR = 4;
    __device__ void operator()(DevState *st, const double *sum) const
    {
        st->it++;
        if (!st->done) {
            st->s[S_M2] = clamp1(sum[0]);
            if (sum[3] > 0.0 || sum[0] != sum[0]) {         // lcg.cpp:247-253
                st->t++;
                st->done = 1; st->status = ST_NAN;
            } else {
                const double rho_new = sum[1];
                const double bk = rho_new / st->s[S_RHO];                       // lcg.cpp:256
                st->s[S_AK] = rho_new / (sum[2] - bk * rho_new / st->s[S_AK]);  // lcg.cpp:235 with d.Ad as above
                st->s[S_BK] = bk;
                st->s[S_RHO] = rho_new;
                st->s[S_G2] = rho_new;
                st->t++;
                stop_rule(st, rho_new, st->s[S_M2]);
            }
        }
        publish(st);
    }
};


// ---- scalar step fused into the pass that consumes it (single GPU) ------------------------------------------------
// One CG iteration used to be six kernels, two of them one-block scalar steps on the critical path (5-8 us each plus a
// kernel boundary).  Here EVERY block of the consuming pass sums the <= 512 partials of the previous pass itself -- in
// reduce_partials' fixed order, so every block obtains the same bits -- and runs the scalar recurrence on a private copy
// of DevState in LDS; the pass reads its coefficients (and the stop flag) from that copy.  Block 0 alone commits the copy,
// to the OTHER buffer of a pair (`next`): no block ever reads a field another block is rewriting, and the kernels enqueued
// afterwards are handed `next` as their state.  Partial sums ping-pong between two tables for the same reason.
template <class Fin, class Op, bool VEC2>
__global__ __launch_bounds__(VB) void k_vecf(Fin fin, Op op, long n, const double *pin, PartCount G, double *pout, const DevState *cur, DevState *next)
{
    constexpr int NRF = Fin::NR > 0 ? Fin::NR : 1;
    __shared__ DevState L;
    __shared__ double sums[NRF];
    static_assert(sizeof(DevState) % 8 == 0, "DevState is copied in 8-byte words");
    {
        const double *src = reinterpret_cast<const double *>(cur);
        double *dst = reinterpret_cast<double *>(&L);
        for (int i = threadIdx.x; i < (int)(sizeof(DevState) / 8); i += VB) dst[i] = src[i];
    }
    reduce_partials<NRF>(pin, G, sums);      // ends with a barrier: L and sums are complete
    if (threadIdx.x == 0) {
        if (blockIdx.x != 0) L.host = nullptr;
        fin(&L, sums);
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        const double *src = reinterpret_cast<const double *>(&L);
        double *dst = reinterpret_cast<double *>(next);
        for (int i = threadIdx.x; i < (int)(sizeof(DevState) / 8); i += VB) dst[i] = src[i];
    }
    op.st = &L;
    if (should_skip(&L, Op::SKIP)) return;
    vec_body<Op, VEC2>(op, n, pout);
}

inline int grid_for(long n_items)
{
    // 512 blocks (two per CU) keep the BLAS-1 passes at their bandwidth (measured on the 10M-row system: 2048 / 1024 / 512
    // blocks = 171 / 172 / 166 us of BLAS-1 per CG iteration) and leave a quarter of the partials to re-reduce
    static const long cap = [] { const char *e = lab_env("LCG_HIP_MAXGRID"); long v = e ? atol(e) : 512; return v < 1 ? 1 : (v > MAXG ? (long)MAXG : v); }();
    long g = (n_items + VB - 1) / VB;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

} // namespace lcgh
