// csr_tiled.hip -- A.x for matrices whose rows draw their columns at random from a BAND (or any pattern in which a few
// thousand rows share a few hundred KB of x, but neighbouring rows do not share cache lines).
//
// On such a matrix the row-block kernels of csr.hip find x in the L2 -- and still crawl: every 8-byte gather moves a
// 128-byte line from L2 to the CU (the 10M-row row-random band, W = 131072: 42 GB through the L2 for 4 GB of matrix,
// 1.39 ms = 0.37 of the HBM peak; non-temporal or L1-bypassing gathers change nothing).  The two-pass binned product
// (csr_binned.hip) would stream 28.5 B per entry.  Here x is staged the way the matrix is: a workgroup of four
// wavefronts owns 4 x 1024 rows (their sums in LDS, one wavefront per 1024 rows as in k_bin_reduce) and walks the
// column tiles its rows touch; per tile it copies 4096 entries of x (32 KB, coalesced, from L2) into LDS and every
// wavefront then streams its rows' entries of that tile -- val (8 B) and a 32-bit (row, column) pair, coalesced --
// gathering x from LDS at word granularity and adding into its row sums with ds_add_f64.  HBM sees 12 B per entry,
// the L2 sees whole lines only, and the random accesses stay inside the CU.
//
// Order of the stream: [chunk of 1024 rows][tile][entries in CSR order]; a group is padded to an even length (padding:
// row 0xFFFF).  A row is summed by one wavefront in stream order: same bits from call to call and from plan to plan
// (k_tl_place ranks entries without atomics).  As in csr_binned.hip products are rounded before they are added.
// Worth it while a (workgroup, tile) pair holds >~ 500 entries (tile copies are L2 traffic: pairs x 32 KB); the plan
// builder measures that and refuses otherwise (scattered columns: the binned product; structured ones: csr.hip).
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "devcommon.hpp"

namespace lcgh {

int device_exclusive_scan(int n, const int *counts, int *rowptr, hipStream_t s, long *total);   // csr.hip

constexpr int TL_MAXNW = 8;         // most wavefronts (chunks) per workgroup
constexpr int TL_C = 4096;          // columns per tile (32 KB of LDS)
constexpr int TL_C_LOG2 = 12;
constexpr int TL_MAXSPAN = 2048;    // most tiles one workgroup's rows may span (LDS histogram of the builder: 4 x 8 KB)
constexpr unsigned TL_PAD = 0xFFFFu;

typedef double v2d_t __attribute__((ext_vector_type(2)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));

struct TiledPlan {
    int rw = 1024, nw = 4;          // rows per wavefront (sums in LDS), wavefronts (chunks) per workgroup
    int n_rows = 0, nwg = 0;
    long n_cols = 0, entries = 0, pairs = 0;
    double *val2 = nullptr;
    unsigned *idx2 = nullptr;       // (row in chunk) << 16 | (column in tile)
    int *tmin = nullptr, *nspan = nullptr, *sofs = nullptr;     // per workgroup: first tile, tiles spanned, offset into gstart
    int *gstart = nullptr;          // [(sofs[g] + lt) * 4 + w]: first entry of group (wavefront w, local tile lt), relative to the chunk's bin
    int *binofs = nullptr;          // [4 * nwg + 1] first entry of each chunk's bin
    size_t bytes = 0;
};

// ---------------------------------------------------------------------------------------------- the product
// PUSH (sharded rows, direct exchange): the first pp.nblocks blocks of the grid carry this rank's boundary entries of x to the
// neighbours (devcommon.hpp: push_block) while the rest multiply -- as in the row-block kernels of csr.hip.
template <int TL_RW, int TL_NW, int UN, bool PUSH = false>
__global__ __launch_bounds__(TL_NW * 64) void k_tile_spmv(int n, int nwg, const int *__restrict__ tmin, const int *__restrict__ nspan,
                                                          const int *__restrict__ sofs, const int *__restrict__ gstart,
                                                          const int *__restrict__ binofs, const double *__restrict__ val2,
                                                          const unsigned *__restrict__ idx2, const double *__restrict__ x,
                                                          long n_cols, double *__restrict__ y, const int *done, PushPlan pp)
{
    static_assert(!PUSH || TL_NW * 64 == VB, "push_block moves PUSH_CHUNK entries with VB threads");
    if (PUSH && (int)blockIdx.x < pp.nblocks) { push_block(pp, blockIdx.x); return; }
    const int bid = PUSH ? blockIdx.x - pp.nblocks : blockIdx.x;
    __shared__ __attribute__((aligned(16))) double sx[TL_C];
    __shared__ __attribute__((aligned(16))) double ys[TL_NW][TL_RW];
    if (done && *done) return;
    // the wavefront's index as a SCALAR: the group bounds below are then scalar loads (s_load), issued one tile
    // ahead -- as vector loads they were a dependent L2 round trip in front of every tile's requests
    // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2).  Consecutive row blocks share all
    // but a few of their tiles, so each XCD is given a CONTIGUOUS eighth of the row blocks: its L2 then fetches an eighth
    // of x (plus the band) instead of all of it.  Speed only: any placement gives the same result.
    const int per_xcd = (nwg + 7) >> 3;
    const int g = (bid & 7) * per_xcd + (bid >> 3);
    if (g >= nwg) return;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = g * TL_NW + w;
    const long row0 = (long)chunk * TL_RW;
    double *my = ys[w];
    {
        v2d_t z; z.x = 0.0; z.y = 0.0;
#pragma unroll
        for (int i = 0; i < TL_RW / 128; i++) reinterpret_cast<v2d_t *>(my)[i * 64 + l] = z;
    }
    const int t0 = tmin[g], ns = nspan[g];
    const int *gs = gstart + (long)sofs[g] * TL_NW;
    const long bin = binofs[chunk];
    const bool x16 = (((uintptr_t)x) & 15) == 0;
    // bounds of this wavefront's group of tile 0 and whether any wavefront has entries in it; refreshed one tile ahead
    int a_nx = 0, b_nx = 0, any_nx = 0;
    if (ns > 0) {
        a_nx = gs[w]; b_nx = gs[TL_NW + w];
        any_nx = 0;
#pragma unroll
        for (int q = 0; q < TL_NW; q++) any_nx += gs[TL_NW + q] - gs[q];
    }
    for (int lt = 0; lt < ns; lt++) {
        const int a = a_nx, b = b_nx, any = any_nx;
        if (lt + 1 < ns) {      // scalar loads for the NEXT tile: their latency runs beside this tile's work
            a_nx = gs[(lt + 1) * TL_NW + w]; b_nx = gs[(lt + 2) * TL_NW + w];
            any_nx = 0;
#pragma unroll
            for (int q = 0; q < TL_NW; q++) any_nx += gs[(lt + 2) * TL_NW + q] - gs[(lt + 1) * TL_NW + q];
        }
        // a tile none of the wavefronts has entries in is skipped by all of them (uniform decision)
        if (any == 0) continue;
        // the group's first UN steps are requested BEFORE the tile is copied: their HBM latency runs beside the copy
        v2d_t va[UN]; v2u_t ia[UN];
        const long p0 = bin + a + 2 * l, pe = bin + b;
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long p = p0 + 128L * u;
            const long pc = p < pe ? p : bin;           // branch-free: lanes past the group re-read the bin's first pair
            va[u] = *reinterpret_cast<const v2d_t *>(val2 + pc);
            ia[u] = *reinterpret_cast<const v2u_t *>(idx2 + pc);
        }
        __syncthreads();                                // the previous tile's readers are done
        // the slice of x (issuing these loads in front of the barrier as well was measured: no gain, 16 more registers)
        const long c0 = (long)(t0 + lt) << TL_C_LOG2;
        const int cn = (int)min((long)TL_C, n_cols - c0);
        constexpr int NL = TL_C / 2 / (TL_NW * 64);
        v2d_t xr[NL];
        if (x16 && cn == TL_C) {
#pragma unroll
            for (int q = 0; q < NL; q++) xr[q] = *reinterpret_cast<const v2d_t *>(x + c0 + 2 * (q * TL_NW * 64 + tid));
        } else {
#pragma unroll
            for (int q = 0; q < NL; q++) {
                const int i = 2 * (q * TL_NW * 64 + tid);
                xr[q].x = x[c0 + (i < cn ? i : 0)];
                xr[q].y = x[c0 + (i + 1 < cn ? i + 1 : 0)];
            }
        }
#pragma unroll
        for (int q = 0; q < NL; q++) reinterpret_cast<v2d_t *>(sx)[q * TL_NW * 64 + tid] = xr[q];
        __syncthreads();
        {
            double x0[UN], x1[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) { x0[u] = sx[ia[u].x & (TL_C - 1)]; x1[u] = sx[ia[u].y & (TL_C - 1)]; }
#pragma unroll
            for (int u = 0; u < UN; u++) {
                if (p0 + 128L * u < pe) {
                    __hip_atomic_fetch_add(my + (ia[u].x >> 16), va[u].x * x0[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((ia[u].y >> 16) != TL_PAD)
                        __hip_atomic_fetch_add(my + (ia[u].y >> 16), va[u].y * x1[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        // groups longer than UN steps (128 entries each): the rest, four steps in flight
        for (long p = p0 + 128L * UN; p < pe; p += 128L * 4) {
            v2d_t vb[4]; v2u_t ib[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const long q = p + 128L * u;
                const long qc = q < pe ? q : bin;
                vb[u] = *reinterpret_cast<const v2d_t *>(val2 + qc);
                ib[u] = *reinterpret_cast<const v2u_t *>(idx2 + qc);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (p + 128L * u < pe) {
                    const unsigned i0 = ib[u].x, i1 = ib[u].y;
                    __hip_atomic_fetch_add(my + (i0 >> 16), vb[u].x * sx[i0 & 0xffffu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((i1 >> 16) != TL_PAD)
                        __hip_atomic_fetch_add(my + (i1 >> 16), vb[u].y * sx[i1 & 0xffffu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    const int cnt = (int)max(0L, min((long)TL_RW, (long)n - row0));
#pragma unroll 4
    for (int i = l; i < cnt; i += 64) y[row0 + i] = my[i];
}

// ---------------------------------------------------------------------------------------------- building the plan
// per workgroup (4096 rows): smallest and largest tile its entries touch
__global__ __launch_bounds__(256) void k_tl_span(int n, int rows_per_wg, long n_cols, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                 int *tmin, int *nspan, int *flags)
{
    __shared__ int slo[4], shi[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    const long r0 = (long)g * rows_per_wg;
    const int r1 = (int)min((long)n, r0 + rows_per_wg);
    const int k0 = rowptr[r0], k1 = rowptr[r1];
    int lo = 0x7fffffff, hi = -1;
    bool bad = false;
    for (int k = k0 + tid; k < k1; k += 256) {
        const int c = col[k];
        if (c < 0 || c >= n_cols) bad = true; else { lo = min(lo, c); hi = max(hi, c); }
    }
    if (bad) flags[0] = 1;
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); }
    if ((tid & 63) == 0) { slo[tid >> 6] = lo; shi[tid >> 6] = hi; }
    __syncthreads();
    if (tid == 0) {
        lo = min(min(slo[0], slo[1]), min(slo[2], slo[3])); hi = max(max(shi[0], shi[1]), max(shi[2], shi[3]));
        const int a = hi >= 0 ? lo >> TL_C_LOG2 : 0, b = hi >= 0 ? hi >> TL_C_LOG2 : -1;
        tmin[g] = a; nspan[g] = b - a + 1;
        atomicMax(&flags[1], b - a + 1);
    }
}

// per workgroup: entries per (wavefront, local tile), padded to even, scanned per wavefront into group starts;
// gstart carries nspan + 1 items per workgroup (the last one = the bins' lengths)
__global__ __launch_bounds__(64 * TL_MAXNW) void k_tl_count(int n, int TL_RW, int TL_NW, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                           const int *__restrict__ tmin, const int *__restrict__ nspan,
                                                           const int *__restrict__ sofs, int *gstart, int *binlen, unsigned long long *pairs)
{
    extern __shared__ int hist[];       // [nw][ns]
    const int g = blockIdx.x, tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int t0 = tmin[g], ns = nspan[g];
    for (int i = tid; i < TL_NW * ns; i += 64 * TL_NW) hist[i] = 0;
    __syncthreads();
    const long r0 = (long)(g * TL_NW + w) * TL_RW;
    const int ra = (int)min((long)n, r0), rb = (int)min((long)n, r0 + TL_RW);
    const int k0 = rowptr[ra], k1 = rowptr[rb];
    for (int k = k0 + l; k < k1; k += 64) atomicAdd(&hist[w * ns + (col[k] >> TL_C_LOG2) - t0], 1);
    __syncthreads();
    // each wavefront scans its own row of the histogram (serial over chunks of 64 tiles)
    int *out = gstart + (long)sofs[g] * TL_NW;
    int run = 0, used = 0;
    for (int b = 0; b < ns; b += 64) {
        const int lt = b + l;
        const int c = lt < ns ? (hist[w * ns + lt] + 1) & ~1 : 0;
        int inc = c;
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off, 64); if (l >= off) inc += v; }
        if (lt < ns) out[lt * TL_NW + w] = run + inc - c;
        run += __shfl(inc, 63, 64);
    }
    if (l == 0) { out[ns * TL_NW + w] = run; binlen[g * TL_NW + w] = run; }
    // (workgroup, tile) pairs that hold anything: what the tile copies will cost
    if (w == 0) {
        for (int lt = l; lt < ns; lt += 64) { int h = 0; for (int q = 0; q < TL_NW; q++) h += hist[q * ns + lt]; used += h > 0; }
        for (int off = 32; off > 0; off >>= 1) used += __shfl_down(used, off, 64);
        if (l == 0) atomicAdd(pairs, (unsigned long long)used);
    }
}

// one wavefront per chunk walks its entries in CSR order, 64 at a time; rank inside the group = entries of that tile
// placed so far + lower lanes of the batch with the same tile (one ballot per distinct tile): no atomics, no sort
__global__ __launch_bounds__(64) void k_tl_place(int n, int TL_RW, int TL_NW, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                 const double *__restrict__ val, const int *__restrict__ tmin,
                                                 const int *__restrict__ nspan, const int *__restrict__ sofs,
                                                 const int *__restrict__ gstart, const int *__restrict__ binofs,
                                                 double *val2, unsigned *idx2)
{
    extern __shared__ int cur[];        // [ns]
    const int chunk = blockIdx.x, l = threadIdx.x;
    const int g = chunk / TL_NW, w = chunk % TL_NW;
    const int t0 = tmin[g], ns = nspan[g];
    for (int t = l; t < ns; t += 64) cur[t] = 0;
    const long r0l = (long)chunk * TL_RW;
    const int r0 = (int)min((long)n, r0l), r1 = (int)min((long)n, r0l + TL_RW);
    if (r0 >= r1) return;
    const int k0 = rowptr[r0], k1 = rowptr[r1];
    const int *gs = gstart + (long)sofs[g] * TL_NW;
    const long bin = binofs[chunk];
    const unsigned long long below = l == 0 ? 0ull : (~0ull >> (64 - l));
    for (int kb = k0; kb < k1; kb += 64) {
        const int k = kb + l;
        const bool active = k < k1;
        const int c = active ? col[k] : 0;
        const int t = active ? (c >> TL_C_LOG2) - t0 : -1;
        int rank = 0;
        unsigned long long todo = __ballot(active);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int tl = __shfl(t, leader, 64);
            const unsigned long long same = __ballot(t == tl);
            const int base = cur[tl];
            if (t == tl) rank = base + __popcll(same & below);
            if (l == leader) cur[tl] = base + __popcll(same);
            todo &= ~same;
        }
        if (active) {
            int lo = r0, hi = r1 - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (rowptr[mid] <= k) lo = mid; else hi = mid - 1; }
            const long p = bin + gs[t * TL_NW + w] + rank;
            val2[p] = val[k];
            idx2[p] = ((unsigned)(lo - r0) << 16) | (unsigned)(c & (TL_C - 1));
        }
    }
}

static void plan_free(TiledPlan *T)
{
    if (!T) return;
    for (void *p : {(void *)T->val2, (void *)T->idx2, (void *)T->tmin, (void *)T->nspan, (void *)T->sofs, (void *)T->gstart, (void *)T->binofs})
        if (p) (void)hipFree(p);
    delete T;
}

void tiled_free(CsrPart &P)
{
    if (P.tl_plan) { if (ctx().inited) (void)hipDeviceSynchronize(); plan_free(static_cast<TiledPlan *>(P.tl_plan)); }
    P.tl_plan = nullptr; P.tl_state = 0;
}

// min_fill: least mean number of entries per (workgroup, tile) pair for the plan to be worth building (0 = build anyway)
struct TiledShape { int rw, nw, un; };
static TiledShape tiled_shape()
{   // LCG_HIP_TILED_SHAPE (A/B runs): 0 = 4 x 1024 rows (default: 0.82-0.85 ms on the 10M-row row-random band), 1 = 8 x 512 rows
    // (0.92-0.94), 2 = 8 x 1024 rows, one workgroup per CU (0.93-0.95).  Also measured and removed: 4 x 640 and 4 x 512 rows (three
    // workgroups per CU: 0.95-1.01), 4 x 1280 and 4 x 1536 rows (fewer tile copies per row: 0.90 / 0.94); a pipeline across tiles with the
    // next tile's requests in flight during this tile's adds (1.05 ms), and double-buffered tiles with requests two tiles ahead in a
    // one-workgroup-per-CU shape (1.18 ms) -- two independent workgroups per CU overlap their phases better than either.
    static const int v = [] { const char *e = std::getenv("LCG_HIP_TILED_SHAPE"); return e ? atoi(e) : 0; }();
    switch (v) {
    case 1: return {512, 8, 3};
    case 2: return {1024, 8, 6};
    default: return {1024, 4, 6};
    }
}

static int plan_build(const CsrPart &P, hipStream_t s, double min_fill, TiledPlan **out, const char **why)
{
    *out = nullptr;
    const int n = P.n_rows;
    const long n_cols = P.n_cols;
    *why = "empty matrix or unknown column count";
    if (n <= 0 || n_cols <= 0 || P.nnz <= 0) return 0;
    const TiledShape shape = tiled_shape();
    const int TL_RW = shape.rw, TL_NW = shape.nw;
    const int nwg = (n + TL_RW * TL_NW - 1) / (TL_RW * TL_NW);
    TiledPlan *T = new TiledPlan();
    T->n_rows = n; T->nwg = nwg; T->n_cols = n_cols; T->rw = TL_RW; T->nw = TL_NW;
    int *flags = nullptr, *binlen = nullptr;
    unsigned long long *pairs = nullptr;
    auto cleanup = [&](int rc) {
        for (void *p : {(void *)flags, (void *)binlen, (void *)pairs}) if (p) (void)hipFree(p);
        if (rc || !*out) { plan_free(T); *out = nullptr; }
        return rc;
    };
    *why = "a HIP call failed while the plan was built (lcg_hip_last_error)";
#define TCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return cleanup(fail(e_, #call, __FILE__, __LINE__)); } while (0)
    TCHK(hipMalloc(&T->tmin, sizeof(int) * (size_t)nwg));
    TCHK(hipMalloc(&T->nspan, sizeof(int) * (size_t)nwg));
    TCHK(hipMalloc(&T->sofs, sizeof(int) * ((size_t)nwg + 1)));
    TCHK(hipMalloc(&T->binofs, sizeof(int) * ((size_t)nwg * TL_NW + 1)));
    TCHK(hipMalloc(&binlen, sizeof(int) * (size_t)nwg * TL_NW));
    static_assert(TL_MAXSPAN * TL_MAXNW * sizeof(int) <= 65536, "builder histogram must fit the default dynamic LDS");
    TCHK(hipMalloc(&flags, 2 * sizeof(int)));
    TCHK(hipMalloc(&pairs, sizeof(unsigned long long)));
    TCHK(hipMemsetAsync(flags, 0, 2 * sizeof(int), s));
    TCHK(hipMemsetAsync(pairs, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_tl_span, dim3(nwg), dim3(256), 0, s, n, TL_RW * TL_NW, n_cols, P.rowptr, P.col, T->tmin, T->nspan, flags);
    TCHK(hipGetLastError());
    int hflags[2] = {0, 0};
    TCHK(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    TCHK(hipStreamSynchronize(s));
    if (hflags[0]) { *why = "a column index lies outside [0, n_cols)"; return cleanup(0); }
    if (hflags[1] > TL_MAXSPAN) { *why = "4096 rows span more than 2048 column tiles (scattered columns: see the binned product)"; return cleanup(0); }
    // gstart holds nspan + 1 items per workgroup
    int *span1 = nullptr;
    TCHK(hipMalloc(&span1, sizeof(int) * (size_t)nwg));
    {
        std::vector<int> h((size_t)nwg);
        hipError_t e = hipMemcpyAsync(h.data(), T->nspan, sizeof(int) * (size_t)nwg, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        for (int &v : h) v += 1;
        if (e == hipSuccess) e = hipMemcpyAsync(span1, h.data(), sizeof(int) * (size_t)nwg, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { hipFree(span1); return cleanup(fail(e, "tile spans", __FILE__, __LINE__)); }
    }
    long items = 0;
    int rc = device_exclusive_scan(nwg, span1, T->sofs, s, &items);
    hipFree(span1);
    if (rc) return cleanup(rc);
    TCHK(hipMalloc(&T->gstart, sizeof(int) * (size_t)items * TL_NW));
    hipLaunchKernelGGL(k_tl_count, dim3(nwg), dim3(64 * TL_NW), sizeof(int) * TL_NW * (size_t)hflags[1], s, n, TL_RW, TL_NW, P.rowptr, P.col,
                       T->tmin, T->nspan, T->sofs, T->gstart, binlen, pairs);
    TCHK(hipGetLastError());
    long total = 0;
    rc = device_exclusive_scan(nwg * TL_NW, binlen, T->binofs, s, &total);
    if (rc) return cleanup(rc);
    unsigned long long hp = 0;
    TCHK(hipMemcpyAsync(&hp, pairs, sizeof hp, hipMemcpyDeviceToHost, s));
    TCHK(hipStreamSynchronize(s));
    T->entries = total; T->pairs = (long)hp;
    if (total <= 0 || total > 0x7fffffffL || hp == 0) { *why = "stream length out of range"; return cleanup(0); }
    if ((double)P.nnz / (double)hp < min_fill) { *why = "too few entries per (workgroup, tile) pair: the tile copies would cost more than the gathers they replace"; return cleanup(0); }
    const size_t e2 = (size_t)total + 1024;
    TCHK(hipMalloc(&T->val2, sizeof(double) * e2));
    TCHK(hipMalloc(&T->idx2, sizeof(unsigned) * e2));
    TCHK(hipMemsetAsync(T->val2, 0, sizeof(double) * e2, s));
    TCHK(hipMemsetAsync(T->idx2, 0xff, sizeof(unsigned) * e2, s));
    hipLaunchKernelGGL(k_tl_place, dim3(nwg * TL_NW), dim3(64), sizeof(int) * (size_t)hflags[1], s, n, TL_RW, TL_NW, P.rowptr, P.col, P.val,
                       T->tmin, T->nspan, T->sofs, T->gstart, T->binofs, T->val2, T->idx2);
    TCHK(hipGetLastError());
    TCHK(hipStreamSynchronize(s));
    T->bytes = e2 * 12 + (size_t)items * TL_NW * 4 + (size_t)nwg * (12 + 4 * TL_NW);
#undef TCHK
    *out = T;
    *why = "ready";
    return cleanup(0);
}

// 1 = plan ready, 0 = this matrix does not use the tiled product, < 0 = failure
int tiled_ready(const CsrPart &P, hipStream_t s, double min_fill)
{
    if (P.tl_state != 0) return P.tl_state > 0 ? 1 : 0;
    if (P.tl_plan) { P.tl_state = 1; return 1; }       // built earlier (e.g. under a forced mode): reuse
    P.tl_state = -1;
    TiledPlan *T = nullptr;
    int rc = plan_build(P, s, min_fill, &T, &P.tl_why);
    if (std::getenv("LCG_HIP_DEBUG_BINNED"))
        std::fprintf(stderr, "[lcg_hip] tiled plan for %d x %ld, %ld entries: %s (rc %d)\n", P.n_rows, (long)P.n_cols, (long)P.nnz, P.tl_why, rc);
    if (rc) { (void)hipGetLastError(); return rc; }
    if (!T) return 0;
    P.tl_plan = T; P.tl_state = 1;
    return 1;
}

int tiled_launch(const CsrPart &P, const double *x, double *y, hipStream_t s, const int *done, const PushPlan *push)
{
    const TiledPlan *T = static_cast<const TiledPlan *>(P.tl_plan);
    if (!T) return fail(hipErrorInvalidValue, "tiled A.x without a plan", __FILE__, __LINE__);
    if (push) {
        if (T->rw != 1024 || T->nw != 4) return fail(hipErrorInvalidValue, "pushing blocks need the 4 x 1024 shape", __FILE__, __LINE__);
        hipLaunchKernelGGL((k_tile_spmv<1024, 4, 6, true>), dim3(8 * ((T->nwg + 7) / 8) + push->nblocks), dim3(VB), 0, s, T->n_rows, T->nwg, T->tmin,
                           T->nspan, T->sofs, T->gstart, T->binofs, T->val2, T->idx2, x, T->n_cols, y, done, *push);
        HIPCHK(hipGetLastError());
        return 0;
    }
#define TL_LAUNCH(RW, NW, UN)                                                                                                  \
    hipLaunchKernelGGL((k_tile_spmv<RW, NW, UN>), dim3(8 * ((T->nwg + 7) / 8)), dim3(NW * 64), 0, s, T->n_rows, T->nwg, T->tmin, T->nspan, T->sofs, T->gstart, \
                       T->binofs, T->val2, T->idx2, x, T->n_cols, y, done, PushPlan())
    if (T->rw == 512 && T->nw == 8) TL_LAUNCH(512, 8, 3);
    else if (T->rw == 1024 && T->nw == 8) TL_LAUNCH(1024, 8, 6);
    else TL_LAUNCH(1024, 4, 6);
#undef TL_LAUNCH
    HIPCHK(hipGetLastError());
    return 0;
}

// bytes per product by construction: HBM stream + y, and the tile copies (L2 traffic) separately
long tiled_traffic_bytes(const CsrPart &P)
{
    const TiledPlan *T = static_cast<const TiledPlan *>(P.tl_plan);
    return T ? 12L * T->entries + 8L * T->n_rows + 8L * T->n_cols : 0;
}
long tiled_tile_copy_bytes(const CsrPart &P)
{
    const TiledPlan *T = static_cast<const TiledPlan *>(P.tl_plan);
    return T ? T->pairs * (long)TL_C * 8 : 0;
}

} // namespace lcgh
