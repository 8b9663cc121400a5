// csr_tiled.hip -- A.x for matrices whose rows draw their columns at random from a BAND (or any pattern in which a few
// thousand rows share a few hundred KB of x, but neighbouring rows do not share cache lines).
//
// On such a matrix the row-block kernels of csr.hip find x in the L2 -- and still crawl: every 8-byte gather moves a
// 128-byte line from L2 to the CU (the 10M-row row-random band, W = 131072: 42 GB through the L2 for 4 GB of matrix,
// 1.39 ms = 0.37 of the HBM peak; non-temporal or L1-bypassing gathers change nothing).  The two-pass binned product
// (csr_binned.hip) would stream 28.5 B per entry.  Here x is staged the way the matrix is: a workgroup owns NW x 1024
// rows (their sums in LDS, one wavefront per 1024 rows as in k_bin_reduce) and walks the column tiles (2048 columns)
// its rows touch; per tile a slice of x goes into LDS and every wavefront streams its rows' entries of that tile,
// gathering x from LDS at word granularity and adding into its row sums with ds_add_f64.  The L2 sees whole lines
// only, and the random accesses stay inside the CU.
//
// Round 3 (k_tile_spmv2).  The round-2 kernel copied each tile through registers between two barriers and requested a
// tile's entries behind the previous tile's adds: 62 % of its wave cycles were parked at s_waitcnt / s_barrier
// (profiles/r03_tiled_sq.csv), 860 us on the 10M-row row-random band.  Now
//   * a workgroup is NW consumer wavefronts + ONE loader wavefront.  The loader copies tile j + 1 into the other half
//     of a double buffer by LDS-DMA (global_load_lds: no registers, and -- being a wavefront of its own -- no entry in
//     the consumers' in-order vmcnt queue) while the consumers work on tile j: one workgroup barrier per tile;
//   * a consumer never drains its loads: its stream is ONE contiguous piece of memory across all tiles, walked in steps
//     of 192 entries through a ring of D register sets, so the requests of the next D steps are in flight whatever tile
//     boundary, barrier or LDS phase the wavefront is in (770 us with the round-2 stream format);
//   * and, the kernel now being bound by the stream itself (knocking out every LDS access or the tile copies changes
//     it by < 5 %; a pure read of this box's HBM runs at 6.0 TB/s, the stream at 5.3), the stream is smaller: a step
//     is one 2 KB block -- three planes of 64 values and one plane of 64-bit words that hold three 21-bit
//     (row in chunk, column in tile) pairs -- 10.67 B per entry instead of 12.
//
// Order of the stream: [chunk of 1024 rows][tile][entries in CSR order], no padding inside a bin; a bin is padded to
// whole steps.  A row is summed by one wavefront in stream order: same bits from call to call and from plan to plan
// (k_tl_place ranks entries without atomics; lanes of one ds_add_f64 that meet in a row are serialised by the LDS in
// lane order on gfx950 -- verified by test, not promised by the ISA).  As in csr_binned.hip products are rounded
// before they are added.  Worth it while a (workgroup, tile) pair holds some hundreds of entries (tile copies are L2
// traffic); the plan builder measures that and refuses otherwise (scattered columns: the binned product; structured
// ones: csr.hip).
#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

#include "devcommon.hpp"

namespace lcgh {


typedef unsigned long long u64t;
constexpr int TL_MAXNW = 8;         // most consumer wavefronts (chunks) per workgroup
constexpr int TL_RW = 1024;         // rows per wavefront (8 KB of sums in LDS): 10 bits
constexpr int TL_TCL2 = 11;         // columns per tile: 2048 (16 KB of x per buffer half): 11 bits
constexpr int TL_TC = 1 << TL_TCL2;
constexpr int TL_STEP = 192;        // entries per step: 64 lanes x 3
constexpr int TL_BLK = 256;         // 64-bit words per step: 3 planes of values + 1 plane of packed pairs = 2 KB
constexpr int TL_SLACK = 24;        // readable steps behind the stream (the ring requests up to D steps past a bin's end)
constexpr unsigned TL_EMASK = 0x1FFFFFu;

typedef u64t v2u64_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

struct TiledPlan {
    int nw = 8;                     // consumer wavefronts (chunks) per workgroup
    int n_rows = 0, nwg = 0;
    long n_cols = 0, entries = 0, steps = 0, pairs = 0;
    u64t *stream = nullptr;         // [steps + TL_SLACK][TL_BLK]
    int *ntile = nullptr, *sofs = nullptr;      // per workgroup: tiles it has entries in, offset of its lists
    int *tl = nullptr;              // [sofs[g] + j]: the j-th of those tiles
    int *gstart = nullptr;          // [(sofs[g] + j) * nw + w]: first entry of group (wavefront w, j-th tile), relative to the chunk's bin; item ntile[g] = the bins' lengths
    int *bstep = nullptr;           // [nw * nwg + 1] first step of each chunk's bin
    size_t bytes = 0;
};

// ---------------------------------------------------------------------------------------------- the product
__device__ __forceinline__ void lds_barrier()
{   // orders LDS traffic only: the consumers' global loads stay in flight across it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// PUSH (sharded rows, direct exchange): the first pp.nblocks blocks of the grid carry this rank's boundary entries of x to the
// neighbours (devcommon.hpp: push_block) while the rest multiply -- as in the row-block kernels of csr.hip.
// DOT: every consumer also leaves its 1024 rows' share of y.u (and y.y) in dp.part[chunk] (dp.part[dp.stride + chunk]) -- the dot the
// Krylov loops take right after the product (csr.hip: csr_part_ax_dot folds the per-chunk sums).
template <int NW, int D, bool PUSH = false, bool NT = false, bool DOT = false>
__global__ __launch_bounds__((NW + 1) * 64) void k_tile_spmv2(int n, int nwg, const int *__restrict__ ntile, const int *__restrict__ sofs,
                                                              const int *__restrict__ tl, const int *__restrict__ gstart,
                                                              const int *__restrict__ bstep, const u64t *__restrict__ stream,
                                                              const double *__restrict__ x, long n_cols, double *__restrict__ y,
                                                              const int *done, PushPlan pp, DotPlan dp = DotPlan())
{
    constexpr int TC = TL_TC;
    if (PUSH && (int)blockIdx.x < pp.nblocks) { push_block(pp, blockIdx.x); return; }
    if (PUSH && pp.nrecv > 0 && (int)blockIdx.x >= (int)gridDim.x - pp.nrecv) { recv_block(pp, (int)blockIdx.x - ((int)gridDim.x - pp.nrecv)); return; }
    const int bid = PUSH ? blockIdx.x - pp.nblocks : blockIdx.x;
    __shared__ __attribute__((aligned(16))) double sx[2][TC];
    __shared__ __attribute__((aligned(16))) double ys[NW][TL_RW];
    if (done && *done) return;
    // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2).  Consecutive row blocks share all
    // but a few of their tiles, so each XCD is given a CONTIGUOUS eighth of the row blocks: its L2 then fetches an eighth
    // of x (plus the band) instead of all of it (numbering the workgroups along ONE front instead: 808 vs 775 us).
    // Speed only: any placement gives the same result.
    const int per_xcd = (nwg + 7) >> 3;
    const int g = (bid & 7) * per_xcd + (bid >> 3);
    if (g >= nwg) return;
    const int tid = threadIdx.x, l = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // a SCALAR: group bounds and tile numbers are then s_loads
    const int nt = ntile[g];
    const int so = sofs[g];
    if (w == NW) {
        // ---- the loader (a second loader wavefront sharing the tile was measured: 1175 -> 1105 us where the tiles come from beyond
        // the L2 -- W = 524288 --, nothing where they do not): tile j into half j & 1 of the buffer; barrier j tells the consumers it has landed, and tells the
        // loader that they are through with tile j - 1, whose half tile j + 1 goes into
        const bool x16 = (((uintptr_t)x) & 15) == 0;
        for (int j = 0; j < nt; j++) {
            const long c0 = (long)tl[so + j] << TL_TCL2;
            const int cn = (int)min((long)TC, n_cols - c0);
            double *dst = sx[j & 1];
            if (x16 && cn == TC) {
#pragma unroll
                for (int q = 0; q < TC / 128; q++)
                    __builtin_amdgcn_global_load_lds((glb_void_t *)(x + c0 + 2 * (q * 64 + l)), (lds_void_t *)(dst + q * 128), 16, 0, 0);
            } else {        // the matrix's last tile, or an x that is not 16-byte aligned (a caller's offset view): 4 bytes per lane,
                            // past the last column any valid address (never gathered)
                const float *xf = reinterpret_cast<const float *>(x + c0);
#pragma unroll 8
                for (int q = 0; q < TC / 32; q++) {
                    const int i = q * 64 + l;
                    const float *src = xf + ((i >> 1) < cn ? i : 0);
                    __builtin_amdgcn_global_load_lds((glb_void_t *)src, (lds_void_t *)(reinterpret_cast<float *>(dst) + q * 64), 4, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
        return;
    }
    // ---- a consumer: chunk = 1024 rows, sums in LDS
    const int chunk = g * NW + w;
    const long row0 = (long)chunk * TL_RW;
    double *my = ys[w];
    {
        double2 z; z.x = 0.0; z.y = 0.0;
#pragma unroll
        for (int i = 0; i < TL_RW / 128; i++) reinterpret_cast<double2 *>(my)[i * 64 + l] = z;
    }
    const int cnt = (int)max(0L, min((long)TL_RW, (long)n - row0));
    if (nt > 0) {
        const int *gs = gstart + (long)so * NW;
        const u64t *sb = stream + (long)bstep[chunk] * TL_BLK + 2 * l;
        const int len = gs[nt * NW + w];
        const int nsteps = (len + TL_STEP - 1) / TL_STEP;
        double v0[D], v1[D], v2[D]; u64t ia[D];
#define TL_LOAD(i, ss)                                                      \
    do {    /* two 16-byte loads per lane: (plane 0, plane 1) and (plane 2, packed pairs) */ \
        const v2u64_t *pb = reinterpret_cast<const v2u64_t *>(sb + (long)(ss) * TL_BLK);      \
        const v2u64_t q01 = NT ? __builtin_nontemporal_load(pb) : pb[0], q2i = NT ? __builtin_nontemporal_load(pb + 64) : pb[64]; \
        v0[i] = __longlong_as_double((long long)q01.x);                     \
        v1[i] = __longlong_as_double((long long)q01.y);                     \
        v2[i] = __longlong_as_double((long long)q2i.x);                     \
        ia[i] = q2i.y;                                                      \
    } while (0)
#pragma unroll
        for (int i = 0; i < D; i++) TL_LOAD(i, i);
        int j = 0;
        int a = 0, b = gs[NW + w];                              // tile 0's group
        int b_nx = nt > 1 ? gs[2 * NW + w] : b;                 // one tile ahead: the scalar load's latency runs beside the work
        const double *sxb = sx[0];
        lds_barrier();                                          // barrier 0: tile 0 has landed
        for (int s0 = 0; s0 < nsteps; s0 += D) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                const int s = s0 + i;
                const double a0 = v0[i], a1 = v1[i], a2 = v2[i];
                const u64t id = ia[i];
                const unsigned e0 = (unsigned)id & TL_EMASK, e1 = (unsigned)(id >> 21) & TL_EMASK, e2 = (unsigned)(id >> 42) & TL_EMASK;
                // this lane's three entries, relative to the bin: CONSECUTIVE positions, so that the entries a row has in one tile --
                // neighbours in the stream -- meet in one lane and not in one ds_add_f64 (same-address lanes serialise)
                const int q0 = TL_STEP * s + 3 * l, q1 = q0 + 1, q2 = q0 + 2;
                const int send = TL_STEP * (s + 1);
                for (;;) {
                    const bool m0 = q0 >= a && q0 < b, m1 = q1 >= a && q1 < b, m2 = q2 >= a && q2 < b;
                    // all gathers of the step before its adds; entries outside [a, b) belong to another tile (or to nobody)
                    const double x0 = sxb[m0 ? (e0 & (TC - 1)) : 0], x1 = sxb[m1 ? (e1 & (TC - 1)) : 0], x2 = sxb[m2 ? (e2 & (TC - 1)) : 0];
                    if (m0) __hip_atomic_fetch_add(my + (e0 >> TL_TCL2), a0 * x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (m1) __hip_atomic_fetch_add(my + (e1 >> TL_TCL2), a1 * x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (m2) __hip_atomic_fetch_add(my + (e2 >> TL_TCL2), a2 * x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (b > send || j + 1 >= nt) break;
                    // this wavefront is through with tile j
                    j++;
                    a = b; b = b_nx;
                    if (j + 2 <= nt) b_nx = gs[(j + 2) * NW + w];
                    sxb = sx[j & 1];
                    lds_barrier();
                }
                // the set is free again: request step s + D into the SAME registers (requested before the adds the new values
                // would need a second set and a copy at the loop's end -- which waits for every load in flight).  Past the
                // bin's end these read the next bin / the slack: unused.
                TL_LOAD(i, s + D);
            }
        }
#undef TL_LOAD
        while (j + 1 < nt) { j++; lds_barrier(); }              // tiles in which this wavefront has nothing left
    }
    if (DOT) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll 4
        for (int i = l; i < cnt; i += 64) {
            const double v = my[i];
            y[row0 + i] = v;
            a0 = fma(v, dp.u[row0 + i], a0);
            a1 = fma(v, v, a1);
        }
        a0 = wave_sum(a0);
        if (l == WSUM_LANE) dp.part[chunk] = a0;
        if (dp.yy) { a1 = wave_sum(a1); if (l == WSUM_LANE) dp.part[dp.stride + chunk] = a1; }
        return;
    }
#pragma unroll 4
    for (int i = l; i < cnt; i += 64) y[row0 + i] = my[i];
}

// ---------------------------------------------------------------------------------------------- building the plan
// per workgroup (NW x 1024 rows): smallest and largest tile its entries touch
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
        const int a = hi >= 0 ? lo >> TL_TCL2 : 0, b = hi >= 0 ? hi >> TL_TCL2 : -1;
        tmin[g] = a; nspan[g] = b - a + 1;
        atomicMax(&flags[1], b - a + 1);
    }
}

__global__ void k_tl_plus1(int n, const int *in, int *out) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = in[i] + 1; }

// per workgroup: entries per (wavefront, tile of the span); the tiles that hold anything are numbered j = 0 .. ntile - 1
// (tl[], and cmap[] for k_tl_place); per wavefront the counts are scanned over those tiles into group starts.  gstart carries
// ntile + 1 items per workgroup (the last one = the bins' lengths); the lists sit at the offsets the SPANS were scanned to
// (an upper bound).  nstep[chunk] = the bin's length in steps.
__global__ __launch_bounds__(64 * TL_MAXNW) void k_tl_count(int n, int TL_NW, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                           const int *__restrict__ tmin, const int *__restrict__ nspan,
                                                           const int *__restrict__ sofs, int *ntile, int *tl, int *cmap, int *gstart, int *nstep,
                                                           unsigned long long *pairs)
{
    extern __shared__ int hist[];       // [nw][ns] + cm[ns]
    __shared__ int nused_s;
    const int g = blockIdx.x, tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int t0 = tmin[g], ns = nspan[g];
    int *cm = hist + TL_NW * ns;
    for (int i = tid; i < TL_NW * ns; i += 64 * TL_NW) hist[i] = 0;
    __syncthreads();
    const long r0 = (long)(g * TL_NW + w) * TL_RW;
    const int ra = (int)min((long)n, r0), rb = (int)min((long)n, r0 + TL_RW);
    const int k0 = rowptr[ra], k1 = rowptr[rb];
    for (int k = k0 + l; k < k1; k += 64) atomicAdd(&hist[w * ns + (col[k] >> TL_TCL2) - t0], 1);
    __syncthreads();
    const int so = sofs[g];
    if (w == 0) {       // number the tiles that hold anything
        int base = 0;
        const unsigned long long below = l == 0 ? 0ull : (~0ull >> (64 - l));
        for (int b = 0; b < ns; b += 64) {
            const int lt = b + l;
            int h = 0;
            if (lt < ns) for (int q = 0; q < TL_NW; q++) h += hist[q * ns + lt];
            const unsigned long long m = __ballot(h > 0);
            const int j = base + __popcll(m & below);
            if (lt < ns) { cm[lt] = h > 0 ? j : -1; cmap[so + lt] = h > 0 ? j : -1; if (h > 0) tl[so + j] = t0 + lt; }
            base += __popcll(m);
        }
        if (l == 0) { nused_s = base; ntile[g] = base; atomicAdd(pairs, (unsigned long long)base); }
    }
    __syncthreads();
    const int nu = nused_s;
    // each wavefront scans its own row of the histogram (serial over chunks of 64 tiles)
    int *out = gstart + (long)so * TL_NW;
    int run = 0;
    for (int b = 0; b < ns; b += 64) {
        const int lt = b + l;
        const int c = lt < ns ? hist[w * ns + lt] : 0;
        int inc = c;
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off, 64); if (l >= off) inc += v; }
        if (lt < ns && cm[lt] >= 0) out[cm[lt] * TL_NW + w] = run + inc - c;
        run += __shfl(inc, 63, 64);
    }
    if (l == 0) { out[nu * TL_NW + w] = run; nstep[g * TL_NW + w] = (run + TL_STEP - 1) / TL_STEP; }
}

// one wavefront per chunk walks its entries in CSR order, 64 at a time; rank inside the group = entries of that tile
// placed so far + lower lanes of the batch with the same tile (one ballot per distinct tile of the batch): no atomics, no sort.
// (The three packed pairs of a word come from different lanes, or different batches: they are OR-ed into the zeroed stream.)
__global__ __launch_bounds__(64) void k_tl_place(int n, int TL_NW, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                 const double *__restrict__ val, const int *__restrict__ tmin,
                                                 const int *__restrict__ nspan, const int *__restrict__ sofs, const int *__restrict__ cmap,
                                                 const int *__restrict__ gstart, const int *__restrict__ bstep, u64t *stream)
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
    const int so = sofs[g];
    const int *gs = gstart + (long)so * TL_NW;
    u64t *sb = stream + (long)bstep[chunk] * TL_BLK;
    const unsigned long long below = l == 0 ? 0ull : (~0ull >> (64 - l));
    for (int kb = k0; kb < k1; kb += 64) {
        const int k = kb + l;
        const bool active = k < k1;
        const int c = active ? col[k] : 0;
        const int t = active ? (c >> TL_TCL2) - t0 : -1;
        int rank = 0;
        unsigned long long todo = __ballot(active);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int tl_ = __shfl(t, leader, 64);
            const unsigned long long same = __ballot(t == tl_);
            const int base = cur[tl_];
            if (t == tl_) rank = base + __popcll(same & below);
            if (l == leader) cur[tl_] = base + __popcll(same);
            todo &= ~same;
        }
        if (active) {
            int lo = r0, hi = r1 - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (rowptr[mid] <= k) lo = mid; else hi = mid - 1; }
            const int p = gs[cmap[so + t] * TL_NW + w] + rank;          // position in the bin
            const int st = p / TL_STEP, r = p - st * TL_STEP, ln = r / 3, pl = r - 3 * ln;
            u64t *blk = sb + (long)st * TL_BLK;
            // a step's block: words [2 l + k] = plane k < 2 of lane l, [128 + 2 l] = plane 2, [128 + 2 l + 1] = the packed pairs
            blk[pl < 2 ? 2 * ln + pl : 128 + 2 * ln] = (u64t)__double_as_longlong(val[k]);
            const u64t e = ((u64t)(unsigned)(lo - r0) << TL_TCL2) | (u64t)((unsigned)c & (TL_TC - 1));
            atomicOr(&blk[128 + 2 * ln + 1], e << (21 * pl));
        }
    }
}

static void plan_free(TiledPlan *T)
{
    if (!T) return;
    for (void *p : {(void *)T->stream, (void *)T->ntile, (void *)T->sofs, (void *)T->tl, (void *)T->gstart, (void *)T->bstep})
        if (p) (void)hipFree(p);
    delete T;
}

void tiled_free(CsrPart &P)
{
    if (P.tl_plan) { if (ctx().inited) (void)hipDeviceSynchronize(); plan_free(static_cast<TiledPlan *>(P.tl_plan)); }
    P.tl_plan = nullptr; P.tl_state = 0;
}

// Consumer wavefronts per workgroup.  LCG_HIP_TILED_NW (A/B runs): 8 (default: 2 x 16 KB of x + 64 KB of sums, one workgroup of
// nine wavefronts per CU) or 4 (two workgroups of five per CU, twice the tile copies: 841 vs 756 us on the 10M-row row-random band).
static int tiled_nw()
{
    static const int v = [] { const char *e = lab_env("LCG_HIP_TILED_NW"); return e && atoi(e) == 4 ? 4 : 8; }();
    return v;
}

// min_fill: least mean number of entries per (workgroup, tile) pair for the plan to be worth building (0 = build anyway)
static int plan_build(const CsrPart &P, hipStream_t s, double min_fill, TiledPlan **out, const char **why)
{
    *out = nullptr;
    const int n = P.n_rows;
    const long n_cols = P.n_cols;
    *why = "empty matrix or unknown column count";
    if (n <= 0 || n_cols <= 0 || P.nnz <= 0) return 0;
    const int TL_NW = tiled_nw();
    const int nwg = (n + TL_RW * TL_NW - 1) / (TL_RW * TL_NW);
    const int maxspan = 65536 / 4 / (TL_NW + 1) - 8;        // builder histogram [nw + 1][span] within the default dynamic LDS
    TiledPlan *T = new TiledPlan();
    T->n_rows = n; T->nwg = nwg; T->n_cols = n_cols; T->nw = TL_NW;
    int *flags = nullptr, *nstep = nullptr, *tmin = nullptr, *nspan = nullptr, *span1 = nullptr, *cmap = nullptr;
    unsigned long long *pairs = nullptr;
    auto cleanup = [&](int rc) {
        for (void *p : {(void *)flags, (void *)nstep, (void *)pairs, (void *)tmin, (void *)nspan, (void *)span1, (void *)cmap}) if (p) (void)hipFree(p);
        if (rc || !*out) { plan_free(T); *out = nullptr; }
        return rc;
    };
    *why = "a HIP call failed while the plan was built (lcg_hip_last_error)";
#define TCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return cleanup(fail(e_, #call, __FILE__, __LINE__)); } while (0)
    TCHK(hipMalloc(&tmin, sizeof(int) * (size_t)nwg));
    TCHK(hipMalloc(&nspan, sizeof(int) * (size_t)nwg));
    TCHK(hipMalloc(&span1, sizeof(int) * (size_t)nwg));
    TCHK(hipMalloc(&T->ntile, sizeof(int) * (size_t)nwg));
    TCHK(hipMalloc(&T->sofs, sizeof(int) * ((size_t)nwg + 1)));
    TCHK(hipMalloc(&T->bstep, sizeof(int) * ((size_t)nwg * TL_NW + 1)));
    TCHK(hipMalloc(&nstep, sizeof(int) * (size_t)nwg * TL_NW));
    TCHK(hipMalloc(&flags, 2 * sizeof(int)));
    TCHK(hipMalloc(&pairs, sizeof(unsigned long long)));
    TCHK(hipMemsetAsync(flags, 0, 2 * sizeof(int), s));
    TCHK(hipMemsetAsync(pairs, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_tl_span, dim3(nwg), dim3(256), 0, s, n, TL_RW * TL_NW, n_cols, P.rowptr, P.col, tmin, nspan, flags);
    TCHK(hipGetLastError());
    int hflags[2] = {0, 0};
    TCHK(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    TCHK(hipStreamSynchronize(s));
    if (hflags[0]) { *why = "a column index lies outside [0, n_cols)"; return cleanup(0); }
    if (hflags[1] > maxspan) { *why = "a workgroup's rows span too many column tiles (scattered columns: see the binned product)"; return cleanup(0); }
    // the lists hold span + 1 items per workgroup (an upper bound of ntile + 1)
    hipLaunchKernelGGL(k_tl_plus1, dim3((nwg + 255) / 256), dim3(256), 0, s, nwg, nspan, span1);
    TCHK(hipGetLastError());
    long items = 0;
    int rc = device_exclusive_scan(nwg, span1, T->sofs, s, &items);
    if (rc) return cleanup(rc);
    TCHK(hipMalloc(&T->gstart, sizeof(int) * (size_t)items * TL_NW));
    TCHK(hipMalloc(&T->tl, sizeof(int) * (size_t)items));
    TCHK(hipMalloc(&cmap, sizeof(int) * (size_t)items));
    hipLaunchKernelGGL(k_tl_count, dim3(nwg), dim3(64 * TL_NW), sizeof(int) * (TL_NW + 1) * (size_t)hflags[1], s, n, TL_NW, P.rowptr, P.col,
                       tmin, nspan, T->sofs, T->ntile, T->tl, cmap, T->gstart, nstep, pairs);
    TCHK(hipGetLastError());
    long total = 0;
    rc = device_exclusive_scan(nwg * TL_NW, nstep, T->bstep, s, &total);
    if (rc) return cleanup(rc);
    unsigned long long hp = 0;
    TCHK(hipMemcpyAsync(&hp, pairs, sizeof hp, hipMemcpyDeviceToHost, s));
    TCHK(hipStreamSynchronize(s));
    T->steps = total; T->entries = P.nnz; T->pairs = (long)hp;
    if (total <= 0 || total > (0x7fffffffL - TL_SLACK) || hp == 0) { *why = "stream length out of range"; return cleanup(0); }
    if ((double)P.nnz / (double)hp < min_fill) {
        *why = "too few entries per (workgroup, tile) pair: the tile copies would cost more than the gathers they replace"; return cleanup(0);
    }
    const size_t words = ((size_t)total + TL_SLACK) * TL_BLK;
    TCHK(hipMalloc(&T->stream, sizeof(u64t) * words));
    TCHK(hipMemsetAsync(T->stream, 0, sizeof(u64t) * words, s));
    hipLaunchKernelGGL(k_tl_place, dim3(nwg * TL_NW), dim3(64), sizeof(int) * (size_t)hflags[1], s, n, TL_NW, P.rowptr, P.col, P.val,
                       tmin, nspan, T->sofs, cmap, T->gstart, T->bstep, T->stream);
    TCHK(hipGetLastError());
    TCHK(hipStreamSynchronize(s));
    T->bytes = words * 8 + (size_t)items * (TL_NW + 1) * 4 + (size_t)nwg * (8 + 4 * TL_NW);
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
    if (debug_on())
        std::fprintf(stderr, "[lcg_hip] tiled plan for %d x %ld, %ld entries: %s (rc %d)\n", P.n_rows, (long)P.n_cols, (long)P.nnz, P.tl_why, rc);
    if (rc) { (void)hipGetLastError(); return rc; }
    if (!T) return 0;
    P.tl_plan = T; P.tl_state = 1;
    return 1;
}

// dot != nullptr: the product leaves per-chunk sums of y.u (y.y) in dot->part (chunks: tiled_chunks); only the default shape carries
// them (eight consumers, ring of three, non-temporal stream) -- tiled_dot_ok says so beforehand
int tiled_chunks(const CsrPart &P)
{
    const TiledPlan *T = static_cast<const TiledPlan *>(P.tl_plan);
    return T ? T->nwg * T->nw : 0;
}
static int tiled_depth() { static const int v = [] { const char *e = lab_env("LCG_HIP_TILED_DEPTH"); return e ? atoi(e) : 3; }(); return v; }
static int tiled_nt() { static const int v = [] { const char *e = lab_env("LCG_HIP_TILED_NT"); return e ? atoi(e) : 1; }(); return v; }
bool tiled_dot_ok(const CsrPart &P)
{
    const TiledPlan *T = static_cast<const TiledPlan *>(P.tl_plan);
    const int d = tiled_depth();
    return T && T->nw == 8 && tiled_nt() && d != 2 && d != 4 && d != 6;
}

int tiled_launch(const CsrPart &P, const double *x, double *y, hipStream_t s, const int *done, const PushPlan *push, const DotPlan *dot)
{
    const TiledPlan *T = static_cast<const TiledPlan *>(P.tl_plan);
    if (!T) return fail(hipErrorInvalidValue, "tiled A.x without a plan", __FILE__, __LINE__);
    const unsigned grid = 8u * (unsigned)((T->nwg + 7) / 8) + (push ? (unsigned)(push->nblocks + push->nrecv) : 0u);
    const PushPlan pp = push ? *push : PushPlan();
    // ring depth and cache policy of the stream (A/B runs).  Measured on the 10M-row row-random band, same box: depth 2 / 3 / 4 / 6 / 8
    // = 690 / 649 / 657 / 682 / 699 us (eight consumers keep 8 x D x 2 KB in flight through a 32 KB L1: deeper rings evict their
    // own lines), and non-temporal loads on top 670 -> 598 us at depth 3 (the stream no longer pushes the tiles of x out of the
    // L2 -- the kernel then reads at the rate of a pure sum over 4 GiB on the same box, 6.0 TB/s)
    const int depth = tiled_depth(), nt = tiled_nt();
    if (dot) {
        if (!tiled_dot_ok(P)) return fail(hipErrorInvalidValue, "tiled A.x: this shape does not carry the dot", __FILE__, __LINE__);
        const DotPlan dpv = *dot;
        if (push) hipLaunchKernelGGL((k_tile_spmv2<8, 3, true, true, true>), dim3(grid), dim3(9 * 64), 0, s, T->n_rows, T->nwg, T->ntile, T->sofs, T->tl,
                                     T->gstart, T->bstep, T->stream, x, T->n_cols, y, done, pp, dpv);
        else hipLaunchKernelGGL((k_tile_spmv2<8, 3, false, true, true>), dim3(grid), dim3(9 * 64), 0, s, T->n_rows, T->nwg, T->ntile, T->sofs, T->tl,
                                T->gstart, T->bstep, T->stream, x, T->n_cols, y, done, pp, dpv);
        HIPCHK(hipGetLastError());
        return 0;
    }
#define TL_ARGS T->n_rows, T->nwg, T->ntile, T->sofs, T->tl, T->gstart, T->bstep, T->stream, x, T->n_cols, y, done, pp
#define TL2(NW, DD)                                                                                                          \
    do {                                                                                                                     \
        if (push && nt) hipLaunchKernelGGL((k_tile_spmv2<NW, DD, true, true>), dim3(grid), dim3((NW + 1) * 64), 0, s, TL_ARGS);   \
        else if (push) hipLaunchKernelGGL((k_tile_spmv2<NW, DD, true, false>), dim3(grid), dim3((NW + 1) * 64), 0, s, TL_ARGS);   \
        else if (nt) hipLaunchKernelGGL((k_tile_spmv2<NW, DD, false, true>), dim3(grid), dim3((NW + 1) * 64), 0, s, TL_ARGS);     \
        else hipLaunchKernelGGL((k_tile_spmv2<NW, DD, false, false>), dim3(grid), dim3((NW + 1) * 64), 0, s, TL_ARGS);            \
    } while (0)
    if (T->nw == 8) {
        switch (depth) { case 2: TL2(8, 2); break; case 4: TL2(8, 4); break; case 6: TL2(8, 6); break; default: TL2(8, 3); }
    } else if (T->nw == 4) {
        TL2(4, 3);
    } else
        return fail(hipErrorInvalidValue, "tiled A.x: no kernel for this plan's shape", __FILE__, __LINE__);
#undef TL2
#undef TL_ARGS
    HIPCHK(hipGetLastError());
    return 0;
}

// bytes per product by construction: HBM stream + lists + y + x once, and the tile copies (L2 traffic) separately
long tiled_traffic_bytes(const CsrPart &P)
{
    const TiledPlan *T = static_cast<const TiledPlan *>(P.tl_plan);
    return T ? 8L * TL_BLK * T->steps + 4L * (T->nw + 1) * T->pairs + 8L * T->n_rows + 8L * T->n_cols : 0;
}
long tiled_tile_copy_bytes(const CsrPart &P)
{
    const TiledPlan *T = static_cast<const TiledPlan *>(P.tl_plan);
    return T ? T->pairs * (8L << TL_TCL2) : 0;
}
size_t tiled_plan_bytes(const CsrPart &P)
{
    const TiledPlan *T = static_cast<const TiledPlan *>(P.tl_plan);
    return T ? T->bytes : 0;
}

} // namespace lcgh
