// csr_binned.hip -- A.x for matrices whose columns are scattered (no reuse of x inside a block of rows).
//
// The row-block kernels of csr.hip gather x[col] from global memory.  When a block's columns are spread
// over tens of MB every 8-byte gather drags a whole cache line through the fabric: the 10M-row "scrambled"
// system moves >= 5x its algorithmic bytes and A.x runs at 0.69 TB/s (profiles/r01_configs.jsonl).  No cache
// on the chip holds x at word granularity except the LDS, and no CU's LDS holds x AND the rows that use it.
// So the product is made in two streaming passes over a re-ordered copy of the matrix ("binned" format), with
// the LDS as the only place where anything is accessed at random:
//
//   pass 1  k_bin_expand   one workgroup per column tile of BN_C = 8192 columns (a slice of x: 64 KB of LDS).
//           It walks the tile's entries (16-bit tile-relative columns, 2 B per entry), reads x from LDS and
//           writes xg[p] = x[col(p)] to the position p the entry has in pass 2's order -- in whole, aligned
//           64-byte granules (8 entries), so HBM sees full-sector writes only.
//   pass 2  k_bin_reduce   one WAVEFRONT per chunk of BN_RW = 2048 rows, whose sums live in 16 KB of LDS.
//           It streams the chunk's bin -- val (8 B), xg (8 B), 16-bit chunk-relative row (2 B), all coalesced
//           16 B per lane -- and adds val*xg into the row's sum with the LDS's own fp64 add (ds_add_f64);
//           then writes y coalesced.  A row is summed by exactly one wavefront, in stream order, so the
//           result does not depend on timing (lanes of one LDS instruction that meet in a row are serialised
//           by the LDS in a fixed order): bit-identical from call to call and from build to build of the
//           plan (the entries of a group keep their CSR order: k_bin_place ranks them without atomics).
//
// Order of the streams.  Entry (row i, col j) belongs to group (chunk w = i / BN_RW, tile t = j / BN_C).
// Pass 2 reads [w][t][entries by CSR position]; pass 1 reads [t][w][same order]; a group is padded to a
// multiple of 8 entries (padding: row 0xFFFF, value 0), a chunk's bin to a multiple of 512.  Per entry the
// two passes move 2 + 0.5 + 8 (pass 1) and 8 + 8 + 2 (pass 2) = 28.5 bytes instead of the 12 of CSR, all
// of them streamed: 9.9 GB instead of >= 21 GB of line traffic on the scrambled system.
// Deviation from csr.hip's kernels: products are rounded before they are added (no FMA chain) and a row is
// summed in column order by one lane sequence, not by four lanes and a tree: y differs from the plain
// kernels' y in the last bits (tests: 1e-13 relative to |A||x|).
#include <algorithm>
#include <cstring>
#include <vector>

#include "devcommon.hpp"

namespace lcgh {


constexpr int BN_C = 8192;          // columns per tile (x slice in LDS: 64 KB)
constexpr int BN_C_LOG2 = 13;
constexpr int BN_RW = 2048;         // rows per wave chunk (sums in LDS: 16 KB per wavefront)
constexpr int BN_G = 8;             // entries per granule (64 B of xg)
constexpr int BN_STEP = 512;        // entries one wavefront takes per step of pass 2
constexpr int BN_MAXT = 12288;      // most column tiles (LDS histogram of the plan builder: 48 KB)
constexpr unsigned short BN_PAD = 0xFFFF;

typedef int v4i_ __attribute__((ext_vector_type(4)));
typedef double v2d_ __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

struct BinnedPlan {
    int n_rows = 0, nw = 0, nt = 0;
    long n_cols = 0;
    long gran2 = 0, gran1 = 0;      // granules of the two streams
    double *val2 = nullptr, *xg = nullptr;
    u16 *lrowP = nullptr, *lcol1 = nullptr;
    int *binofs = nullptr;          // [nw + 1] first granule of each chunk's bin
    int *dstg = nullptr;            // [gran1] granule of stream 2 each granule of stream 1 fills
    int *part = nullptr;            // [3 * nparts]: tile, first granule, end granule of each pass-1 work item
    int nparts = 0;
    size_t bytes = 0;               // device memory held by the plan
};

// ------------------------------------------------------------------------------------ the two passes
constexpr int BN_XB = 512;          // threads of a pass-1 workgroup (64 KB of LDS each: two per CU, 16 wavefronts)
__global__ __launch_bounds__(BN_XB) void k_bin_expand(long n_cols, const int *__restrict__ part, const u16 *__restrict__ lcol1,
                                                      const int *__restrict__ dstg, const double *__restrict__ x,
                                                      double *__restrict__ xg, const int *done)
{
    constexpr int VB = BN_XB;
    __shared__ __attribute__((aligned(16))) double sx[BN_C];
    if (done && *done) return;
    const int tid = threadIdx.x;
    const int t = part[3 * blockIdx.x], g0 = part[3 * blockIdx.x + 1], g1 = part[3 * blockIdx.x + 2];
    const long c0 = (long)t * BN_C;
    const int cn = (int)min((long)BN_C, n_cols - c0);
    // the slice of x, every load of a lane in flight before the first LDS store: 16 B per lane where x allows it
    if ((((uintptr_t)x) & 15) == 0) {
        constexpr int NL = BN_C / 2 / VB;
        v2d_ v[NL];
#pragma unroll
        for (int q = 0; q < NL; q++) {
            const int i = 2 * (q * VB + tid);
            v[q] = i + 1 < cn ? *reinterpret_cast<const v2d_ *>(x + c0 + i) : v2d_{i < cn ? x[c0 + i] : 0.0, 0.0};
        }
#pragma unroll
        for (int q = 0; q < NL; q++) reinterpret_cast<v2d_ *>(sx)[q * VB + tid] = v[q];
    } else {
        constexpr int NL = BN_C / VB;
        double v[NL];
#pragma unroll
        for (int q = 0; q < NL; q++) { const int i = q * VB + tid; v[q] = x[c0 + (i < cn ? i : 0)]; }
#pragma unroll
        for (int q = 0; q < NL; q++) sx[q * VB + tid] = v[q];
    }
    __syncthreads();
    // A lane takes one PAIR of entries (a quarter of a granule): a 4-byte load of two tile-relative columns, two
    // LDS reads, one 16-byte store.  Consecutive lanes take consecutive pairs, so a wavefront's store is 16
    // consecutive granules = 1 KB contiguous wherever they lie in one group (a lane per granule wrote 64 lanes x
    // 16 B at a 64-byte stride: four times the requests at a quarter of the size, 1.39 ms instead of 0.7 for the
    // 3.7 GB of this pass on the 10M-row scrambled system).  Eight pairs per lane in flight.
    const unsigned *lc2 = reinterpret_cast<const unsigned *>(lcol1);
    v2d_ *out2 = reinterpret_cast<v2d_ *>(xg);
    constexpr int UN = 8;
    const long e0 = 4L * g0, e1 = 4L * g1;
    for (long e = e0 + tid; e < e1; e += (long)UN * VB) {
        unsigned lc[UN]; int d[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long eu = e + (long)u * VB;
            const long ec = eu < e1 ? eu : e;           // branch-free: lanes past the end repeat their first pair
            lc[u] = lc2[ec];
            d[u] = dstg[ec >> 2];
        }
        v2d_ v[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) { v[u].x = sx[lc[u] & 0xffffu]; v[u].y = sx[lc[u] >> 16]; }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long eu = e + (long)u * VB;
            // non-temporal: xg is written once here and read once by pass 2 -- 2094 -> 1960 us for the two passes on the 10M-row
            // scrambled system (the write stream no longer competes with the streams pass 1 reads for the L2)
            if (eu < e1) __builtin_nontemporal_store(v[u], &out2[4L * d[u] + (eu & 3)]);
        }
    }
}

// lane l of a step takes entries p + i*128 + 2*l + j (i < 4, j < 2): every load is 16 B per lane, 1 KB per
// wavefront, contiguous; lrowP holds the step's rows in that lane order (8 per lane = one 16-byte load)
__global__ __launch_bounds__(VB) void k_bin_reduce(int n, int nw, const int *__restrict__ binofs, const double *__restrict__ val2,
                                                   const u16 *__restrict__ lrowP, const double *__restrict__ xg,
                                                   double *__restrict__ y, const int *done)
{
    __shared__ __attribute__((aligned(16))) double ys[VB / 64][BN_RW];
    if (done && *done) return;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int wc = blockIdx.x * (VB / 64) + w;
    if (wc >= nw) return;               // no workgroup barrier below: wavefronts are independent
    double *my = ys[w];
    {
        v2d_ z; z.x = 0.0; z.y = 0.0;
#pragma unroll
        for (int i = 0; i < BN_RW / 128; i++) reinterpret_cast<v2d_ *>(my)[i * 64 + l] = z;
    }
    const long p0 = (long)BN_G * binofs[wc], p1 = (long)BN_G * binofs[wc + 1];
    for (long p = p0; p < p1; p += 2 * BN_STEP) {
        // two steps in flight: 18 loads of 16 B per lane before the first add
        const bool two = p + BN_STEP < p1;
        const long pb = two ? p + BN_STEP : p;
        v2d_ xa[4], va[4], xb[4], vb[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // non-temporal: both streams pass through once (1958 -> 1864 us for the two passes on the 10M-row scrambled system)
            xa[i] = __builtin_nontemporal_load(reinterpret_cast<const v2d_ *>(xg + p + i * 128 + 2 * l));
            va[i] = __builtin_nontemporal_load(reinterpret_cast<const v2d_ *>(val2 + p + i * 128 + 2 * l));
            xb[i] = __builtin_nontemporal_load(reinterpret_cast<const v2d_ *>(xg + pb + i * 128 + 2 * l));
            vb[i] = __builtin_nontemporal_load(reinterpret_cast<const v2d_ *>(val2 + pb + i * 128 + 2 * l));
        }
        const v4i_ ra = *reinterpret_cast<const v4i_ *>(lrowP + p + 8 * l);
        const v4i_ rb = *reinterpret_cast<const v4i_ *>(lrowP + pb + 8 * l);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned r0 = (unsigned)ra[i] & 0xffffu, r1 = ((unsigned)ra[i] >> 16) & 0xffffu;
            if (r0 != BN_PAD) __hip_atomic_fetch_add(my + r0, va[i].x * xa[i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (r1 != BN_PAD) __hip_atomic_fetch_add(my + r1, va[i].y * xa[i].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (two) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned r0 = (unsigned)rb[i] & 0xffffu, r1 = ((unsigned)rb[i] >> 16) & 0xffffu;
                if (r0 != BN_PAD) __hip_atomic_fetch_add(my + r0, vb[i].x * xb[i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (r1 != BN_PAD) __hip_atomic_fetch_add(my + r1, vb[i].y * xb[i].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    const long row0 = (long)wc * BN_RW;
    const int cnt = (int)min((long)BN_RW, (long)n - row0);
#pragma unroll 4
    for (int i = l; i < cnt; i += 64) y[row0 + i] = my[i];
}

// ------------------------------------------------------------------------------------ building the plan
// per chunk: entries per tile (LDS histogram), their granule counts scanned into the chunk's bin
__global__ __launch_bounds__(VB) void k_bin_count(int n, int nt, long n_cols, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                  int *cnt, int *gofs2, int *bing, int *flags)
{
    extern __shared__ int hist[];       // [nt] + [VB]
    int *part = hist + nt;
    const int wc = blockIdx.x, tid = threadIdx.x;
    const long r0 = (long)wc * BN_RW;
    const int r1 = (int)min((long)n, r0 + BN_RW);
    for (int t = tid; t < nt; t += VB) hist[t] = 0;
    __syncthreads();
    const int k0 = rowptr[r0], k1 = rowptr[r1];
    bool bad = false;
    for (int k = k0 + tid; k < k1; k += VB) {
        const int c = col[k];
        if (c < 0 || c >= n_cols) bad = true; else atomicAdd(&hist[c >> BN_C_LOG2], 1);
    }
    if (bad) flags[0] = 1;
    __syncthreads();
    // exclusive scan of the granule counts: each thread owns a contiguous run of tiles
    const int per = (nt + VB - 1) / VB;
    const int t0 = min(nt, tid * per), t1 = min(nt, t0 + per);
    int sum = 0, big = 0;
    for (int t = t0; t < t1; t++) { const int h = hist[t]; cnt[(long)wc * nt + t] = h; sum += (h + BN_G - 1) / BN_G; big = max(big, h); }
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < VB; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int t = t0; t < t1; t++) { gofs2[(long)wc * nt + t] = run; run += (hist[t] + BN_G - 1) / BN_G; }
    if (tid == VB - 1) bing[wc] = (part[VB - 1] + BN_STEP / BN_G - 1) / (BN_STEP / BN_G) * (BN_STEP / BN_G);
    if (big > 0) atomicMax(&flags[1], big);
}

// per tile: granule offsets of its groups in stream 1 (order [tile][chunk]) and the tile's total
__global__ __launch_bounds__(VB) void k_bin_tile_scan(int nw, int nt, const int *__restrict__ cnt, int *gofs1, int *tileg)
{
    __shared__ int part[VB];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int per = (nw + VB - 1) / VB;
    const int w0 = min(nw, tid * per), w1 = min(nw, w0 + per);
    int sum = 0;
    for (int w = w0; w < w1; w++) sum += (cnt[(long)w * nt + t] + BN_G - 1) / BN_G;
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < VB; off <<= 1) {
        const int v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int w = w0; w < w1; w++) { gofs1[(long)t * nw + w] = run; run += (cnt[(long)w * nt + t] + BN_G - 1) / BN_G; }
    if (tid == VB - 1) tileg[t] = part[VB - 1];
}

// One wavefront per chunk walks the chunk's entries in CSR order, 64 at a time, and gives every entry its
// place in both streams.  The rank of an entry inside its (chunk, tile) group is the number of entries of
// that group in front of it in CSR order: a running count per tile in LDS plus, inside the batch of 64, the
// number of lower lanes with the same tile (one ballot per distinct tile of the batch).  No atomics, no sort:
// the layout is a pure function of the matrix, whatever the size of a group.
__global__ __launch_bounds__(64) void k_bin_place(int n, int nw, int nt, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                  const double *__restrict__ val, const int *__restrict__ gofs2,
                                                  const int *__restrict__ binofs, const int *__restrict__ gofs1,
                                                  const int *__restrict__ tileofs, double *val2, u16 *lrowP, u16 *lcol1, int *dstg)
{
    extern __shared__ int cur[];        // [nt] entries of each tile placed so far
    const int wc = blockIdx.x, l = threadIdx.x;
    const int r0 = wc * BN_RW;
    const int r1 = min(n, r0 + BN_RW);
    for (int t = l; t < nt; t += 64) cur[t] = 0;
    const int k0 = rowptr[r0], k1 = rowptr[r1];
    const long bin0 = (long)BN_G * binofs[wc];
    const unsigned long long below = l == 0 ? 0ull : (~0ull >> (64 - l));
    for (int kb = k0; kb < k1; kb += 64) {
        const int k = kb + l;
        const bool active = k < k1;
        const int c = active ? col[k] : 0;
        const int t = active ? c >> BN_C_LOG2 : -1;
        int rank = 0;
        unsigned long long todo = __ballot(active);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int tl = __shfl(t, leader, 64);
            const unsigned long long same = __ballot(t == tl);
            const int base = cur[tl];                                   // every lane reads before the leader writes
            if (t == tl) rank = base + __popcll(same & below);
            if (l == leader) cur[tl] = base + __popcll(same);
            todo &= ~same;
        }
        if (active) {
            // row of CSR position k inside the chunk: last row whose rowptr <= k
            int lo = r0, hi = r1 - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (rowptr[mid] <= k) lo = mid; else hi = mid - 1; }
            const long g = (long)wc * nt + t;
            const long p2 = bin0 + (long)BN_G * gofs2[g] + rank;
            const long p1 = (long)BN_G * (tileofs[t] + gofs1[(long)t * nw + wc]) + rank;
            val2[p2] = val[k];
            const long q = p2 - bin0;
            const long s = q >> 9; const int rem = (int)(q & 511);
            const int i = rem >> 7, ln = (rem & 127) >> 1, j = rem & 1;
            lrowP[bin0 + (s << 9) + ln * 8 + i * 2 + j] = (u16)(lo - r0);
            lcol1[p1] = (u16)(c - (t << BN_C_LOG2));
            if ((rank & (BN_G - 1)) == 0) dstg[p1 / BN_G] = (int)(p2 / BN_G);
        }
    }
}

static void plan_free(BinnedPlan *B)
{
    if (!B) return;
    for (void *p : {(void *)B->val2, (void *)B->xg, (void *)B->lrowP, (void *)B->lcol1, (void *)B->binofs, (void *)B->dstg, (void *)B->part})
        if (p) (void)hipFree(p);
    delete B;
}

void binned_free(CsrPart &P)
{
    if (P.bn_plan) { if (ctx().inited) (void)hipDeviceSynchronize(); plan_free(static_cast<BinnedPlan *>(P.bn_plan)); }
    P.bn_plan = nullptr; P.bn_state = 0;
}

// Build the plan of P (real values, columns < P.n_cols).  Returns 0 and sets *out, or 0 with *out == nullptr when
// the matrix does not qualify (too many tiles, an index out of range), or a failure code.
static int plan_build(const CsrPart &P, hipStream_t s, BinnedPlan **out, const char **why)
{
    *out = nullptr;
    const int n = P.n_rows;
    const long n_cols = P.n_cols;
    *why = "empty matrix or unknown column count";
    if (n <= 0 || n_cols <= 0 || P.nnz <= 0) return 0;
    const int nw = (n + BN_RW - 1) / BN_RW;
    const long ntl = (n_cols + BN_C - 1) / BN_C;
    *why = "too many column tiles";
    if (ntl > BN_MAXT || (long)nw * ntl > 0x7fffffffL / 2) return 0;
    *why = "a HIP call failed while the plan was built (lcg_hip_last_error)";
    const int nt = (int)ntl;
    const long ng = (long)nw * nt;
    BinnedPlan *B = new BinnedPlan();
    B->n_rows = n; B->nw = nw; B->nt = nt; B->n_cols = n_cols;
    int *cnt = nullptr, *gofs2 = nullptr, *gofs1 = nullptr, *bing = nullptr, *tileg = nullptr, *tileofs = nullptr, *flags = nullptr;
    auto cleanup = [&](int rc) {
        for (void *p : {(void *)cnt, (void *)gofs2, (void *)gofs1, (void *)bing, (void *)tileg, (void *)tileofs, (void *)flags})
            if (p) (void)hipFree(p);
        if (rc || !*out) { plan_free(B); *out = nullptr; }
        return rc;
    };
#define BCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return cleanup(fail(e_, #call, __FILE__, __LINE__)); } while (0)
    BCHK(hipMalloc(&cnt, sizeof(int) * (size_t)ng));
    BCHK(hipMalloc(&gofs2, sizeof(int) * (size_t)ng));
    BCHK(hipMalloc(&gofs1, sizeof(int) * (size_t)ng));
    BCHK(hipMalloc(&bing, sizeof(int) * (size_t)nw));
    BCHK(hipMalloc(&tileg, sizeof(int) * (size_t)nt));
    BCHK(hipMalloc(&tileofs, sizeof(int) * ((size_t)nt + 1)));
    BCHK(hipMalloc(&B->binofs, sizeof(int) * ((size_t)nw + 1)));
    BCHK(hipMalloc(&flags, 2 * sizeof(int)));
    BCHK(hipMemsetAsync(flags, 0, 2 * sizeof(int), s));
    const size_t lds1 = sizeof(int) * ((size_t)nt + VB);
    hipLaunchKernelGGL(k_bin_count, dim3(nw), dim3(VB), lds1, s, n, nt, n_cols, P.rowptr, P.col, cnt, gofs2, bing, flags);
    BCHK(hipGetLastError());
    int hflags[2] = {0, 0};
    BCHK(hipMemcpyAsync(hflags, flags, sizeof hflags, hipMemcpyDeviceToHost, s));
    BCHK(hipStreamSynchronize(s));
    if (hflags[0]) { *why = "a column index lies outside [0, n_cols)"; return cleanup(0); }
    long total2 = 0, total1 = 0;
    int rc = device_exclusive_scan(nw, bing, B->binofs, s, &total2);
    if (rc) return cleanup(rc);
    hipLaunchKernelGGL(k_bin_tile_scan, dim3(nt), dim3(VB), 0, s, nw, nt, cnt, gofs1, tileg);
    BCHK(hipGetLastError());
    rc = device_exclusive_scan(nt, tileg, tileofs, s, &total1);
    if (rc) return cleanup(rc);
    // (granule offsets are int32, entry positions 64-bit everywhere: up to 2^31 granules = 1.7e10 padded entries.  At 6e7 rows of 33
    //  scattered entries a (chunk, tile) group holds 9 entries on average and the padding to whole granules adds ~40 %: 2.8e9 entries,
    //  3.5e8 granules -- round 2's limit of 2^31 / 8 granules refused that system)
    if (total2 <= 0 || total1 <= 0 || total2 > 0x7fffffffL - BN_STEP || total1 > total2) { *why = "stream length out of range"; return cleanup(0); }
    B->gran2 = total2; B->gran1 = total1;
    const size_t e2 = (size_t)BN_G * total2 + BN_STEP;      // one step of slack: the last wavefront may load past its bin
    const size_t e1 = (size_t)BN_G * total1 + BN_G;
    BCHK(hipMalloc(&B->val2, sizeof(double) * e2));
    BCHK(hipMalloc(&B->xg, sizeof(double) * e2));
    BCHK(hipMalloc(&B->lrowP, sizeof(u16) * e2));
    BCHK(hipMalloc(&B->lcol1, sizeof(u16) * e1));
    BCHK(hipMalloc(&B->dstg, sizeof(int) * ((size_t)total1 + 1)));
    BCHK(hipMemsetAsync(B->val2, 0, sizeof(double) * e2, s));
    BCHK(hipMemsetAsync(B->xg, 0, sizeof(double) * e2, s));
    BCHK(hipMemsetAsync(B->lrowP, 0xff, sizeof(u16) * e2, s));
    BCHK(hipMemsetAsync(B->lcol1, 0, sizeof(u16) * e1, s));
    BCHK(hipMemsetAsync(B->dstg, 0, sizeof(int) * ((size_t)total1 + 1), s));
    hipLaunchKernelGGL(k_bin_place, dim3(nw), dim3(64), sizeof(int) * (size_t)nt, s, n, nw, nt, P.rowptr, P.col, P.val, gofs2, B->binofs,
                       gofs1, tileofs, B->val2, B->lrowP, B->lcol1, B->dstg);
    BCHK(hipGetLastError());
    // pass-1 work items: a tile's granules in pieces, so that the grid fills the chip whatever the tile count
    std::vector<int> htile((size_t)nt + 1);
    BCHK(hipMemcpyAsync(htile.data(), tileofs, sizeof(int) * ((size_t)nt + 1), hipMemcpyDeviceToHost, s));
    BCHK(hipStreamSynchronize(s));
    const long piece = std::min<long>(8192, std::max<long>(1024, total1 / 2048));
    std::vector<int> parts;
    for (int t = 0; t < nt; t++)
        for (long g = htile[t]; g < htile[t + 1]; g += piece) {
            parts.push_back(t); parts.push_back((int)g); parts.push_back((int)std::min<long>(htile[t + 1], g + piece));
        }
    B->nparts = (int)(parts.size() / 3);
    if (B->nparts == 0) { *why = "no work items"; return cleanup(0); }
    BCHK(hipMalloc(&B->part, sizeof(int) * parts.size()));
    BCHK(hipMemcpyAsync(B->part, parts.data(), sizeof(int) * parts.size(), hipMemcpyHostToDevice, s));
    BCHK(hipStreamSynchronize(s));
    B->bytes = e2 * (8 + 8 + 2) + e1 * 2 + ((size_t)total1 + 1) * 4 + ((size_t)nw + 1) * 4 + parts.size() * 4;
#undef BCHK
    *out = B;
    *why = "ready";
    return cleanup(0);
}

// 1 = plan ready, 0 = this matrix does not use the binned product, < 0 = failure
int binned_ready(const CsrPart &P, hipStream_t s)
{
    if (P.bn_state != 0) return P.bn_state > 0 ? 1 : 0;
    if (P.bn_plan) { P.bn_state = 1; return 1; }       // built earlier (e.g. under a forced mode): reuse
    P.bn_state = -1;
    BinnedPlan *B = nullptr;
    int rc = plan_build(P, s, &B, &P.bn_why);
    if (debug_on())
        std::fprintf(stderr, "[lcg_hip] binned plan for %d x %ld, %ld entries: %s (rc %d%s%s)\n", P.n_rows, (long)P.n_cols, (long)P.nnz, P.bn_why, rc,
                     rc ? ": " : "", rc ? ctx().err.c_str() : "");
    if (rc) { (void)hipGetLastError(); return rc; }
    if (!B) return 0;
    P.bn_plan = B; P.bn_state = 1;
    return 1;
}

int binned_launch(const CsrPart &P, const double *x, double *y, hipStream_t s, const int *done)
{
    const BinnedPlan *B = static_cast<const BinnedPlan *>(P.bn_plan);
    if (!B) return fail(hipErrorInvalidValue, "binned A.x without a plan", __FILE__, __LINE__);
    hipLaunchKernelGGL(k_bin_expand, dim3(B->nparts), dim3(BN_XB), 0, s, B->n_cols, B->part, B->lcol1, B->dstg, x, B->xg, done);
    hipLaunchKernelGGL(k_bin_reduce, dim3((B->nw + VB / 64 - 1) / (VB / 64)), dim3(VB), 0, s, B->n_rows, B->nw, B->binofs, B->val2,
                       B->lrowP, B->xg, y, done);
    HIPCHK(hipGetLastError());
    return 0;
}

// bytes the two passes move per product (for reporting): streams + x slices + y
long binned_traffic_bytes(const CsrPart &P)
{
    const BinnedPlan *B = static_cast<const BinnedPlan *>(P.bn_plan);
    if (!B) return 0;
    return (long)BN_G * B->gran1 * (2 + 8) + B->gran1 * 4 + (long)B->nparts * BN_C * 8 + (long)BN_G * B->gran2 * (8 + 8 + 2) + 8L * B->n_rows;
}
size_t binned_plan_bytes(const CsrPart &P)
{
    const BinnedPlan *B = static_cast<const BinnedPlan *>(P.bn_plan);
    return B ? B->bytes : 0;
}

} // namespace lcgh
