// csr.hip -- CSR matrices in HBM: the row-block A.x kernels, the choice of a kernel family per matrix (or per range of rows), the
// handle's C ABI.  (Ingest, Jacobi diagonal, op(A), generators: csr_build.hip.)
//
// A.x is the kernel the whole path is judged on: >= 80 % of the bytes of a CG iteration.
// It is HBM-bound (0.17 flop/byte): no MFMA.  The row-block kernels of this file:
//
//  k_spmv_lds1<R>   (default)  One 256-thread block owns R consecutive rows (R*T = 256).
//      Stage 1: the block's contiguous slice of val/col streams from HBM into LDS with
//      16-byte-per-lane coalesced loads, ALL issued before the first LDS store (the CSR
//      arrays are read exactly once, at full width, whatever the row lengths).  Stage 2: lane (row = tid % R, j = tid / R) walks
//      its row's entries j, j+T, ... out of LDS and gathers x; consecutive lanes hold
//      consecutive ROWS, so for matrices with diagonal / stencil structure the x gather of a
//      wavefront is one contiguous run, and for arbitrary columns it is no worse than any
//      other mapping.  The T partial sums of a row meet in LDS; y is written coalesced.
//  k_spmv_ldsp      the same mapping for large real matrices with packed block-relative columns (18 / 21 bits) and, where a
//      64-row block's columns advance by one per row, RUN blocks: row 0's columns only, x gathers sent out with the value
//      stream (the headline's kernel: DESIGN.md section 3.1a).  <DOT>: carries the dot that follows the product.
//  k_spmv_run1      short rows (stencils) whose blocks are runs: one wavefront per 64-row block, no barrier.
//  k_spmv_lds1d     small systems: the plain body + the dot that follows the product (two launches per CG iteration).
//  k_spmv_wave<T>   T consecutive lanes share a row (T = 64: wavefront per row), partial
//      sums folded with __shfl_down.  Better for long rows (> ~100 entries).
// (csr_tiled.hip, csr_binned.hip: the products for columns that do not run along diagonals.)
//
// Algorithmic bytes: 12*nnz + 4*(N+1) + 8*N (x) + 8*N (y), real.
#include <algorithm>
#include <type_traits>
#include <cstring>
#include <functional>
#include <numeric>

#include <chrono>

#include "csr_plan.hpp"

namespace lcgh {

// ------------------------------------------------------------------- wave-per-row family
// PUSH: the first pp.nblocks blocks of the grid do not multiply; they carry this rank's boundary
// entries of x to the neighbours (devcommon.hpp: push_block) while the rest of the grid works.
template <class V, int T, bool ACC, bool PUSH = false>
__global__ __launch_bounds__(VB) void k_spmv_wave(int n, const int *__restrict__ rowptr,
                                                  const int *__restrict__ col, const V *__restrict__ val,
                                                  const V *__restrict__ x, V *__restrict__ y,
                                                  const int *done, PushPlan pp)
{
    if (PUSH && (int)blockIdx.x < pp.nblocks) { push_block(pp, blockIdx.x); return; }
    if (PUSH && pp.nrecv > 0 && (int)blockIdx.x >= (int)gridDim.x - pp.nrecv) { recv_block(pp, (int)blockIdx.x - ((int)gridDim.x - pp.nrecv)); return; }
    const int bid = PUSH ? blockIdx.x - pp.nblocks : blockIdx.x;
    if (done && *done) return;
    const int sub = threadIdx.x % T;
    const long row = ((long)bid * VB + threadIdx.x) / T;
    V acc = vzero(V());
    if (row < n) {
        const int s = rowptr[row], e = rowptr[row + 1];
        int k = s + sub;
        // two independent gathers in flight per lane
        for (; k + T < e; k += 2 * T) {
            const int c0 = col[k], c1 = col[k + T];
            const V a0 = val[k], a1 = val[k + T];
            acc = mac(a0, x[c0], acc);
            acc = mac(a1, x[c1], acc);
        }
        if (k < e) acc = mac(val[k], x[col[k]], acc);
    }
#pragma unroll
    for (int off = T / 2; off > 0; off >>= 1) acc = vadd(acc, shfl_down_v(acc, off, T));
    if (sub == 0 && row < n) y[row] = ACC ? vadd(y[row], acc) : acc;
}

// ----------------------------------------------------------------------- LDS-staged family
// LDS budget per block: 26,880 B of staging (6 blocks fit a CU's 160 KiB; registers currently
// admit 5).  The row-sum exchange buffer aliases the staging buffer.
// (LdsCfg<V>::CH, the largest window: csr_plan.hpp)
// native vector types: arrays of these stay in registers (arrays of HIP's struct-wrapped int4 /
// double2 were demoted to scratch here: 2x slower)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// largest slice (entries from the 4-aligned start of a block's first row to the end of its last
// row) over all blocks of R rows: decides once per matrix whether every block fits one window
__global__ void k_max_slice(int n, int R, const int *rowptr, int *out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const long row0 = (long)b * R;
    if (row0 >= n) return;
    const int r1 = (int)min((long)n, row0 + R);
    atomicMax(out, rowptr[r1] - (rowptr[row0] & ~3));
}
void launch_max_slice(int n, int R, const int *rowptr, int *out, hipStream_t s)
{
    const int nb = (n + R - 1) / R;
    hipLaunchKernelGGL(k_max_slice, dim3((nb + VB - 1) / VB), dim3(VB), 0, s, n, R, rowptr, out);
}

// ---- one-window kernel: the host has verified (k_max_slice) that every block's slice fits one
// LDS window.  ALL of the block's HBM loads are issued before the first LDS store, 16 B per
// lane per access (~25 KB in flight per block); then every lane keeps UNR gathers of x in flight.
// The work of one row block, shared by the plain kernel and the one that carries a dot (below).  Returns the finished y
// of row row0 + tid % R on the lanes that wrote it (`mine`).
template <class V, int R, bool ACC, int CHE = LdsCfg<V>::CH>
__device__ __forceinline__ V lds1_block(int bid, int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                        const V *__restrict__ val, const V *__restrict__ x, V *__restrict__ y, V *sval, int *scol,
                                        bool &mine)
{
    constexpr int T = VB / R;
    constexpr int CH = CHE;                                 // entries per LDS window (multiple of 4)
    constexpr int NRND = (CH + VB * 4 - 1) / (VB * 4);      // 4-entry units per lane
    constexpr int VU = sizeof(V) / 4;                       // 16-byte pieces of val per 4 entries
    constexpr int UNR = 4;                                  // x gathers in flight per lane
    static_assert(T * R <= CH, "row-sum exchange must fit the staging buffer");
    V(*sred)[R] = reinterpret_cast<V(*)[R]>(sval);

    const int tid = threadIdx.x;
    const int row0 = bid * R;
    const int nrows = min(R, n - row0);
    const int rl = tid % R, j0 = tid / R;
    const int base = rowptr[row0] & ~3;
    const int cnt = rowptr[row0 + nrows] - base;
    // this lane's row bounds, requested BEFORE the block's stream: behind the fence below they would be issued when the stream has
    // arrived and cost the gather phase a memory latency of its own (vmcnt counts in order)
    const int rsafe = rl < nrows ? rl : 0;              // (branch-free, like the stream loads below)
    int rs = rowptr[row0 + rsafe], re = rowptr[row0 + rsafe + 1];
    if (rl >= nrows) { rs = 0; re = 0; }

    v4i pc[NRND]; v2d pv[NRND * VU];
#pragma unroll
    for (int r = 0; r < NRND; r++) {
        const int u = tid * 4 + r * VB * 4;
        // branch-free: lanes past the slice re-read its first unit (an L1 hit) instead of being
        // masked off -- a conditional load makes the compiler drain vmcnt at every join, which
        // serialises the rounds.  col/val carry >= 64 B of slack (CsrPart::padded): no tail case.
        const long g = (long)base + (u < cnt ? u : 0);
        pc[r] = *reinterpret_cast<const v4i *>(col + g);
#pragma unroll
        for (int q = 0; q < VU; q++) pv[r * VU + q] = reinterpret_cast<const v2d *>(val + g)[q];
    }
    // keep every load above every LDS store below: without this fence the scheduler pairs each
    // load with its store to save registers and the block has a third of the bytes in flight
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < NRND; r++) {
        const int u = tid * 4 + r * VB * 4;
        if (u < cnt) {
            *reinterpret_cast<v4i *>(scol + u) = pc[r];
#pragma unroll
            for (int q = 0; q < VU; q++) reinterpret_cast<v2d *>(sval + u)[q] = pv[r * VU + q];
        }
    }
    __syncthreads();
    // lane (row rl, slot j0) takes entries rs+j0, rs+j0+T, ... of its row
    V acc = vzero(V());
    int k = rs + j0;
    for (; k + (UNR - 1) * T < re; k += UNR * T) {
        int c[UNR]; V a[UNR], xv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; q++) { c[q] = scol[k + q * T - base]; a[q] = sval[k + q * T - base]; }
#pragma unroll
        for (int q = 0; q < UNR; q++) xv[q] = x[c[q]];
#pragma unroll
        for (int q = 0; q < UNR; q++) acc = mac(a[q], xv[q], acc);
    }
    for (; k < re; k += T) acc = mac(sval[k - base], x[scol[k - base]], acc);
    __syncthreads();
    // the T partial sums of a row meet in LDS (staging buffer reused); y written coalesced
    mine = j0 == 0 && rl < nrows;
    if (T > 1) {
        sred[j0][rl] = acc;
        __syncthreads();
        if (mine) {
            V v = sred[0][rl];
#pragma unroll
            for (int j = 1; j < T; j++) v = vadd(v, sred[j][rl]);
            acc = ACC ? vadd(y[row0 + rl], v) : v;
            y[row0 + rl] = acc;
        }
    } else if (mine) {
        if (ACC) acc = vadd(y[row0 + rl], acc);
        y[row0 + rl] = acc;
    }
    return acc;
}

// CHE: the LDS window in entries (12 bytes each for real matrices); smaller where every block of the matrix fits -- more workgroups
// per CU (pk_window below: the ladder of k_spmv_ldsp)
template <class V, int R, bool ACC, bool PUSH = false, int CHE = LdsCfg<V>::CH>
__global__ __launch_bounds__(VB) void k_spmv_lds1(int n, long nnz, const int *__restrict__ rowptr,
                                                  const int *__restrict__ col, const V *__restrict__ val,
                                                  const V *__restrict__ x, V *__restrict__ y,
                                                  const int *done, PushPlan pp)
{
    if (PUSH && (int)blockIdx.x < pp.nblocks) { push_block(pp, blockIdx.x); return; }
    if (PUSH && pp.nrecv > 0 && (int)blockIdx.x >= (int)gridDim.x - pp.nrecv) { recv_block(pp, (int)blockIdx.x - ((int)gridDim.x - pp.nrecv)); return; }
    const int bid = PUSH ? blockIdx.x - pp.nblocks : blockIdx.x;
    constexpr int CH = CHE;
    __shared__ __attribute__((aligned(16))) V sval[CH];
    __shared__ __attribute__((aligned(16))) int scol[CH];
    if (done && *done) return;
    bool mine;
    (void)lds1_block<V, R, ACC, CHE>(bid, n, rowptr, col, val, x, y, sval, scol, mine);
}

// ---- the same product carrying a dot: the Krylov loops follow every A.x with y.u (CG / PCG: d.Ad, CGS / BiCGStab: Ap.r0,
// As.s and As.As; lcg.cpp:234, 389, 548-552, 720-724, 735-740) -- a pass of its own over two vectors and, on small systems, a
// launch of its own on the critical path.  Here the lanes that write y multiply it with u on the way out (u is requested
// before the block's stream, so the load hides behind it) and the workgroup leaves ONE partial sum per running sum in its
// slot of Ctx::ax_partials, which the consuming scalar step adds up in index order (devcommon.hpp: reduce_partials,
// PartCount) -- no atomics, the same bits from call to call.  y itself is bit-identical to the plain kernel's.
template <int R>
__global__ __launch_bounds__(VB) void k_spmv_lds1d(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                   const double *__restrict__ val, const double *__restrict__ x,
                                                   double *__restrict__ y, const int *done, DotPlan dp)
{
    constexpr int CH = LdsCfg<double>::CH;
    __shared__ __attribute__((aligned(16))) double sval[CH];
    __shared__ __attribute__((aligned(16))) int scol[CH];
    __shared__ double wsum[2][VB / 64];
    if (done && *done) return;
    const int rl = threadIdx.x % R, j0 = threadIdx.x / R;
    const long row = (long)blockIdx.x * R + rl;
    const double uv = dp.u[(j0 == 0 && row < n) ? row : 0];
    bool mine;
    const double v = lds1_block<double, R, false>(blockIdx.x, n, rowptr, col, val, x, y, sval, scol, mine);
    double a0 = mine ? v * uv : 0.0, a1 = mine ? v * v : 0.0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int NW = R >= 64 ? R / 64 : 1;        // wavefronts that hold finished rows
    if (w < NW) {
        a0 = wave_sum(a0); a1 = wave_sum(a1);
        if (lane == WSUM_LANE) { wsum[0][w] = a0; wsum[1][w] = a1; }
    }
    if (NW > 1) __syncthreads();
    if (threadIdx.x < 2 && (threadIdx.x == 0 || dp.yy)) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NW; q++) t += wsum[threadIdx.x][q];
        dp.part[threadIdx.x * AXP_CAP + blockIdx.x] = t;
    }
}

// ---- windowed kernel: some block's rows are too long for one window; the slice is walked
// window by window (same mapping, general row clipping)
template <class V, int R, bool ACC, bool PUSH = false>
__global__ __launch_bounds__(VB) void k_spmv_ldsw(int n, long nnz, const int *__restrict__ rowptr,
                                                  const int *__restrict__ col, const V *__restrict__ val,
                                                  const V *__restrict__ x, V *__restrict__ y,
                                                  const int *done, PushPlan pp)
{
    if (PUSH && (int)blockIdx.x < pp.nblocks) { push_block(pp, blockIdx.x); return; }
    if (PUSH && pp.nrecv > 0 && (int)blockIdx.x >= (int)gridDim.x - pp.nrecv) { recv_block(pp, (int)blockIdx.x - ((int)gridDim.x - pp.nrecv)); return; }
    const int bid = PUSH ? blockIdx.x - pp.nblocks : blockIdx.x;
    constexpr int T = VB / R;
    constexpr int CH = LdsCfg<V>::CH;
    constexpr int VU = sizeof(V) / 4;
    __shared__ __attribute__((aligned(16))) V sval[CH];
    __shared__ __attribute__((aligned(16))) int scol[CH];
    V(*sred)[R] = reinterpret_cast<V(*)[R]>(sval);
    if (done && *done) return;

    const int tid = threadIdx.x;
    const int row0 = bid * R;
    const int nrows = min(R, n - row0);
    const int rl = tid % R, j0 = tid / R;
    const int s = rowptr[row0], e = rowptr[row0 + nrows];
    int rs = 0, re = 0;
    if (rl < nrows) { rs = rowptr[row0 + rl]; re = rowptr[row0 + rl + 1]; }
    V acc = vzero(V());
    for (int base = s & ~3; base < e; base += CH) {
        const int cnt = min(CH, e - base);
        for (int u = tid * 4; u < cnt; u += VB * 4) {
            const long g = (long)base + u;
            if (g + 3 < nnz) {
                *reinterpret_cast<v4i *>(scol + u) = *reinterpret_cast<const v4i *>(col + g);
#pragma unroll
                for (int q = 0; q < VU; q++) reinterpret_cast<v2d *>(sval + u)[q] = reinterpret_cast<const v2d *>(val + g)[q];
            } else {
                for (int q = 0; q < 4 && g + q < nnz; q++) { scol[u + q] = col[g + q]; sval[u + q] = val[g + q]; }
            }
        }
        __syncthreads();
        const int lo = max(rs, base), hi = min(re, base + cnt);
        int k = rs + j0;
        if (k < lo) k += ((lo - k + T - 1) / T) * T;
        for (; k + T < hi; k += 2 * T) {
            const int c0 = scol[k - base], c1 = scol[k + T - base];
            const V a0 = sval[k - base], a1 = sval[k + T - base];
            acc = mac(a0, x[c0], acc);
            acc = mac(a1, x[c1], acc);
        }
        if (k < hi) acc = mac(sval[k - base], x[scol[k - base]], acc);
        __syncthreads();
    }
    if (T > 1) {
        sred[j0][rl] = acc;
        __syncthreads();
        if (j0 == 0 && rl < nrows) {
            V v = sred[0][rl];
#pragma unroll
            for (int j = 1; j < T; j++) v = vadd(v, sred[j][rl]);
            y[row0 + rl] = ACC ? vadd(y[row0 + rl], v) : v;
        }
    } else if (rl < nrows) {
        y[row0 + rl] = ACC ? vadd(y[row0 + rl], acc) : acc;
    }
}


// ---- packed columns ----------------------------------------------------------------------------
// The one lever left on the stream itself: a block of 64 rows stores its columns relative to the
// block's smallest column, six 21-bit fields per 16 bytes (2.67 instead of 4 B per entry: 10.67
// instead of 12 B of stream per entry), unpacked on the way into LDS.  Same row mapping and the
// same summation order as k_spmv_lds1 (bit-identical y), eight gathers in flight per lane.
// Measured on the headline system: 0.686 vs 0.719 ms (-4.6 %).  Costs 0.87 GB beside the plain
// `col` (which the other operations keep using), built on the device at the first product.
typedef unsigned long long u64;
typedef int v2i __attribute__((ext_vector_type(2)));
constexpr int PK_SPAN = 1 << 21;

// A RUN block (round 2, third session): every row of the block has the same number of entries L and every entry's column is
// one more than the entry in the same slot of the row above -- the shape of constant diagonals and stencils away from the
// matrix edges.  Such a block needs no columns of its own beyond row 0's: column(row r, slot k) = column(row 0, slot k) + r.
// It stores row 0's L columns as plain int32 (four per 16-byte group) and is marked by base[b] = -1 - L.
// A TEMPLATE block (round 3): what a stencil's blocks look like where the grid's boundaries pass through them.  The rows do not
// all hold the same entries any more, but every entry of the block lies on one of D <= 64 diagonals (column - row in block) -- the
// union of its rows' --, no row holds more than 32 entries, and every row holds them in ascending column order.  Such a block stores the D
// offsets and one 64-bit mask per row (which diagonals the row has) -- 39 groups of 16 bytes for a 27-point stencil instead of ~250 -- and is
// marked by base[b] = -(1 << 30) - D.  (Stencils with several unknowns per grid point need the width: 7 neighbours x 3 unknowns lie on 35.)  Run blocks are the special case "all masks full"; they keep their own, cheaper, form.
constexpr int TPL_MAXD = 64;         // diagonals of a template block (one bit each in a row's 64-bit mask)
constexpr int TPL_MAXROW = 32;       // longest row of a template block (the packed kernel takes a row in ONE predicated batch of UNR * T >= 32 entries)
constexpr int TPL_CODE = 1 << 30;       // base[b] = -TPL_CODE - D; ngroups[b] (before k_pk_groups) = -TPL_GRP - D
constexpr int TPL_GRP = 1 << 20;

// One wavefront, lane = row of the block: builds tpl[0 .. D) = the ascending offsets (column - row in block) the block's entries
// lie on -- seeded with the longest row's, then completed by the rows that have others (a block that holds the last line of one
// grid plane and the first line of the next has rows that miss DIFFERENT neighbours) -- and returns the lane's mask over them;
// *bad when there are more than TPL_MAXD of them or a row's columns do not ascend.  tpl: TPL_MAXD + 1 ints of LDS, dcount: one.
__device__ __forceinline__ unsigned long long tpl_row_mask(long row0, int r1, const int *__restrict__ rowptr, const int *__restrict__ col, int *tpl,
                                                 int *dcount, int *D_out, bool *bad)
{
    const int lane = threadIdx.x & 63;
    const long row = row0 + lane;
    int s = 0, len = 0;
    if (row < r1) { s = rowptr[row]; len = rowptr[row + 1] - s; }
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    // the longest row (first of them) seeds the template
    int best = len, who = lane;
    for (int off = 32; off > 0; off >>= 1) {
        const int ob = __shfl_down(best, off, 64), ow = __shfl_down(who, off, 64);
        if (ob > best || (ob == best && ow < who)) { best = ob; who = ow; }
    }
    best = __shfl(best, 0, 64); who = __shfl(who, 0, 64);
    *D_out = best;
    *bad = true;
    if (best < 1 || best > TPL_MAXROW) return 0ull;
    bool mine_bad = false;
    {   // columns must ascend strictly: entry e of a row is then the e-th set bit of its mask
        int prev = -0x7fffffff - 1;
        for (int k = 0; k < len; k++) { const int c = col[s + k]; if (c <= prev) mine_bad = true; prev = c; }
    }
    if (__ballot(mine_bad)) return 0ull;
    if (lane == who) { for (int k = 0; k < len; k++) tpl[k] = col[s + k] - lane; *dcount = len; }
    wave_sync();
    auto find = [&](int o, int D) {     // index of o in tpl[0 .. D), or -1
        int lo = 0, hi = D - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (tpl[mid] < o) lo = mid + 1; else hi = mid; }
        return tpl[lo] == o ? lo : -1;
    };
    for (int round = 0; round < 64; round++) {      // every round takes in one more row's offsets: at most 64
        const int D = *dcount;
        bool missing = false;
        for (int k = 0; k < len && !missing; k++) missing = find(col[s + k] - lane, D) < 0;
        const unsigned long long m = __ballot(missing);
        if (!m) break;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) {
            int Dn = D;
            for (int k = 0; k < len; k++) {
                const int o = col[s + k] - lane;
                if (find(o, Dn) >= 0) continue;
                if (Dn >= TPL_MAXD) { Dn = TPL_MAXD + 1; break; }
                int p = Dn;
                while (p > 0 && tpl[p - 1] > o) { tpl[p] = tpl[p - 1]; p--; }
                tpl[p] = o; Dn++;
            }
            *dcount = Dn;
        }
        wave_sync();
        if (*dcount > TPL_MAXD) return 0ull;
    }
    const int D = *dcount;
    *D_out = D;
    unsigned long long mask = 0ull;
    bool b = false;
    for (int k = 0; k < len; k++) { const int p = find(col[s + k] - lane, D); if (p >= 0) mask |= 1ull << p; else b = true; }
    *bad = __ballot(b) != 0ull;
    return mask;
}

__global__ __launch_bounds__(64) void k_pk_meta(int n, const int *rowptr, const int *col, int *base, int *ngroups, int *maxspan, int runs, int R)
{   // ngroups[b] = entries of block b for now, -L for a run block, -TPL_GRP - D for a template block (k_pk_groups turns them into groups);
    // maxspan[0] = widest block, [1] = longest row, [2] = run blocks, [3] = template blocks
    // R: rows per block -- 64, or 32 / 16 for long rows (then runs == 0: run and template blocks are shapes of 64 rows, one per lane)
    __shared__ int tpl[TPL_MAXD + 1], dcount;
    const int b = blockIdx.x;
    const long row0 = (long)b * R;
    const int r1 = (int)min((long)n, row0 + R);
    const int s = rowptr[row0], e = rowptr[r1];
    int lo = 0x7fffffff, hi = 0;
    for (int k = s + threadIdx.x; k < e; k += 64) { const int c = col[k]; lo = min(lo, c); hi = max(hi, c); }
    int len = 0, lmin = 0x7fffffff;
    if (row0 + threadIdx.x < r1) { len = rowptr[row0 + threadIdx.x + 1] - rowptr[row0 + threadIdx.x]; lmin = len; }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); len = max(len, __shfl_down(len, off, 64));
        lmin = min(lmin, __shfl_down(lmin, off, 64));
    }
    const int L = __shfl(len, 0, 64), Lmin = __shfl(lmin, 0, 64);
    int run = runs && L == Lmin && L > 0 && L <= 0x3fffffff;
    if (run) {      // uniform
        int bad = 0;
        for (int k = s + L + threadIdx.x; k < e; k += 64) bad |= col[k] != col[k - L] + 1;
        run = __ballot(bad) == 0;
    }
    int D = 0;
    bool tplb = false;
    if (!run && runs == 1 && e > s) {       // uniform (runs == 2: run blocks only, LCG_HIP_PACKED_TEMPLATES=0)
        bool bad;
        (void)tpl_row_mask(row0, r1, rowptr, col, tpl, &dcount, &D, &bad);
        tplb = !bad;        // (uniform)
    }
    if (threadIdx.x == 0) {
        if (e == s) { lo = 0; hi = 0; }
        base[b] = run ? -1 - L : (tplb ? -TPL_CODE - D : lo);
        ngroups[b] = run ? -L : (tplb ? -TPL_GRP - D : e - s);
        if (!tplb && !run) atomicMax(maxspan, hi - lo);     // (only the blocks that keep packed columns store them relative to their smallest)
        atomicMax(maxspan + 1, L);
        if (run) atomicAdd(maxspan + 2, 1);
        if (tplb) atomicAdd(maxspan + 3, 1);
    }
}

__global__ void k_pk_groups(int nb, int per, int *ngroups)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nb) {
        const int g = ngroups[b];
        // (a template block: its D offsets, then a mask per row -- 32 bits wide up to 32 diagonals, 64 beyond)
        // (a run block: row 0's L columns and, behind them, the slot of the block's own diagonal)
        ngroups[b] = g <= -TPL_GRP ? (-g - TPL_GRP + 1 + 3) / 4 + (-g - TPL_GRP <= 32 ? PK_R / 4 : PK_R / 2)
                                   : (g < 0 ? (-g + 1 + 3) / 4 : (per > 0 ? (g + per - 1) / per : 0));
    }
}

// Field j of a 128-bit group (lo, hi): BITS = 21 -> six fields, three per 64-bit half; BITS = 18 -> seven
// fields, the fourth straddling the halves.
template <int BITS> __device__ __forceinline__ int pk_field(u64 lo, u64 hi, int j);
template <> __device__ __forceinline__ int pk_field<21>(u64 lo, u64 hi, int j)
{
    return (int)(((j < 3 ? lo : hi) >> (21 * (j % 3))) & 0x1fffff);
}
template <> __device__ __forceinline__ int pk_field<18>(u64 lo, u64 hi, int j)
{
    const int sh = 18 * j;
    const u64 v = sh + 18 <= 64 ? lo >> sh : (sh >= 64 ? hi >> (sh - 64) : (lo >> sh) | (hi << (64 - sh)));
    return (int)(v & 0x3ffff);
}

// Several unknowns per grid point (a 27-point stencil x 3: 81 entries per row): every row's entries come in groups of g consecutive
// columns -- the g x g coupling block of two points.  ok[g - 2] stays 1 when EVERY row is made of such groups (g = 2, 3, 4): the packed
// form of the long-row blocks then keeps one column field per group (18 bits for three entries: 8.75 instead of 10.3 B per entry).
__global__ __launch_bounds__(VB) void k_pk_dof(int n, const int *__restrict__ rowptr, const int *__restrict__ col, int *ok)
{
    const int i = blockIdx.x * VB + threadIdx.x;
    unsigned bad = 0;       // bit g - 2
    if (i < n) {
        const int s = rowptr[i], len = rowptr[i + 1] - s;
        bad = (len % 2 != 0 ? 1u : 0u) | (len % 3 != 0 ? 2u : 0u) | (len % 4 != 0 ? 4u : 0u);
        int prev = len > 0 ? col[s] : 0;
        for (int k = 1; k < len && bad != 7u; k++) {        // one walk for the three group sizes
            const int c = col[s + k];
            if (c != prev + 1) bad |= (k % 2 != 0 ? 1u : 0u) | (k % 3 != 0 ? 2u : 0u) | (k % 4 != 0 ? 4u : 0u);
            prev = c;
        }
    }
    for (int g = 0; g < 3; g++) if (__ballot((bad >> g) & 1u) != 0ull && (threadIdx.x & 63) == 0) ok[g] = 0;
}

template <int BITS>
__global__ __launch_bounds__(VB) void k_pk_pack(int n, const int *rowptr, const int *col, const int *base, const int *pofs, v4i *packed, int R, int dof)
{
    constexpr int PER = BITS > 0 ? 128 / BITS : 1;
    const int b = blockIdx.x;
    const long row0 = (long)b * R;
    const int r1 = (int)min((long)n, row0 + R);
    const int s = rowptr[row0], e = rowptr[r1];
    const int bs = base[b];
    if (BITS == 0 && bs >= 0) return;       // runs only: the other blocks keep nothing
    if (bs <= -TPL_CODE) {   // template block: the D offsets (padded to whole groups), then the 64 row masks
        __shared__ int tpl[TPL_MAXD + 1], dcount;
        if (threadIdx.x < 64) {
            int D; bool bad;
            const unsigned long long m = tpl_row_mask(row0, r1, rowptr, col, tpl, &dcount, &D, &bad);
            int *dst = reinterpret_cast<int *>(packed + pofs[b]);
            const int Dp = (D + 1 + 3) & ~3;
            // behind the D offsets: which of them is the block's own diagonal (offset = the block's first row) when EVERY row of the block
            // has it, else -1 -- k_spmv_ldsp<DOT> then takes u = x from the gathers (as for run blocks)
            int idg = -1;
            for (int i = 0; i < D; i++) if (tpl[i] == (int)row0) idg = i;
            const bool lacks = idg < 0 || (row0 + (long)threadIdx.x < r1 && !((m >> idg) & 1ull));
            if (__ballot(lacks)) idg = -1;
            if ((int)threadIdx.x < D) dst[threadIdx.x] = tpl[threadIdx.x];
            if (threadIdx.x == 0) { dst[D] = idg; for (int i = D + 1; i < Dp; i++) dst[i] = 0; }      // (D may be 64: one lane writes the tail)
            if (D <= 32) reinterpret_cast<unsigned *>(dst + Dp)[threadIdx.x] = (unsigned)m;
            else reinterpret_cast<unsigned long long *>(dst + Dp)[threadIdx.x] = m;      // (Dp ints = whole 16-byte groups: aligned)
        }
        return;
    }
    if (bs < 0) {       // run block: row 0's columns as they are
        const int L = -1 - bs;
        int *dst = reinterpret_cast<int *>(packed + pofs[b]);
        // behind them: the slot whose column is the block's first row (every row's diagonal, then), or -1: k_spmv_ldsp<DOT> takes u = x from it
        __shared__ int kd;
        if (threadIdx.x == 0) kd = -1;
        __syncthreads();
        for (int k = threadIdx.x; k < L; k += VB) if (col[s + k] == (int)row0) kd = k;       // (columns ascend: one match at most)
        __syncthreads();
        for (int k = threadIdx.x; k < ((L + 1 + 3) & ~3); k += VB) dst[k] = k < L ? col[s + k] : (k == L ? kd : 0);
        return;
    }
    const int ng = (e - s + PER * dof - 1) / (PER * dof);       // (dof > 1: a field is the first column of a group of dof entries)
    for (int g = threadIdx.x; g < ng; g += VB) {
        u64 lo = 0, hi = 0;
        for (int j = 0; j < PER; j++) {
            const int k = s + (PER * g + j) * dof;
            const u64 c = k < e ? (u64)(col[k] - bs) : 0;
            if (BITS == 21) { if (j < 3) lo |= c << (21 * j); else hi |= c << (21 * (j - 3)); }
            else if (BITS > 0) {
                const int sh = BITS * j;
                if (sh < 64) { lo |= c << sh; if (sh + BITS > 64) hi |= c >> (64 - sh); }
                else hi |= c << (sh - 64);
            }
        }
        v4i o; o.x = (int)(unsigned)lo; o.y = (int)(unsigned)(lo >> 32); o.z = (int)(unsigned)hi; o.w = (int)(unsigned)(hi >> 32);
        packed[pofs[b] + g] = o;
    }
}

// NS: gathers a lane keeps in flight.  NS = 8: batches of 8 while they are full, then one by one.
// NS > 8: the FIRST batch is predicated (slots past the row's end read entry 0 with a zero
// coefficient), so rows of up to NS*T entries -- 33 entries on 4 lanes are 9 for one lane, 8 for the
// others -- are done in one round without a serial tail (0.706 vs 0.722 ms on the headline system).
// end of a block of k_spmv_ldsp<DOT>: the first wavefront holds the finished rows (vfin) and their u; one sum per block
__device__ __forceinline__ void ldsp_dot_tail(const DotPlan &dp, int bid, int j0, double vfin, double uv)
{
    if (threadIdx.x >= 64) return;      // uniform per wavefront: the finished rows sit in wavefront 0 (all of it at 64 rows per block,
    if (j0 != 0) vfin = 0.0;            // its first 32 / 16 lanes at 32 / 16 -- the other lanes add zeros)
    const double a0 = wave_sum(vfin * uv);
    if ((threadIdx.x & 63) == WSUM_LANE) dp.part[bid] = a0;
    if (dp.yy) {                        // uniform
        const double a1 = wave_sum(vfin * vfin);
        if ((threadIdx.x & 63) == WSUM_LANE) dp.part[dp.stride + bid] = a1;
    }
}

// DOT: the block also leaves its rows' share of y.u (and y.y) in dp.part[block] (dp.part[dp.stride + block]): the dot the
// Krylov loops take right after the product, without a pass of its own over two 80 MB vectors (see k_spmv_lds1d; here one
// partial per block of 64 rows, folded to <= 512 by k_axp_fold before the scalar step adds them up).
// CHE: entries a block holds at most -- the LDS window, about 12 bytes per entry.  With the 2240 entries the shapes are chosen for a
// workgroup asks for 26.9 KB and FIVE fit a CU (the LDS is handed out in granules: six would need <= 26.6 KB each).  Where every
// block of the matrix is smaller the window shrinks: 2208 entries (26.5 KB, six workgroups per CU: the headline's 64 x 33 -- same-box
// A/B 1477 -> 1498 CG iterations/s, 6 of 6 runs), 1872 (22.5 KB, seven: a 27-point stencil's 64 x 27 -- product 401 -> 374 us with
// six already), 1696 (20.4 KB, eight: rows of up to 26 entries).
constexpr int PK_CH_SMALL = 2208;
constexpr int PK_CH_7 = 1872, PK_CH_8 = 1696;
static DotPlan ldsp_plain_plan() { DotPlan d; d.ystore = y_store_policy(); return d; }
static int pk_window(int max_slice)
{
    static const int env = [] { const char *e = std::getenv("LCG_HIP_PACKED_WINDOW"); return e ? atoi(e) : 0; }();     // A/B runs: least window
    const int need = std::max(max_slice, env);
    return need <= PK_CH_8 ? PK_CH_8 : need <= PK_CH_7 ? PK_CH_7 : need <= PK_CH_SMALL ? PK_CH_SMALL : LdsCfg<double>::CH;
}
// RR: rows per block.  64 (T = 4 lanes per row) for rows of up to ~34 entries; 32 / 16 (T = 8 / 16) for longer rows -- packed columns
// only, the first batch predicated for ANY NS (round 3, late: 27-point stencils with 2 / 3 unknowns per point, 54 / 81 entries per row,
// took the 12-byte k_spmv_lds1 before).
template <bool PUSH, int NS, int BITS, bool DOT = false, int CHE = LdsCfg<double>::CH, int RR = PK_R>
__global__ __launch_bounds__(VB) void k_spmv_ldsp(int n, const int *__restrict__ rowptr, const v4i *__restrict__ packed,
                                                  const int *__restrict__ pofs, const int *__restrict__ pbase,
                                                  const double *__restrict__ val, const double *__restrict__ x,
                                                  double *__restrict__ y, const int *done, PushPlan pp, DotPlan dp)
{
    if (PUSH && (int)blockIdx.x < pp.nblocks) { push_block(pp, blockIdx.x); return; }
    if (PUSH && pp.nrecv > 0 && (int)blockIdx.x >= (int)gridDim.x - pp.nrecv) { recv_block(pp, (int)blockIdx.x - ((int)gridDim.x - pp.nrecv)); return; }
    const int bid = PUSH ? blockIdx.x - pp.nblocks : blockIdx.x;
    constexpr int R = RR;
    constexpr int T = VB / R;
    constexpr int UNR = NS;
    constexpr int CH = CHE;                             // entries per block at most (checked by the host)
    constexpr int PER = 128 / BITS;                     // columns per 16-byte group: 6 (21 bits) or 7 (18 bits)
    constexpr int NG = (CH + PER - 1) / PER;            // groups
    constexpr int GR = (NG + VB - 1) / VB;              // rounds of 16-byte group loads
    constexpr int VR = (CH / 2 + 1 + VB - 1) / VB;      // rounds of 16-byte val loads
    __shared__ __attribute__((aligned(16))) double sval[CH + 2];
    __shared__ __attribute__((aligned(16))) int scol[NG * PER];
    double(*sred)[R] = reinterpret_cast<double(*)[R]>(sval);
    if (done && *done) return;

    const int tid = threadIdx.x;
    const int row0 = bid * R;
    const int nrows = min(R, n - row0);
    const int rl = tid % R, j0 = tid / R;
    const int s = rowptr[row0], e = rowptr[row0 + nrows];
    const int dof = RR == 64 ? 1 : dp.dof;             // (uniform; groups of consecutive columns are a shape of the long-row blocks)
    const int cnt = e - s, ng = (cnt + PER * dof - 1) / (PER * dof);
    const int po = pofs[bid], bs = pbase[bid];
    const int bv = s & ~1, cntv = e - bv;
    double uv = 0.0;
    // (u == x and a run block that holds its own diagonal: u of row r IS one of the block's gathers, x[column(0, kd) + r] with
    //  column(0, kd) = the block's first row -- 80 MB less to read per product of the headline system.  kd comes with the block's columns:
    //  searching row 0's columns here, 33 dependent scalar loads in front of the stream's requests, made the product 22 % SLOWER.)
    int kd = -1;
    if (DOT && RR == 64 && dp.ux && bs < 0)     // (k_pk_pack left it behind row 0's L columns / the template's D offsets: -1 = not every row of the block holds its diagonal)
        kd = reinterpret_cast<const int *>(packed + po)[bs > -TPL_CODE ? -1 - bs : -bs - TPL_CODE];
    if (DOT && kd < 0) uv = dp.u[(j0 == 0 && rl < nrows) ? row0 + rl : 0];     // requested ahead of the stream: hidden behind it

    if (RR == 64 && bs <= -TPL_CODE) {     // (run and template blocks are shapes of 64 rows: the builder marks none at 32 / 16 rows per block)
        // TEMPLATE block (k_pk_meta): every entry lies on one of D <= 64 diagonals and every row says by a mask which of them it
        // has.  As in a run block nothing but the values streams and the x gathers go out beside the value stream -- their
        // addresses come from the row's mask and the D offsets (held one per lane, fetched by a wavefront shuffle), not from staged
        // columns.  Entry e of a row is the e-th set bit of its mask (columns ascend); lane (row, j0) takes entries j0, j0 + T, ...
        // and the partial sums meet as in the general path: the same bits.
        // (NS < 8 is launched only for matrices whose LONGEST row has NS * T entries at most: its template blocks' rows fit too)
        static_assert(UNR * T >= TPL_MAXROW || NS < 8, "one predicated batch covers the longest row of a template block");
        const int D = -bs - TPL_CODE;
        const int *tplp = reinterpret_cast<const int *>(packed + po);
        const int *mskp = tplp + ((D + 1 + 3) & ~3);       // 64 masks: 32 bits wide up to 32 diagonals, 64 beyond (tplp[D]: the diagonal's index)
        const bool live = rl < nrows;
        double vfin = 0.0;
        // two instances of the same code, chosen by a scalar branch: WIDE (33 .. 64 diagonals) works on 64-bit masks, the other --
        // every plain stencil -- on 32-bit ones (the wide arithmetic costs the narrow case 4-8 %, measured)
        auto block = [&](auto wide_tag) {
            constexpr bool WIDE = decltype(wide_tag)::value;
            typedef typename std::conditional<WIDE, unsigned long long, unsigned>::type M;
            // the small loads FIRST: the gathers wait for the mask, and a wait for a load issued behind the value stream is a wait
            // for the value stream (vmcnt counts in order) -- two memory latencies per block instead of one
            const M mask = live ? reinterpret_cast<const M *>(mskp)[rl] : (M)0;
            const int myoff = tplp[rl < D ? rl : 0];            // lane l < D of every wavefront holds offset l
            const int rs = (live ? rowptr[row0 + rl] : bv) - bv;    // this lane's row in the staged values
            v2d pv[VR];
#pragma unroll
            for (int r = 0; r < VR; r++) {
                const int u = 2 * (tid + r * VB);
                pv[r] = *reinterpret_cast<const v2d *>(val + (long)bv + (u < cntv ? u : 0));
            }
            const int len = WIDE ? __popcll((unsigned long long)mask) : __popc((unsigned)mask);
            const M full = D >= (WIDE ? 64 : 32) ? ~(M)0 : (((M)1 << D) - (M)1);
            const bool dense = __ballot(live && mask != full) == 0ull;      // uniform: every live row has all D diagonals
            // Entry j0 + q T of a row is the (j0 + q T)-th set bit of its mask.  The lane walks its mask instead of searching it: drop
            // the j0 lowest set bits once (j0 is the wavefront's number: a uniform loop), then per entry read the lowest set bit and drop
            // T -- 9 integer instructions per entry where the halving search took about 30 (7-point x 3 unknowns: 300 -> see DESIGN)
            M m = mask;
            if (!dense)
                for (int i = __builtin_amdgcn_readfirstlane(j0); i > 0; i--) m &= m - (M)1;
            double xv[UNR];
#pragma unroll
            for (int q = 0; q < UNR; q++) {
                int slot = j0 + q * T;
                if (!dense) {       // uniform
                    slot = m ? (WIDE ? __builtin_ctzll((unsigned long long)m) : __builtin_ctz((unsigned)m)) : 0;
#pragma unroll
                    for (int t = 0; t < T; t++) m &= m - (M)1;
                }
                const int off = __shfl(myoff, slot & 63, 64);
                const bool ok = live && j0 + q * T < len;
                xv[q] = x[ok ? rl + off : 0];       // (the offsets are row 0's columns, as in a run block: column = offset + row in block)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < VR; r++) {
                const int u = 2 * (tid + r * VB);
                if (u < cntv) *reinterpret_cast<v2d *>(sval + u) = pv[r];
            }
            __syncthreads();
            double acc = 0.0;
            // (the diagonal is entry e_d of this row -- the set bits of its mask below the diagonal's -- and lives in lane e_d mod T, batch e_d / T)
            double *sud = reinterpret_cast<double *>(scol);
            const int ed = (DOT && kd >= 0) ? (WIDE ? __popcll((unsigned long long)(mask & (((M)1 << kd) - (M)1))) : __popc((unsigned)(mask & (((M)1 << kd) - (M)1)))) : -1;
            const int qd = (ed >= 0 && ed % T == j0) ? ed / T : -1;
#pragma unroll
            for (int q = 0; q < UNR; q++) {
                const int e = j0 + q * T;
                acc = (live && e < len) ? fma(sval[rs + e], xv[q], acc) : acc;
                if (DOT && q == qd && live) sud[rl] = xv[q];
            }
            __syncthreads();
            sred[j0][rl] = acc;
            __syncthreads();
            if (DOT && kd >= 0 && j0 == 0 && live) uv = sud[rl];
            if (j0 == 0 && live) {
                double v = sred[0][rl];
#pragma unroll
                for (int j = 1; j < T; j++) v += sred[j][rl];
                store_y(y + row0 + rl, v, dp.ystore);
                vfin = v;
            }
        };
        if (D > 32) block(std::true_type{}); else block(std::false_type{});
        if (DOT) ldsp_dot_tail(dp, bid, j0, vfin, uv);
        return;
    }
    if (RR == 64 && bs < 0) {
        // RUN block (k_pk_meta): every row holds L entries and column(row r, slot k) = column(row 0, slot k) + r.  Nothing but
        // the values streams; the columns are row 0's L integers, read through the scalar cache (a wavefront is one slot j0 of
        // 64 rows: its k is uniform), and -- the point -- the x gathers no longer wait for the staged columns: they go out
        // TOGETHER with the value stream, one memory latency per block instead of two.  Entry order per lane and the order of
        // the additions are those of the general path below: the same bits.
        const int L = -1 - bs;
        const int w = __builtin_amdgcn_readfirstlane(j0);
        const int *bcol = reinterpret_cast<const int *>(packed + po);
        const bool live = rl < nrows;
        v2d pv[VR];
#pragma unroll
        for (int r = 0; r < VR; r++) {
            const int u = 2 * (tid + r * VB);
            // (non-temporal value loads, so that the stream would not push x out of the L2: 581 vs 520 us -- measured and removed)
            pv[r] = *reinterpret_cast<const v2d *>(val + (long)bv + (u < cntv ? u : 0));
        }
        double xv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; q++) {
            const int k = w + q * T;                     // uniform
            const int c0 = bcol[k < L ? k : 0];           // scalar load
            xv[q] = x[(k < L && live) ? c0 + rl : 0];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < VR; r++) {
            const int u = 2 * (tid + r * VB);
            if (u < cntv) *reinterpret_cast<v2d *>(sval + u) = pv[r];
        }
        __syncthreads();
        double acc = 0.0;
        const int rs = s + rl * L - bv;                    // first entry of this lane's row in the staged values
        double *sud = reinterpret_cast<double *>(scol);     // (a run block stages no columns: its u, where the diagonal is among the gathers)
        if (live) {
#pragma unroll
            for (int q = 0; q < UNR; q++) {
                const int k = w + q * T;
                acc = k < L ? fma(sval[rs + k], xv[q], acc) : acc;
                if (DOT && k == kd) sud[rl] = xv[q];        // (uniform: one wavefront, one q)
            }
            for (int k = w + UNR * T; k < L; k += T) {
                const double xk = x[bcol[k] + rl];
                acc = fma(sval[rs + k], xk, acc);
                if (DOT && k == kd) sud[rl] = xk;
            }
        }
        __syncthreads();
        sred[j0][rl] = acc;
        __syncthreads();
        if (DOT && kd >= 0 && j0 == 0 && live) uv = sud[rl];
        double vfin = 0.0;
        if (j0 == 0 && live) {
            double v = sred[0][rl];
#pragma unroll
            for (int j = 1; j < T; j++) v += sred[j][rl];
            store_y(y + row0 + rl, v, dp.ystore);
            vfin = v;
        }
        if (DOT) ldsp_dot_tail(dp, bid, j0, vfin, uv);
        return;
    }

    const int rsafe = rl < nrows ? rl : 0;              // (requested before the stream, branch-free: see lds1_block)
    int rs = rowptr[row0 + rsafe], re = rowptr[row0 + rsafe + 1];
    if (rl >= nrows) { rs = 0; re = 0; }
    v4i pg[GR]; v2d pv[VR];
    // (dof > 1: a group's fields are spread over dof lanes -- lane t takes group t / dof and, of each of its fields' dof consecutive
    //  columns, the (t % dof)-th: as many LDS stores per lane as with a field per column.  With the group's lanes taking all dof
    //  columns of a field each, a third of the lanes did three times the stores and the product got SLOWER: 1296 against 1166 us on
    //  the 27-point stencil x 3.)
    const unsigned dmul = dof == 3 ? 0x5556u : 0u;       // t / 3 = (t * 0x5556) >> 16 for t < 2^15
#pragma unroll
    for (int r = 0; r < GR; r++) {
        const int t = tid + r * VB;
        const int gi = dof == 1 ? t : (dof == 2 ? t >> 1 : (dof == 4 ? t >> 2 : (int)(((unsigned)t * dmul) >> 16)));
        pg[r] = packed[po + (gi < ng ? gi : 0)];
    }
#pragma unroll
    for (int r = 0; r < VR; r++) {
        const int u = 2 * (tid + r * VB);
        pv[r] = *reinterpret_cast<const v2d *>(val + (long)bv + (u < cntv ? u : 0));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < GR; r++) {
        const int t = tid + r * VB;
        const int gi = dof == 1 ? t : (dof == 2 ? t >> 1 : (dof == 4 ? t >> 2 : (int)(((unsigned)t * dmul) >> 16)));
        if (gi < ng) {
            const u64 lo = (u64)(unsigned)pg[r].x | ((u64)(unsigned)pg[r].y << 32);
            const u64 hi = (u64)(unsigned)pg[r].z | ((u64)(unsigned)pg[r].w << 32);
            if (dof > 1) {
                // one field per group of dof consecutive columns: the staged columns are what they would be with a field each
                const int d = t - gi * dof;
#pragma unroll
                for (int j = 0; j < PER; j++) {
                    const int i0 = (PER * gi + j) * dof + d;
                    if (i0 < cnt) scol[i0] = bs + pk_field<BITS>(lo, hi, j) + d;      // (fields past the block's end stay out: scol holds CH entries)
                }
            } else if (BITS == 21) {
                v2i a, b, c;
                a.x = bs + (int)(lo & 0x1fffff); a.y = bs + (int)((lo >> 21) & 0x1fffff);
                b.x = bs + (int)((lo >> 42) & 0x1fffff); b.y = bs + (int)(hi & 0x1fffff);
                c.x = bs + (int)((hi >> 21) & 0x1fffff); c.y = bs + (int)((hi >> 42) & 0x1fffff);
                v2i *dst = reinterpret_cast<v2i *>(scol + 6 * gi);
                dst[0] = a; dst[1] = b; dst[2] = c;
            } else {
#pragma unroll
                for (int j = 0; j < PER; j++) scol[PER * gi + j] = bs + pk_field<BITS>(lo, hi, j);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < VR; r++) {
        const int u = 2 * (tid + r * VB);
        if (u < cntv) *reinterpret_cast<v2d *>(sval + u) = pv[r];
    }
    __syncthreads();
    double acc = 0.0;
    int k = rs + j0;
    {   // the first batch is predicated for every NS (it was for NS > 8 only: rows of fewer than 8 T entries then went through the
        // serial loop at the end, one gather at a time -- 29 entries per row were slower than 33: 782 vs 759 us at 10M rows)
        int c[UNR]; double a[UNR], xv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; q++) {
            const int kk = k + q * T;
            const bool ok = kk < re;
            const int cc = scol[ok ? kk - s : 0];
            c[q] = ok ? cc : 0;         // an empty block never wrote scol[0]: column 0 is always a valid address
            a[q] = sval[ok ? kk - bv : 0];
        }
        // (non-temporal gathers: 3.86 instead of 1.44 ms on the row-random band, 0.97 instead of 0.75 on constant diagonals;
        //  L1-bypassing sc1 gathers: no difference -- measured in round 2 and removed)
#pragma unroll
        for (int q = 0; q < UNR; q++) xv[q] = x[c[q]];
#pragma unroll
        for (int q = 0; q < UNR; q++) acc = (k + q * T < re) ? fma(a[q], xv[q], acc) : acc;   // select, not a zero product: same bits as the plain loop
        k += UNR * T;
    }
    for (; k + (UNR - 1) * T < re; k += UNR * T) {
        int c[UNR]; double a[UNR], xv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; q++) { c[q] = scol[k + q * T - s]; a[q] = sval[k + q * T - bv]; }
#pragma unroll
        for (int q = 0; q < UNR; q++) xv[q] = x[c[q]];
#pragma unroll
        for (int q = 0; q < UNR; q++) acc = fma(a[q], xv[q], acc);
    }
    for (; k < re; k += T) acc = fma(sval[k - bv], x[scol[k - s]], acc);
    __syncthreads();
    sred[j0][rl] = acc;
    __syncthreads();
    double vfin = 0.0;
    if (j0 == 0 && rl < nrows) {
        double v = sred[0][rl];
#pragma unroll
        for (int j = 1; j < T; j++) v += sred[j][rl];
        store_y(y + row0 + rl, v, dp.ystore);
        vfin = v;
    }
    if (DOT) ldsp_dot_tail(dp, bid, j0, vfin, uv);
}

// The <= 512-way second stage of a product's per-block sums: block i adds the slice [i * per, (i + 1) * per) of `big` in a
// fixed order and leaves it in out[i] (the y.y sums, `stride` further on, in out[AXP_CAP + i]).
__global__ __launch_bounds__(VB) void k_axp_fold(const double *__restrict__ big, int nblk, int stride, int per, int yy, double *__restrict__ out,
                                                 const int *done)
{
    __shared__ double sh[2][VB / 64];
    if (done && *done) return;
    const int lo = blockIdx.x * per, hi = min(nblk, lo + per);
    double a0 = 0.0, a1 = 0.0;
    for (int j0 = lo; j0 < hi; j0 += 4 * VB) {
        double t[4], w[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = j0 + threadIdx.x + q * VB;
            t[q] = big[j < hi ? j : lo];
            w[q] = yy ? big[stride + (j < hi ? j : lo)] : 0.0;
            if (j >= hi) { t[q] = 0.0; w[q] = 0.0; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { a0 += t[q]; a1 += w[q]; }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == WSUM_LANE) { sh[0][wv] = a0; sh[1][wv] = a1; }
    __syncthreads();
    if (threadIdx.x < 2 && (threadIdx.x == 0 || yy)) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < VB / 64; q++) t += sh[threadIdx.x][q];
        out[threadIdx.x * AXP_CAP + blockIdx.x] = t;
    }
}

// ---- short rows (<= 17 entries on average: 5-point / 7-point stencils, the 1000 x 1000 Laplacian of BASELINE configs[1]) ---------
// The LDS-staged kernels put 256 or 128 rows of such a matrix into a block and spend it on staging ~1300 entries behind two
// barriers.  Where the 64-row blocks are RUNS (k_pk_meta) ONE WAVEFRONT owns a block, lane = row: the L columns of row 0 come
// through the scalar cache, the gathers of x (column + lane: one contiguous run per entry slot) go out at once with the
// block's values, which are read coalesced (the block's 64 L values are contiguous) and handed to their rows through a
// wavefront-private piece of LDS -- no barrier anywhere.  (Reading them straight from the row, 8 bytes at a stride of L
// doubles, was measured first: every 128-byte line is then fetched by L instructions and the L1 does not hold a CU's waves'
// lines in between -- 20.0 us on the 1M-row Laplacian against 15.9 for the staged kernel.)  The few blocks that are not runs
// (grid-row boundaries) are walked from the CSR arrays by the same wavefront.  Entry k of a row is added to partial sum
// k mod T, the partial sums in index order -- the arithmetic of k_spmv_lds1 with VB / T rows per block (T = 1 for up to 8.5
// entries per row, 2 up to 17), which is the kernel the automatic choice would otherwise take: the same bits.
// the work of one wavefront on block b (< number of 64-row blocks); returns the finished y of row 64 b + lane (0 past the end)
template <int T>
__device__ __forceinline__ double run1_block(int b, int wv, int n, int LP, const int *__restrict__ rowptr, const int *__restrict__ col,
                                             const double *__restrict__ val, const int *__restrict__ pofs,
                                             const int *__restrict__ pbase, const int *__restrict__ pcols,
                                             const double *__restrict__ x, double *__restrict__ y, double *wlds)
{
    constexpr int NB = 8;                               // entries in flight per lane and batch (a multiple of T)
    const int lane = threadIdx.x & 63;
    const long row0 = (long)b * 64;
    const int nrows = (int)min(64L, n - row0);
    const bool live = lane < nrows;
    const int bs = pbase[b], po = pofs[b], s = rowptr[row0];
    double acc[T];
#pragma unroll
    for (int j = 0; j < T; j++) acc[j] = 0.0;
    if (bs <= -TPL_CODE) {
        // TEMPLATE block (k_pk_meta): the block's entries lie on D <= 64 diagonals, a mask per row says which.  A uniform loop over
        // the diagonals: the offset is a scalar, the gather x[offset + lane] one run with holes, the row's value the
        // popcount-th of its entries -- no column is read.  Entry e goes to partial sum e mod T like everywhere: the same bits.
        const int D = -bs - TPL_CODE;
        const int *tplp = pcols + 4 * (long)po;
        const bool wide = D > 32;       // uniform: 64-bit masks beyond 32 diagonals
        unsigned long long mask = 0ull;
        if (live) mask = wide ? reinterpret_cast<const unsigned long long *>(tplp + ((D + 1 + 3) & ~3))[lane]
                              : (unsigned long long)reinterpret_cast<const unsigned *>(tplp + ((D + 1 + 3) & ~3))[lane];
        const unsigned mlo = (unsigned)mask, mhi = (unsigned)(mask >> 32);
        const int rs = live ? rowptr[row0 + lane] - s : 0;     // this lane's row in the block's values
        // the block's values, coalesced, into the wavefront's piece of LDS as they lie in memory (64 x LP doubles hold them: LP >= the
        // longest row); read from the rows directly -- a stride of one row per lane -- every batch re-fetched the rows' lines
        double *mine = wlds + (size_t)wv * 64 * LP;
        const int tot = rowptr[row0 + nrows] - s;
        // one batch of NB diagonals: which rows have diagonal k, which of their entries it is (recomputed where it is needed: a few
        // integer instructions are cheaper than the registers that would carry them past the staging loop), x[offset + lane]
        auto has = [&](int k) { return k < D && (((k < 32 ? mlo : mhi) >> (k & 31)) & 1u); };             // (k uniform: a scalar choice)
        auto entry = [&](int k) { return __popc((k < 32 ? mlo : mhi) & ((1u << (k & 31)) - 1u)) + (k < 32 ? 0 : __popc(mlo)); };
        auto gather = [&](int k0, double (&xv)[NB]) {
#pragma unroll
            for (int q = 0; q < NB; q++) {
                const int k = k0 + q;                       // uniform
                const int off = tplp[k < D ? k : 0];        // scalar load
                xv[q] = x[has(k) ? off + lane : 0];
            }
        };
        // the first batch's gathers leave with the values, as in a run block: one memory latency less per block
        double xv[NB];
        gather(0, xv);
        for (int i0 = 0; i0 < tot; i0 += NB * 64) {
            double v[NB];
#pragma unroll
            for (int q = 0; q < NB; q++) { const int i = i0 + q * 64 + lane; v[q] = val[s + (i < tot ? i : 0)]; }
#pragma unroll
            for (int q = 0; q < NB; q++) { const int i = i0 + q * 64 + lane; if (i < tot) mine[i] = v[q]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k0 = 0; k0 < D; k0 += NB) {
            if (k0 > 0) gather(k0, xv);
            double a[NB];
#pragma unroll
            for (int q = 0; q < NB; q++) a[q] = has(k0 + q) ? mine[rs + entry(k0 + q)] : 0.0;
#pragma unroll
            for (int q = 0; q < NB; q++) {
                const bool ok = has(k0 + q);
                if (T == 1) acc[0] = ok ? fma(a[q], xv[q], acc[0]) : acc[0];
                else {
                    const int e = entry(k0 + q);
#pragma unroll
                    for (int j = 0; j < T; j++) acc[j] = (ok && (e % T) == j) ? fma(a[q], xv[q], acc[j]) : acc[j];
                }
            }
        }
    } else if (bs < 0) {
        const int L = -1 - bs;
        const int *bcol = pcols + 4 * (long)po;
        double *mine = wlds + (size_t)wv * 64 * LP;
        const int tot = nrows * L;
        // the batch is as wide as the rows are long (4 / 6 / 8: L is uniform) -- with eight slots for every block a tridiagonal system
        // issued eight value loads and eight gathers per lane for three entries, the 5-point Laplacian for five
        auto run_rows = [&](auto nbt) {
            constexpr int NBX = decltype(nbt)::value;
            static_assert(NBX % T == 0, "entry k goes to partial sum k mod T");
            const float invL = 1.0f / (float)L;
            double xv[NBX];
#pragma unroll
            for (int q = 0; q < NBX; q++) {                  // the gathers of the first batch leave with the values
                const int c0 = bcol[q < L ? q : 0];
                xv[q] = x[(q < L && live) ? c0 + lane : 0];
            }
            for (int j0 = 0; j0 < L; j0 += NBX) {
                double v[NBX];
#pragma unroll
                for (int q = 0; q < NBX; q++) {
                    const int e = (j0 + q) * 64 + lane;
                    v[q] = val[s + ((j0 + q < L && e < tot) ? e : 0)];
                }
#pragma unroll
                for (int q = 0; q < NBX; q++) {
                    const int e = (j0 + q) * 64 + lane;
                    if (j0 + q < L && e < tot) {
                        int r = (int)(((float)e + 0.5f) * invL);        // e / L (e < 2^16: exact after the correction below)
                        r -= r * L > e; r += (r + 1) * L <= e;
                        mine[r * LP + (e - r * L)] = v[q];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const double *row = mine + lane * LP;
            for (int k0 = 0; k0 < L; k0 += NBX) {
                if (k0 > 0) {
#pragma unroll
                    for (int q = 0; q < NBX; q++) {
                        const int k = k0 + q;
                        const int c0 = bcol[k < L ? k : 0];
                        xv[q] = x[(k < L && live) ? c0 + lane : 0];
                    }
                }
#pragma unroll
                for (int q = 0; q < NBX; q++) {
                    const double a = (k0 + q < L && live) ? row[k0 + q] : 0.0;
                    acc[q % T] = (k0 + q < L) ? fma(a, xv[q], acc[q % T]) : acc[q % T];
                }
            }
        };
        if (L <= 4) run_rows(std::integral_constant<int, 4>{});
        else if (L <= 6) run_rows(std::integral_constant<int, 6>{});
        else run_rows(std::integral_constant<int, NB>{});
    } else {
        // not a run: every lane walks its own row out of the CSR arrays, NB entries at a time -- all columns and values of a
        // batch requested before the first gather, all gathers before the first product (a one-by-one walk is a chain of
        // 2 L dependent loads, and 13 % of the Laplacian's blocks take this path)
        int rs = 0, re = 0;
        if (live) { rs = rowptr[row0 + lane]; re = rowptr[row0 + lane + 1]; }
        // (the batch as wide as the block's longest row needs, like the run blocks above)
        int longest = re - rs;
        for (int off = 32; off > 0; off >>= 1) longest = max(longest, __shfl_xor(longest, off, 64));
        auto walk_rows = [&](auto nbt) {
            constexpr int NBX = decltype(nbt)::value;
            static_assert(NBX % T == 0, "entry k goes to partial sum k mod T");
            for (int k0 = rs; __ballot(k0 < re) != 0; k0 += NBX) {
                int c[NBX]; double a[NBX], xv[NBX];
#pragma unroll
                for (int q = 0; q < NBX; q++) { const bool ok = k0 + q < re; c[q] = col[ok ? k0 + q : 0]; a[q] = val[ok ? k0 + q : 0]; }
#pragma unroll
                for (int q = 0; q < NBX; q++) xv[q] = x[k0 + q < re ? c[q] : 0];
#pragma unroll
                for (int q = 0; q < NBX; q++) acc[q % T] = (k0 + q < re) ? fma(a[q], xv[q], acc[q % T]) : acc[q % T];
            }
        };
        if (longest <= 4) walk_rows(std::integral_constant<int, 4>{});
        else if (longest <= 6) walk_rows(std::integral_constant<int, 6>{});
        else walk_rows(std::integral_constant<int, NB>{});
    }
    double v = acc[0];
#pragma unroll
    for (int j = 1; j < T; j++) v += acc[j];
    if (live) y[row0 + lane] = v;
    return live ? v : 0.0;
}

template <int T>
__global__ __launch_bounds__(VB) void k_spmv_run1(int n, int LP, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                  const double *__restrict__ val, const int *__restrict__ pofs,
                                                  const int *__restrict__ pbase, const int *__restrict__ pcols,
                                                  const double *__restrict__ x, double *__restrict__ y, const int *done)
{
    extern __shared__ double wlds[];                    // [VB / 64][64 * LP]: a wavefront's values, row by row (LP odd: no bank conflicts)
    if (done && *done) return;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * (VB / 64) + wv;
    if ((long)b * 64 >= n) return;
    (void)run1_block<T>(b, wv, n, LP, rowptr, col, val, pofs, pbase, pcols, x, y, wlds);
}

// The same product carrying the dot that follows it (see k_spmv_lds1d): EIGHT wavefronts per workgroup -- there is no
// barrier in the product, so the workgroup's size is free -- each adds its rows' y_i u_i (y_i y_i) with one DPP sum, and the
// workgroup leaves ONE partial per running sum: a quarter of the partials a 256-thread grid would hand the consuming pass
// (1954 instead of 3907 on the 1M-row Laplacian, whose every block re-adds them; sixteen wavefronts were measured too:
// 17.9 us per product against 15.4 -- large workgroups fill the CUs less evenly -- and four: 15.2).
constexpr int RUN1D_WG = 512;
template <int T>
__global__ __launch_bounds__(RUN1D_WG) void k_spmv_run1d(int n, int LP, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                         const double *__restrict__ val, const int *__restrict__ pofs,
                                                         const int *__restrict__ pbase, const int *__restrict__ pcols,
                                                         const double *__restrict__ x, double *__restrict__ y, const int *done, DotPlan dp)
{
    extern __shared__ double wlds[];                    // [16][64 * LP]
    __shared__ double wsum[2][RUN1D_WG / 64];
    __shared__ int arrived;
    if (done && *done) return;
    if (threadIdx.x == 0) arrived = 0;
    __syncthreads();                                    // the only barrier, before anybody has started: costs nothing
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * (RUN1D_WG / 64) + wv;
    double a0 = 0.0, a1 = 0.0;
    if ((long)b * 64 < n) {
        const long row = (long)b * 64 + lane;
        const double uv = dp.u[row < n ? row : 0];      // requested ahead of the block's loads
        const double v = run1_block<T>(b, wv, n, LP, rowptr, col, val, pofs, pbase, pcols, x, y, wlds);
        a0 = wave_sum(v * uv);                          // v = 0 on the lanes past the end
        if (dp.yy) a1 = wave_sum(v * v);
    }
    // No barrier at the end (a workgroup would hold its slots until its slowest wavefront -- the blocks that are not runs --
    // is through): every wavefront leaves its sums in LDS and takes a ticket; whoever draws the last one adds them up in
    // wavefront order (the LDS serves a wavefront's store before its ticket, so the last ticket sees every store).
    if (lane == WSUM_LANE) {
        wsum[0][wv] = a0; wsum[1][wv] = a1;
        const int ticket = __hip_atomic_fetch_add(&arrived, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (ticket == RUN1D_WG / 64 - 1) {
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int q = 0; q < RUN1D_WG / 64; q++) { t0 += wsum[0][q]; t1 += wsum[1][q]; }
            dp.part[blockIdx.x] = t0;
            if (dp.yy) dp.part[dp.stride + blockIdx.x] = t1;
        }
    }
}



// Build the packed columns of P once (device; two short synchronisations).  Returns true when ready.
// runs_only (k_spmv_run1, short rows): only the run blocks get anything -- row 0's columns; the other blocks are walked
// from the CSR arrays -- so the copy costs a few integers per block, and small systems take it too (pk_state = 2).

static bool packed_build(const CsrPart &P, hipStream_t s, bool runs_only, int R = PK_R)
{
    if (P.pk_state != 0) return P.pk_state == (runs_only ? 2 : 1) && P.pk_R == R;
    P.pk_state = -1;
    static const int env = [] { const char *e = std::getenv("LCG_HIP_PACKED"); return e ? atoi(e) : -1; }();
    const int mode = env >= 0 ? env : P.pk_mode;
    if (mode == 0) return false;
    if (!runs_only && mode < 0 && P.nnz < (1 << 22)) return false;       // small systems are launch-bound: not worth the memory
    PlanTimer timer(P, s);
    const int n = P.n_rows;
    const int nb = (n + R - 1) / R;
    int *ngr = nullptr, *span = nullptr;
    long total = 0;
    int hspan[4] = {0, 0, 0, 0};
    static const int runs = [] { const char *e = lab_env("LCG_HIP_PACKED_RUNS"); return e ? atoi(e) : 1; }();    // 0: A/B runs without run blocks
    int dof = 1;
    if (R != PK_R && !runs_only) {      // long rows: do all rows consist of groups of 2 / 3 / 4 consecutive columns?
        static const int dof_off = [] { const char *e = lab_env("LCG_HIP_PACKED_DOF"); return e && atoi(e) == 0; }();     // (A/B runs)
        int *d = nullptr, h[3] = {1, 1, 1};
        bool okd = !dof_off && hipMalloc(&d, sizeof h) == hipSuccess && hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, s) == hipSuccess;
        if (okd) {
            hipLaunchKernelGGL(k_pk_dof, dim3((n + VB - 1) / VB), dim3(VB), 0, s, n, P.rowptr, P.col, d);
            okd = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        }
        if (d) hipFree(d);
        if (okd) dof = h[2] ? 4 : (h[1] ? 3 : (h[0] ? 2 : 1));
        else (void)hipGetLastError();
    }
    bool ok = hipMalloc(&P.pk_base, sizeof(int) * (size_t)nb) == hipSuccess && hipMalloc(&P.pk_ofs, sizeof(int) * ((size_t)nb + 1)) == hipSuccess &&
              hipMalloc(&ngr, sizeof(int) * (size_t)nb) == hipSuccess && hipMalloc(&span, 4 * sizeof(int)) == hipSuccess &&
              hipMemsetAsync(span, 0, 4 * sizeof(int), s) == hipSuccess;
    if (ok) {
        // (k_pk_meta: 0 no run blocks, 1 run blocks and template blocks, 2 run blocks only)
        static const int tpls = [] { const char *e = lab_env("LCG_HIP_PACKED_TEMPLATES"); return e ? atoi(e) : 1; }();    // 0: A/B runs without template blocks
        hipLaunchKernelGGL(k_pk_meta, dim3(nb), dim3(64), 0, s, n, P.rowptr, P.col, P.pk_base, ngr, span, (runs && R == PK_R) ? (tpls ? 1 : 2) : 0, R);
        ok = hipMemcpyAsync(hspan, span, 4 * sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    if (ok) ok = runs_only ? (runs && 2L * (hspan[2] + hspan[3]) >= nb) : hspan[0] < PK_SPAN;     // runs only: worth it when most blocks are runs or templates
    static const int force_bits = [] { const char *e = lab_env("LCG_HIP_PACKED_BITS"); return e ? atoi(e) : 0; }();   // 21: A/B runs
    const int bits = runs_only ? 0 : ((hspan[0] < (1 << 18) && force_bits != 21) ? 18 : 21);     // seven 18-bit columns per group where the blocks are narrow enough
    if (ok) {
        hipLaunchKernelGGL(k_pk_groups, dim3((nb + VB - 1) / VB), dim3(VB), 0, s, nb, runs_only ? 0 : (128 / bits) * dof, ngr);
        ok = device_exclusive_scan(nb, ngr, P.pk_ofs, s, &total) == 0;
    }
    if (ok) ok = total > 0 && total < 0x7fffffffL;
    if (ok) ok = hipMalloc(&P.pk_data, 16 * ((size_t)total + 4)) == hipSuccess;
    if (ok) {
        if (runs_only)
            hipLaunchKernelGGL(k_pk_pack<0>, dim3(nb), dim3(VB), 0, s, n, P.rowptr, P.col, P.pk_base, P.pk_ofs, static_cast<v4i *>(P.pk_data), R, 1);
        else if (bits == 18)
            hipLaunchKernelGGL(k_pk_pack<18>, dim3(nb), dim3(VB), 0, s, n, P.rowptr, P.col, P.pk_base, P.pk_ofs, static_cast<v4i *>(P.pk_data), R, dof);
        else
            hipLaunchKernelGGL(k_pk_pack<21>, dim3(nb), dim3(VB), 0, s, n, P.rowptr, P.col, P.pk_base, P.pk_ofs, static_cast<v4i *>(P.pk_data), R, dof);
        ok = hipGetLastError() == hipSuccess;
    }
    if (ngr) hipFree(ngr);
    if (span) hipFree(span);
    if (!ok) {
        (void)hipGetLastError();
        if (P.pk_base) hipFree(P.pk_base);
        if (P.pk_ofs) hipFree(P.pk_ofs);
        if (P.pk_data) hipFree(P.pk_data);
        P.pk_base = P.pk_ofs = nullptr; P.pk_data = nullptr;
        return false;
    }
    P.pk_maxrow = hspan[1];
    P.pk_runs = hspan[2];
    P.pk_tpls = hspan[3];
    P.pk_groups = total;
    P.pk_bits = bits;
    P.pk_R = R;
    P.pk_dof = dof;
    P.pk_state = runs_only ? 2 : 1;
    return true;
}
static bool packed_ready(const CsrPart &P, hipStream_t s, int R = PK_R) { return packed_build(P, s, false, R); }

// what k_spmv_ldsp multiplied with (static strings: lcg_hip_csr_last_kernel hands them out)
static const char *ldsp_name(const CsrPart &P, bool dot)
{
    const int form = (P.pk_runs > 0 ? 1 : 0) + (P.pk_tpls > 0 ? 2 : 0);
    if (dot) {
        static const char *const d[4] = {"k_spmv_ldsp (LDS-staged, packed columns) carrying the dot that follows the product",
                                         "k_spmv_ldsp (LDS-staged, run blocks + packed columns) carrying the dot that follows the product",
                                         "k_spmv_ldsp (LDS-staged, template blocks + packed columns) carrying the dot that follows the product",
                                         "k_spmv_ldsp (LDS-staged, run blocks + template blocks + packed columns) carrying the dot that follows the product"};
        return d[form];
    }
    static const char *const a[2][4] = {{"k_spmv_ldsp (LDS-staged, 18-bit packed columns)", "k_spmv_ldsp (LDS-staged, run blocks + 18-bit packed columns)",
                                         "k_spmv_ldsp (LDS-staged, template blocks + 18-bit packed columns)",
                                         "k_spmv_ldsp (LDS-staged, run blocks + template blocks + 18-bit packed columns)"},
                                        {"k_spmv_ldsp (LDS-staged, 21-bit packed columns)", "k_spmv_ldsp (LDS-staged, run blocks + 21-bit packed columns)",
                                         "k_spmv_ldsp (LDS-staged, template blocks + 21-bit packed columns)",
                                         "k_spmv_ldsp (LDS-staged, run blocks + template blocks + 21-bit packed columns)"}};
    return a[P.pk_bits == 18 ? 0 : 1][form];
}

// rows per block of the LDS-staged kernels so that R*mean_row entries fit one LDS window, and whether EVERY block's slice
// does (blocks whose slice exceeds the window take the window-by-window kernel); scanned once per (matrix, R)
template <class V>
static int lds_shape(const CsrPart &P, int variant, double mean_row, hipStream_t s, int *R_out, bool *onewin)
{
    const int n = P.n_rows;
    const double cap = LdsCfg<V>::CH - 64;
    const int R = variant < -1 ? -variant
                               : (256 * mean_row <= cap ? 256 : 128 * mean_row <= cap ? 128 : 64 * mean_row <= cap ? 64
                                  : 32 * mean_row <= cap ? 32 : 16);
    if (P.slice_R != R) {
        int *d = nullptr, h = 0;
        HIPCHK(hipMalloc(&d, sizeof(int)));
        HIPCHK(hipMemsetAsync(d, 0, sizeof(int), s));
        const int nb = (n + R - 1) / R;
        hipLaunchKernelGGL(k_max_slice, dim3((nb + VB - 1) / VB), dim3(VB), 0, s, n, R, P.rowptr, d);
        hipError_t e = hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        hipFree(d);
        if (e != hipSuccess) return fail(e, "slice scan", __FILE__, __LINE__);
        P.slice_R = R; P.max_slice = h;
    }
    *R_out = R;
    *onewin = P.padded && P.max_slice <= LdsCfg<V>::CH;
    return 0;
}

static bool long_rows_packed()
{   // LCG_HIP_PACKED_LONG=0: long rows stay with k_spmv_lds1 (A/B runs)
    static const bool on = [] { const char *e = lab_env("LCG_HIP_PACKED_LONG"); return !e || atoi(e) != 0; }();
    return on;
}

template <class V, bool ACC, bool PUSH = false>
static int spmv_dispatch(const CsrPart &P, int variant, double mean_row, const V *x, V *y, hipStream_t s,
                         const int *done, const PushPlan &pp = PushPlan())
{
    const int n = P.n_rows;
    const unsigned xb = PUSH ? (unsigned)(pp.nblocks + pp.nrecv) : 0u;       // pushing blocks in front of the grid, receiving blocks behind it
    if constexpr (sizeof(V) == 8 && !ACC && !PUSH) {
        if (n > 0 && P.lr_mode == 1)        // a range of long rows (ranges_chosen, class 3)
            return long_rows_launch(P, reinterpret_cast<const double *>(x), reinterpret_cast<double *>(y), s, done);
        if (n > 0 && (variant == 0 || variant == -1) && ranges_chosen(P, s)) {
            RangePlan *R = static_cast<RangePlan *>(P.rg_plan);
            bool changed = false;
            for (size_t i = 0; i < R->parts.size(); i++) {
                const CsrPart &Q = R->parts[i];
                int rc = spmv_dispatch<V, false, false>(Q, variant, Q.n_rows ? (double)Q.nnz / Q.n_rows : 0.0, x, y + R->r0[i], s, done);
                if (rc) return rc;
                if (Q.last_kernel != R->seen[i]) { R->seen[i] = Q.last_kernel; changed = true; }
            }
            if (changed) {
                R->desc.clear();
                for (size_t i = 0; i < R->parts.size(); i++) {
                    char buf[64];
                    std::snprintf(buf, sizeof buf, "%srows [%d, %d): ", i ? " | " : "", R->r0[i], R->r0[i] + R->parts[i].n_rows);
                    R->desc += buf; R->desc += R->parts[i].last_kernel;
                }
            }
            P.last_kernel = R->desc.c_str();
            return 0;
        }
        if (n > 0 && (variant == 0 || variant == -1) && binned_chosen(P, s)) {
            P.last_kernel = "k_bin_expand + k_bin_reduce (two-pass binned product, x and row sums in LDS)";
            return binned_launch(P, reinterpret_cast<const double *>(x), reinterpret_cast<double *>(y), s, done);
        }
        if (n > 0 && (variant == 0 || variant == -1) && tiled_chosen(P, s)) {
            P.last_kernel = "k_tile_spmv (one-pass tiled product: x tiles and row sums in LDS)";
            return tiled_launch(P, reinterpret_cast<const double *>(x), reinterpret_cast<double *>(y), s, done);
        }
    }
    if constexpr (sizeof(V) == 8 && !ACC && PUSH) {     // sharded rows, direct exchange: the tiled product carries the pushing blocks too
        if (n > 0 && (variant == 0 || variant == -1) && tiled_chosen(P, s)) {
            P.last_kernel = "k_tile_spmv (one-pass tiled product: x tiles and row sums in LDS) + pushing blocks";
            return tiled_launch(P, reinterpret_cast<const double *>(x), reinterpret_cast<double *>(y), s, done, &pp);
        }
    }
    if (n == 0) {
        if (PUSH && xb > 0) {       // nothing to multiply, but the neighbours still wait for x and the flags
            hipLaunchKernelGGL((k_spmv_wave<V, 1, ACC, PUSH>), dim3(xb), dim3(VB), 0, s, 0, P.rowptr, P.col,
                               reinterpret_cast<const V *>(P.val), x, y, done, pp);
            HIPCHK(hipGetLastError());
        }
        return 0;
    }
    const V *val = reinterpret_cast<const V *>(P.val);
    const bool al16 = (((uintptr_t)P.val | (uintptr_t)P.col) & 15) == 0;
    if (variant == 0) {
        if (!al16 || mean_row > 160.0) variant = mean_row > 48 ? 64 : (mean_row > 24 ? 32 : (mean_row > 12 ? 16 : 8));
        else variant = -1;
    }
    if (variant < 0) {
        if (!al16) return fail(hipErrorInvalidValue, "LDS-staged A.x needs 16-byte aligned col/val", __FILE__, __LINE__);
        int R = 0; bool onewin = false;
        { int rc = lds_shape<V>(P, variant, mean_row, s, &R, &onewin); if (rc) return rc; }
        if constexpr (sizeof(V) == 8 && !ACC) {
            if constexpr (!PUSH) {
                // short rows whose blocks of 64 are mostly runs: one wavefront per block, no staging (k_spmv_run1)
                static const bool run1_off = [] { const char *e = lab_env("LCG_HIP_RUN1"); return e && atoi(e) == 0; }();
                // (its wavefront-private LDS is dynamic: 4 wavefronts x 64 rows x LP doubles must stay within the 64 KB a launch
                //  may ask for without further ado -- rows of up to 30 entries)
                if (!run1_off && variant == -1 && (R == 256 || R == 128) && packed_build(P, s, true) && P.pk_maxrow <= 30) {
                    const unsigned g = (unsigned)(((n + 63) / 64 + VB / 64 - 1) / (VB / 64));
                    const int LP = P.pk_maxrow | 1;
                    const size_t lds = sizeof(double) * (VB / 64) * 64 * (size_t)LP;
                    if (R == 256)
                        hipLaunchKernelGGL((k_spmv_run1<1>), dim3(g), dim3(VB), lds, s, n, LP, P.rowptr, P.col, reinterpret_cast<const double *>(val), P.pk_ofs, P.pk_base,
                                           static_cast<const int *>(P.pk_data), reinterpret_cast<const double *>(x), reinterpret_cast<double *>(y), done);
                    else
                        hipLaunchKernelGGL((k_spmv_run1<2>), dim3(g), dim3(VB), lds, s, n, LP, P.rowptr, P.col, reinterpret_cast<const double *>(val), P.pk_ofs, P.pk_base,
                                           static_cast<const int *>(P.pk_data), reinterpret_cast<const double *>(x), reinterpret_cast<double *>(y), done);
                    HIPCHK(hipGetLastError());
                    P.last_kernel = "k_spmv_run1 (one wavefront per 64-row block, run blocks without staging)";
                    return 0;
                }
            }
            if (R == PK_R && onewin && packed_ready(P, s)) {
                // gathers per lane in the first batch: enough for the longest row when that is 9..12 per lane
                const int per_lane = (P.pk_maxrow + VB / PK_R - 1) / (VB / PK_R);
                const int ns = per_lane <= 6 ? 6 : per_lane <= 7 ? 7 : per_lane <= 8 ? 8 : per_lane <= 9 ? 9 : per_lane <= 10 ? 10 : per_lane <= 12 ? 12 : 8;
                const int win = pk_window(P.max_slice);             // the smallest LDS window every block fits: more workgroups per CU
#define PK_LAUNCH(NSS, BB, CC)                                                                                      \
        hipLaunchKernelGGL((k_spmv_ldsp<PUSH, NSS, BB, false, CC>), dim3((n + PK_R - 1) / PK_R + xb), dim3(VB), 0, s, n, P.rowptr, \
                           static_cast<const v4i *>(P.pk_data), P.pk_ofs, P.pk_base, reinterpret_cast<const double *>(val), \
                           reinterpret_cast<const double *>(x), reinterpret_cast<double *>(y), done, pp, ldsp_plain_plan())
#define PK_CASE(NSS)                                                                                                \
    case NSS:                                                                                                       \
        if (win == PK_CH_8) { if (P.pk_bits == 18) PK_LAUNCH(NSS, 18, PK_CH_8); else PK_LAUNCH(NSS, 21, PK_CH_8); } \
        else if (win == PK_CH_7) { if (P.pk_bits == 18) PK_LAUNCH(NSS, 18, PK_CH_7); else PK_LAUNCH(NSS, 21, PK_CH_7); } \
        else if (win == PK_CH_SMALL) { if (P.pk_bits == 18) PK_LAUNCH(NSS, 18, PK_CH_SMALL); else PK_LAUNCH(NSS, 21, PK_CH_SMALL); } \
        else { if (P.pk_bits == 18) PK_LAUNCH(NSS, 18, LdsCfg<double>::CH); else PK_LAUNCH(NSS, 21, LdsCfg<double>::CH); } \
        break;
                switch (ns) { PK_CASE(6) PK_CASE(7) PK_CASE(8) PK_CASE(9) PK_CASE(10) PK_CASE(12) }
#undef PK_LAUNCH
#undef PK_CASE
                HIPCHK(hipGetLastError());
                P.last_kernel = ldsp_name(P, false);
                return 0;
            }
            if constexpr (!PUSH) {
                // long rows (a block of 64 does not fit the window): blocks of 32 / 16 rows, 8 / 16 lanes per row, packed columns only.
                // ONE predicated batch of NS gathers per lane, NS the smallest of the instantiated ones that covers the longest row
                // (nine where the rows are longer still: the loop takes the rest) -- every gather beyond a lane's entries is a wasted
                // instruction, and with nine for all the packed form was no faster than the plain one.
                if ((R == 32 || R == 16) && onewin && long_rows_packed() && packed_ready(P, s, R)) {
                    const int T = VB / R;
                    const int per_lane = (P.pk_maxrow + T - 1) / T;
                    const int win = pk_window(P.max_slice);
                    DotPlan pkl_plan = ldsp_plain_plan(); pkl_plan.dof = P.pk_dof;
#define PKL_LAUNCH(NSS, BB, CC, RRR)                                                                                \
        hipLaunchKernelGGL((k_spmv_ldsp<false, NSS, BB, false, CC, RRR>), dim3((n + RRR - 1) / RRR), dim3(VB), 0, s, n, P.rowptr, \
                           static_cast<const v4i *>(P.pk_data), P.pk_ofs, P.pk_base, reinterpret_cast<const double *>(val), \
                           reinterpret_cast<const double *>(x), reinterpret_cast<double *>(y), done, PushPlan(), pkl_plan)
#define PKL_WIN(NSS, BB, RRR)                                                                                       \
        do { if (win == PK_CH_8) PKL_LAUNCH(NSS, BB, PK_CH_8, RRR); else if (win == PK_CH_7) PKL_LAUNCH(NSS, BB, PK_CH_7, RRR);   \
             else if (win == PK_CH_SMALL) PKL_LAUNCH(NSS, BB, PK_CH_SMALL, RRR); else PKL_LAUNCH(NSS, BB, LdsCfg<double>::CH, RRR); } while (0)
#define PKL_BITS(NSS, RRR) do { if (P.pk_bits == 18) PKL_WIN(NSS, 18, RRR); else PKL_WIN(NSS, 21, RRR); } while (0)
                    if (R == 32) { if (per_lane <= 7) PKL_BITS(7, 32); else PKL_BITS(9, 32); }
                    else { if (per_lane <= 6) PKL_BITS(6, 16); else PKL_BITS(9, 16); }
#undef PKL_LAUNCH
#undef PKL_WIN
#undef PKL_BITS
                    HIPCHK(hipGetLastError());
                    P.last_kernel = P.pk_dof > 1 ? (P.pk_bits == 18 ? "k_spmv_ldsp (LDS-staged, long rows: one 18-bit packed column per group of consecutive columns)"
                                                                   : "k_spmv_ldsp (LDS-staged, long rows: one 21-bit packed column per group of consecutive columns)")
                                                 : (P.pk_bits == 18 ? "k_spmv_ldsp (LDS-staged, long rows: 18-bit packed columns)" : "k_spmv_ldsp (LDS-staged, long rows: 21-bit packed columns)");
                    return 0;
                }
            }
        }
        // (real matrices, large enough for the memory system to matter: the smallest LDS window every block fits -- 27-point stencil
        //  x 3 unknowns, 16 rows of 81 entries per block: 625 us with five workgroups per CU -> see DESIGN 3.1)
        int win = 0;
        if constexpr (sizeof(V) == 8 && !ACC) { if (onewin && P.nnz >= (1 << 22)) win = pk_window(P.max_slice); }
#define LDS1_LAUNCH(RR, CC)                                                                            \
            hipLaunchKernelGGL((k_spmv_lds1<V, RR, ACC, PUSH, CC>), dim3((n + RR - 1) / RR + xb), dim3(VB), 0, s, n, \
                               (long)(P.end_abs >= 0 ? P.end_abs : P.nnz), P.rowptr, P.col, val, x, y, done, pp)
#define LDS_CASE(RR)                                                                                   \
    case RR:                                                                                           \
        if (onewin) {                                                                                  \
            if constexpr (sizeof(V) == 8 && !ACC) {                                                    \
                if (win == PK_CH_8) LDS1_LAUNCH(RR, PK_CH_8);                                          \
                else if (win == PK_CH_7) LDS1_LAUNCH(RR, PK_CH_7);                                     \
                else if (win == PK_CH_SMALL) LDS1_LAUNCH(RR, PK_CH_SMALL);                             \
                else LDS1_LAUNCH(RR, LdsCfg<V>::CH);                                                   \
            } else LDS1_LAUNCH(RR, LdsCfg<V>::CH);                                                     \
        } else                                                                                          \
            hipLaunchKernelGGL((k_spmv_ldsw<V, RR, ACC, PUSH>), dim3((n + RR - 1) / RR + xb), dim3(VB), 0, s, n, \
                               (long)(P.end_abs >= 0 ? P.end_abs : P.nnz), P.rowptr, P.col, val, x, y, done, pp); \
        break;
        switch (R) {
            LDS_CASE(256) LDS_CASE(128) LDS_CASE(64) LDS_CASE(32) LDS_CASE(16)
        default: return fail(hipErrorInvalidValue, "bad LDS A.x rows-per-block", __FILE__, __LINE__);
        }
#undef LDS_CASE
#undef LDS1_LAUNCH
        P.last_kernel = onewin ? "k_spmv_lds1 (LDS-staged CSR)" : "k_spmv_ldsw (LDS-staged CSR, windowed)";
    } else {
#define WAVE_CASE(TT)                                                                                  \
    case TT: {                                                                                         \
        const long threads = (long)n * TT;                                                             \
        hipLaunchKernelGGL((k_spmv_wave<V, TT, ACC, PUSH>), dim3((unsigned)((threads + VB - 1) / VB) + xb), dim3(VB), 0, s, \
                           n, P.rowptr, P.col, val, x, y, done, pp);                                   \
    } break;
        switch (variant) {
            WAVE_CASE(1) WAVE_CASE(2) WAVE_CASE(4) WAVE_CASE(8) WAVE_CASE(16) WAVE_CASE(32) WAVE_CASE(64)
        default: return fail(hipErrorInvalidValue, "bad A.x lanes-per-row", __FILE__, __LINE__);
        }
#undef WAVE_CASE
        P.last_kernel = "k_spmv_wave (lanes per row, shuffle reduction)";
    }
    HIPCHK(hipGetLastError());
    return 0;
}

int spmv_launch(const CsrPart &P, bool is_complex, int variant, double mean_row, const double *x, double *y,
                bool accumulate, hipStream_t s, const int *done)
{
    if (is_complex) {
        auto xv = reinterpret_cast<const double2 *>(x);
        auto yv = reinterpret_cast<double2 *>(y);
        return accumulate ? spmv_dispatch<double2, true>(P, variant, mean_row, xv, yv, s, done)
                          : spmv_dispatch<double2, false>(P, variant, mean_row, xv, yv, s, done);
    }
    return accumulate ? spmv_dispatch<double, true>(P, variant, mean_row, x, y, s, done)
                      : spmv_dispatch<double, false>(P, variant, mean_row, x, y, s, done);
}

int spmv_launch_push(const CsrPart &P, bool is_complex, int variant, double mean_row, const double *x, double *y,
                     hipStream_t s, const int *done, const PushPlan &pp)
{
    if (is_complex)
        return spmv_dispatch<double2, false, true>(P, variant, mean_row, reinterpret_cast<const double2 *>(x),
                                                   reinterpret_cast<double2 *>(y), s, done, pp);
    return spmv_dispatch<double, false, true>(P, variant, mean_row, x, y, s, done, pp);
}

// The packed kernel carrying y.u (and y.y): one partial per block of 64 rows into a buffer of the part's own, folded to <= 512 sums
// by a second small kernel into part[0 .. *slots) (y.y: part[AXP_CAP ..)) -- together they replace a pass over two vectors of the
// part's height.  pp != nullptr: the product of a row shard with its pushing blocks in front (comm.hip).  1 = launched, 0 = this
// part does not take the packed kernel (nothing was launched), < 0 failure.
// nofold != nullptr: the second stage is left to the caller (comm.hip folds the per-block sums in the kernel that finishes the
// shard's product anyway): *nofold = number of per-block sums waiting in P.dot_part, nothing is written to `part`.
static bool ensure_dot_part(const CsrPart &P, long count)
{   // (the packed kernel leaves a sum per 64 rows, the tiled one per 1024: a part that changes family needs the larger buffer)
    if (P.dot_part && P.dot_cap >= count) return true;
    if (P.dot_part) { if (ctx().inited) (void)hipDeviceSynchronize(); (void)hipFree(P.dot_part); P.dot_part = nullptr; P.dot_cap = 0; }
    if (hipMalloc(&P.dot_part, sizeof(double) * 2 * (size_t)count) != hipSuccess) { (void)hipGetLastError(); P.dot_part = nullptr; return false; }
    P.dot_cap = count;
    return true;
}

int csr_part_ax_dot(const CsrPart &P, int variant, double mean_row, const double *x, double *y, const double *u, int yy, double *part,
                    int *slots, hipStream_t s, const int *done, const PushPlan *pp, int *nofold)
{
    const int n = P.n_rows;
    if (n <= 0 || (variant != 0 && variant != -1)) return 0;
    if (mean_row > 160.0 || ((((uintptr_t)P.val | (uintptr_t)P.col) & 15) != 0)) return 0;
    if (!pp && binned_chosen(P, s)) return 0;
    if (tiled_chosen(P, s)) {
        // the tiled product: one sum per chunk of 1024 rows (per consumer wavefront), folded like the packed kernel's per-block sums
        static const bool tl_off = [] { const char *e = lab_env("LCG_HIP_AX_DOT_TILED"); return e && atoi(e) == 0; }();
        if (tl_off || !tiled_dot_ok(P)) return 0;
        const int nchunk = tiled_chunks(P);
        if (nchunk <= 0) return 0;
        if (!ensure_dot_part(P, nchunk)) return 0;
        DotPlan dp; dp.u = u; dp.part = P.dot_part; dp.yy = yy; dp.stride = nchunk;
        int rc = tiled_launch(P, x, y, s, done, pp, &dp);
        if (rc) return rc;
        P.last_kernel = pp ? "k_tile_spmv (one-pass tiled product: x tiles and row sums in LDS) + pushing blocks, carrying the dot that follows the product"
                           : "k_tile_spmv (one-pass tiled product: x tiles and row sums in LDS) carrying the dot that follows the product";
        if (nofold) { *nofold = nchunk; *slots = 0; return 1; }
        const int g2 = std::min(512, (nchunk + VB - 1) / VB);
        const int per = (nchunk + g2 - 1) / g2;
        hipLaunchKernelGGL(k_axp_fold, dim3((nchunk + per - 1) / per), dim3(VB), 0, s, P.dot_part, nchunk, nchunk, per, yy, part, done);
        HIPCHK(hipGetLastError());
        *slots = (nchunk + per - 1) / per;
        return 1;
    }
    int R = 0; bool onewin = false;
    { int rc = lds_shape<double>(P, -1, mean_row, s, &R, &onewin); if (rc) return rc; }
    static const bool big_off = [] { const char *e = lab_env("LCG_HIP_AX_DOT_PACKED"); return e && atoi(e) == 0; }();
    if (!onewin || big_off) return 0;
    // (long rows -- blocks of 32 / 16 rows with packed columns, spmv_dispatch -- keep the dot as a pass of its own: carried in the
    //  product it cost 32 us on the 27-point stencil x 3 unknowns, 187,500 blocks of 16 rows, where the separate pass costs 7)
    if (R != PK_R || !packed_ready(P, s)) return 0;
    const int nblk = (n + PK_R - 1) / PK_R;
    if (!ensure_dot_part(P, nblk)) return 0;
    DotPlan dp; dp.u = u; dp.part = P.dot_part; dp.yy = yy; dp.stride = nblk; dp.ystore = y_store_policy();
    {   static const bool ux_off = [] { const char *e = lab_env("LCG_HIP_DOT_UX"); return e && atoi(e) == 0; }();      // (A/B runs)
        dp.ux = (u == x && !ux_off) ? 1 : 0; }
    const int per_lane = (P.pk_maxrow + VB / PK_R - 1) / (VB / PK_R);
    const int ns = per_lane <= 6 ? 6 : per_lane <= 7 ? 7 : per_lane <= 8 ? 8 : per_lane <= 9 ? 9 : per_lane <= 10 ? 10 : per_lane <= 12 ? 12 : 8;
    const PushPlan ppv = pp ? *pp : PushPlan();
    const unsigned xb = pp ? (unsigned)(pp->nblocks + pp->nrecv) : 0u;
    const int win = pk_window(P.max_slice);
#define PKD_LAUNCH1(PU, NSS, BB, CC)                                                                                \
            hipLaunchKernelGGL((k_spmv_ldsp<PU, NSS, BB, true, CC>), dim3(nblk + xb), dim3(VB), 0, s, n, P.rowptr,  \
                               static_cast<const v4i *>(P.pk_data), P.pk_ofs, P.pk_base, P.val, x, y, done, ppv, dp)
#define PKD_LAUNCH(PU, NSS, BB)                                                                                     \
    do {                                                                                                            \
        if (win == PK_CH_8) PKD_LAUNCH1(PU, NSS, BB, PK_CH_8);                                                      \
        else if (win == PK_CH_7) PKD_LAUNCH1(PU, NSS, BB, PK_CH_7);                                                 \
        else if (win == PK_CH_SMALL) PKD_LAUNCH1(PU, NSS, BB, PK_CH_SMALL);                                         \
        else PKD_LAUNCH1(PU, NSS, BB, LdsCfg<double>::CH);                                                          \
    } while (0)
#define PKD_CASE(NSS)                                                                                               \
    case NSS:                                                                                                       \
        if (pp) { if (P.pk_bits == 18) PKD_LAUNCH(true, NSS, 18); else PKD_LAUNCH(true, NSS, 21); }                 \
        else { if (P.pk_bits == 18) PKD_LAUNCH(false, NSS, 18); else PKD_LAUNCH(false, NSS, 21); }                  \
        break;
    switch (ns) { PKD_CASE(6) PKD_CASE(7) PKD_CASE(8) PKD_CASE(9) PKD_CASE(10) PKD_CASE(12) }
#undef PKD_LAUNCH
#undef PKD_LAUNCH1
#undef PKD_CASE
    HIPCHK(hipGetLastError());
    P.last_kernel = ldsp_name(P, true);
    if (nofold) { *nofold = nblk; *slots = 0; return 1; }
    const int g2 = std::min(512, (nblk + VB - 1) / VB);
    const int per = (nblk + g2 - 1) / g2;
    hipLaunchKernelGGL(k_axp_fold, dim3((nblk + per - 1) / per), dim3(VB), 0, s, P.dot_part, nblk, nblk, per, yy, part, done);
    HIPCHK(hipGetLastError());
    P.last_kernel = ldsp_name(P, true);
    *slots = (nblk + per - 1) / per;
    return 1;
}

// A.x with the dot(s) that follow it in the Krylov loops carried in the product (k_spmv_lds1d / k_spmv_ldsp<DOT> / k_spmv_run1d): real
// matrices, the LDS-staged one-window family.  Everything else answers 0 and the caller multiplies and reduces in two launches as before.
int csr_ax_dot(lcg_hip_csr *A, const double *x, double *y, const double *u, int yy, double *part, int *slots, hipStream_t s,
               const int *done)
{
    static const bool off = [] { const char *e = std::getenv("LCG_HIP_AX_DOT"); return e && atoi(e) == 0; }();
    if (off || !A || A->is_complex || A->n_rows <= 0) return 0;
    if (A->distributed) return dist_ax_dot(A, x, y, u, yy, part, slots);
    const CsrPart &P = A->main;
    const int n = P.n_rows;
    if (A->variant != 0 && A->variant != -1) return 0;
    if (ranges_chosen(P, s)) return 0;  // multiplied range by range: the dot keeps its own pass
    if (A->mean_row > 160.0 || ((((uintptr_t)P.val | (uintptr_t)P.col) & 15) != 0)) return 0;
    if (binned_chosen(P, s)) return 0;
    if (tiled_chosen(P, s)) return csr_part_ax_dot(P, A->variant, A->mean_row, x, y, u, yy, part, slots, s, done, nullptr, nullptr);
    int R = 0; bool onewin = false;
    { int rc = lds_shape<double>(P, -1, A->mean_row, s, &R, &onewin); if (rc) return rc; }
    if (!onewin) return 0;
    if (R == PK_R && packed_ready(P, s)) return csr_part_ax_dot(P, A->variant, A->mean_row, x, y, u, yy, part, slots, s, done, nullptr, nullptr);
    if ((R == 32 || R == 16) && long_rows_packed() && packed_ready(P, s, R)) return 0;     // (the product alone: see csr_part_ax_dot)
    const int nblk = (n + R - 1) / R;
    static const bool run1_off = [] { const char *e = lab_env("LCG_HIP_RUN1"); return e && atoi(e) == 0; }();
    if (!run1_off && (R == 256 || R == 128) && packed_build(P, s, true) && P.pk_maxrow <= 15) {
        // short-row stencils (k_spmv_run1d): eight wavefronts, one partial per workgroup of 512 rows (8 x 64 x LP doubles of
        // dynamic LDS: rows of up to 15 entries; longer ones take the plain product and the separate pass)
        const int nwg = (int)((((long)n + 63) / 64 + RUN1D_WG / 64 - 1) / (RUN1D_WG / 64));
        if (nwg <= AXP_CAP) {
            DotPlan dp; dp.u = u; dp.part = part; dp.yy = yy;
            const int LP = P.pk_maxrow | 1;
            const size_t lds = sizeof(double) * (RUN1D_WG / 64) * 64 * (size_t)LP;
            if (R == 256)
                hipLaunchKernelGGL((k_spmv_run1d<1>), dim3(nwg), dim3(RUN1D_WG), lds, s, n, LP, P.rowptr, P.col, P.val, P.pk_ofs, P.pk_base,
                                   static_cast<const int *>(P.pk_data), x, y, done, dp);
            else
                hipLaunchKernelGGL((k_spmv_run1d<2>), dim3(nwg), dim3(RUN1D_WG), lds, s, n, LP, P.rowptr, P.col, P.val, P.pk_ofs, P.pk_base,
                                   static_cast<const int *>(P.pk_data), x, y, done, dp);
            HIPCHK(hipGetLastError());
            P.last_kernel = "k_spmv_run1d (one wavefront per 64-row block, run blocks without staging) carrying the dot that follows the product";
            *slots = nwg;
            return 1;
        }
    }
    // Where it pays (measured, scripts/ax_dot_lab.py + scripts/ab_small.py): systems whose iteration is a chain of kernel
    // latencies -- the product grows by ~0.6 us, a ~3 us pass and its launch go.  At 1M rows (3907 row blocks) the product grew by
    // 3.2 us and every block of the consuming pass re-added 3907 partials: 39.2 vs 38.5 us per PCG iteration, so from
    // LCG_HIP_AX_DOT_MAXBLK (default 2048) row blocks on the separate pass stays.
    static const int maxblk = [] { const char *e = lab_env("LCG_HIP_AX_DOT_MAXBLK"); const int v = e ? atoi(e) : 2048; return v < 1 ? 1 : (v > AXP_CAP ? AXP_CAP : v); }();
    if (nblk > maxblk) return 0;
    DotPlan dp; dp.u = u; dp.part = part; dp.yy = yy;
    const int g = nblk;
#define LDSD_CASE(RR) case RR: hipLaunchKernelGGL((k_spmv_lds1d<RR>), dim3(g), dim3(VB), 0, s, n, P.rowptr, P.col, P.val, x, y, done, dp); break;
    switch (R) {
        LDSD_CASE(256) LDSD_CASE(128) LDSD_CASE(64) LDSD_CASE(32) LDSD_CASE(16)
    default: return 0;
    }
#undef LDSD_CASE
    HIPCHK(hipGetLastError());
    P.last_kernel = "k_spmv_lds1d (LDS-staged CSR carrying the dot that follows the product)";
    *slots = g;
    return 1;
}

void free_part(CsrPart &P)
{
    long_rows_free(P);
    ranges_free(P);
    binned_free(P);
    tiled_free(P);
    if (P.owned) { hipFree(P.rowptr); hipFree(P.col); hipFree(P.val); }
    if (P.pk_base) hipFree(P.pk_base);
    if (P.pk_ofs) hipFree(P.pk_ofs);
    if (P.pk_data) hipFree(P.pk_data);
    if (P.dot_part) hipFree(P.dot_part);
    P = CsrPart();
}

} // namespace lcgh

using namespace lcgh;

namespace lcgh { void dist_free(lcg_hip_csr *A); }

extern "C" {

int lcg_hip_csr_create(lcg_hip_csr_t *out, int n_rows, int n_cols, int64_t nnz, const int *rowptr, const int *col,
                       const double *val, int is_complex, int mem, int adopt)
{
    if (!out || n_rows <= 0 || nnz < 0 || nnz > 0x7fffffffLL || !rowptr || !col || !val) return LCG_HIP_E_ARG;
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    lcg_hip_csr *A = new lcg_hip_csr();
    A->n_rows = n_rows; A->n_cols = n_cols; A->is_complex = is_complex != 0;
    A->mean_row = (double)nnz / n_rows;
    if (mem == LCG_HIP_MEM_DEVICE && adopt) {
        A->main.n_rows = n_rows; A->main.nnz = nnz; A->main.owned = false; A->main.padded = adopt == 2;
        A->main.n_cols = n_cols;
        A->main.rowptr = const_cast<int *>(rowptr); A->main.col = const_cast<int *>(col); A->main.val = const_cast<double *>(val);
    } else {
        rc = alloc_part(A->main, n_rows, nnz, A->is_complex);
        if (rc) { delete A; return rc; }
        A->main.n_cols = n_cols;
        const hipMemcpyKind kind = mem == LCG_HIP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        hipError_t e = hipMemcpyAsync(A->main.rowptr, rowptr, sizeof(int) * ((size_t)n_rows + 1), kind, c.stream);
        if (e == hipSuccess && nnz) e = hipMemcpyAsync(A->main.col, col, sizeof(int) * (size_t)nnz, kind, c.stream);
        if (e == hipSuccess && nnz) e = hipMemcpyAsync(A->main.val, val, sizeof(double) * (is_complex ? 2 : 1) * (size_t)nnz, kind, c.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
        if (e != hipSuccess) { free_part(A->main); delete A; return fail(e, "csr upload", __FILE__, __LINE__); }
    }
    *out = A;
    return 0;
}

int lcg_hip_csr_destroy(lcg_hip_csr_t A)
{
    if (!A) return 0;
    ctx().forget_places();       // (driver.hpp: Placement remembers timings by the value array's address)
    ctx().released_bytes += (size_t)A->main.nnz * (A->is_complex ? 20 : 12);     // (the plans beside it not counted: the walk's "fresh allocator" rule needs the order of magnitude)
    dist_free(A);
    free_part(A->main);
    for (int i = 1; i < 4; i++) free_part(A->op[i]);
    if (A->invdiag) hipFree(A->invdiag);
    delete A;
    return 0;
}

int lcg_hip_csr_rows(lcg_hip_csr_t A) { return A ? A->n_rows : 0; }
int64_t lcg_hip_csr_nnz(lcg_hip_csr_t A) { return A ? A->main.nnz : 0; }
int lcg_hip_csr_arrays(lcg_hip_csr_t A, const int **rowptr, const int **col, const double **val)
{
    if (!A) return LCG_HIP_E_ARG;
    if (rowptr) *rowptr = A->main.rowptr;
    if (col) *col = A->main.col;
    if (val) *val = A->main.val;
    return 0;
}
int lcg_hip_csr_set_kernel(lcg_hip_csr_t A, int variant)
{
    if (!A) return LCG_HIP_E_ARG;
    A->variant = variant;
    return 0;
}

int lcg_hip_csr_set_packed(lcg_hip_csr_t A, int mode)
{
    if (!A || mode < -1 || mode > 1) return LCG_HIP_E_ARG;
    for (CsrPart *P : {&A->main, &A->loc}) {
        P->pk_mode = mode;
        ranges_free(*P);                    // the ranges inherit the modes: cut again at the next product
        if (P->pk_state > 0 && mode == 0) {         // give the memory back
            if (ctx().inited) (void)hipDeviceSynchronize();
            hipFree(P->pk_base); hipFree(P->pk_ofs); hipFree(P->pk_data);
            P->pk_base = P->pk_ofs = nullptr; P->pk_data = nullptr;
        }
        if (P->pk_state < 0 || mode == 0) P->pk_state = 0;     // decide again at the next product
    }
    return 0;
}

int lcg_hip_csr_set_binned(lcg_hip_csr_t A, int mode)
{
    if (!A || mode < -1 || mode > 1) return LCG_HIP_E_ARG;
    for (CsrPart *P : {&A->main, &A->loc}) {
        P->bn_mode = mode;
        ranges_free(*P);                    // the ranges inherit the modes: cut again at the next product
        if (mode == 0) binned_free(*P);     // gives the plan's memory back
        else P->bn_state = 0;               // decide again at the next product (a plan that exists is kept and reused)
    }
    return 0;
}

int lcg_hip_csr_set_tiled(lcg_hip_csr_t A, int mode)
{
    if (!A || mode < -1 || mode > 1) return LCG_HIP_E_ARG;
    for (CsrPart *P : {&A->main, &A->loc}) {
        P->tl_mode = mode;
        ranges_free(*P);                    // the ranges inherit the modes: cut again at the next product
        if (mode == 0) tiled_free(*P);
        else P->tl_state = 0;               // decide again at the next product (a plan that exists is kept and reused)
    }
    return 0;
}

int lcg_hip_csr_set_ranges(lcg_hip_csr_t A, int mode)
{
    if (!A || mode < -1 || mode > 1) return LCG_HIP_E_ARG;
    if (ctx().inited) (void)hipDeviceSynchronize();
    for (CsrPart *P : {&A->main, &A->loc}) { ranges_free(*P); P->rg_mode = mode; }
    return 0;
}

int lcg_hip_csr_ranges(lcg_hip_csr_t A, int cap, int *first_row)
{
    if (!A) return 0;
    const CsrPart &P = A->distributed ? A->loc : A->main;
    const RangePlan *R = static_cast<const RangePlan *>(P.rg_plan);
    if (!R || P.rg_state <= 0) return 0;
    for (int i = 0; i < cap && i < (int)R->r0.size() && first_row; i++) first_row[i] = R->r0[i];
    return (int)R->r0.size();
}

const char *lcg_hip_csr_tiled_status(lcg_hip_csr_t A)
{
    if (!A) return "";
    return A->distributed ? A->loc.tl_why : A->main.tl_why;
}

const char *lcg_hip_csr_last_kernel(lcg_hip_csr_t A)
{
    if (!A) return "";
    return A->distributed ? A->loc.last_kernel : A->main.last_kernel;
}

const char *lcg_hip_csr_binned_status(lcg_hip_csr_t A)
{
    if (!A) return "";
    return A->distributed ? A->loc.bn_why : A->main.bn_why;
}

int64_t lcg_hip_csr_packed_runs(lcg_hip_csr_t A, int64_t *blocks_out)
{
    if (!A) return 0;
    const CsrPart &P = A->distributed ? A->loc : A->main;
    if (blocks_out) *blocks_out = (P.n_rows + P.pk_R - 1) / P.pk_R;
    return P.pk_state > 0 ? P.pk_runs : 0;
}

static int64_t part_traffic_model(const CsrPart &P)
{
    const char *k = P.last_kernel;
    if (!k || !*k) return 0;
    if (P.rg_state > 0 && P.rg_plan) {      // range by range; x is counted once
        const RangePlan *R = static_cast<const RangePlan *>(P.rg_plan);
        int64_t b = 0;
        const int64_t xbytes = 8 * (P.n_cols > 0 ? P.n_cols : (int64_t)P.n_rows);
        for (const CsrPart &Q : R->parts) b += part_traffic_model(Q) - xbytes;
        return b + xbytes;
    }
    const int64_t n = P.n_rows, ncols = P.n_cols > 0 ? P.n_cols : P.n_rows;
    const int64_t vectors = 4 * (n + 1) + 8 * ncols + 8 * n;       // row pointers, x once, y
    const int64_t nb = (n + P.pk_R - 1) / P.pk_R;
    if (std::strncmp(k, "k_lr_", 5) == 0) return 12 * P.nnz + vectors;       // CSR as it is, the chunk lists are small
    if (std::strncmp(k, "k_bin_", 6) == 0) return binned_traffic_bytes(P);
    if (std::strncmp(k, "k_tile", 6) == 0) return tiled_traffic_bytes(P);
    if (std::strncmp(k, "k_spmv_ldsp", 11) == 0)       // values + packed columns (run blocks: row 0's columns only) + two words per block
        return 8 * P.nnz + 16 * (int64_t)P.pk_groups + 8 * nb + vectors;
    if (std::strncmp(k, "k_spmv_run1", 11) == 0) {     // values + row 0's columns of the run blocks + the CSR columns of the other blocks
        const double other = nb > 0 ? 1.0 - (double)(P.pk_runs + P.pk_tpls) / (double)nb : 1.0;
        return 8 * P.nnz + (int64_t)(4.0 * other * (double)P.nnz) + 16 * (int64_t)P.pk_groups + 8 * nb + vectors;
    }
    return 12 * P.nnz + vectors;                        // the CSR arrays as they are
}

int64_t lcg_hip_csr_packed_templates(lcg_hip_csr_t A)
{
    if (!A) return 0;
    const CsrPart &P = A->distributed ? A->loc : A->main;
    return P.pk_state > 0 ? P.pk_tpls : 0;
}

int64_t lcg_hip_csr_last_traffic_model(lcg_hip_csr_t A)
{
    if (!A || A->is_complex) return 0;
    return part_traffic_model(A->distributed ? A->loc : A->main);
}

int lcg_hip_csr_plan_info(lcg_hip_csr_t A, double *build_ms, int64_t *extra_bytes)
{
    if (!A) return LCG_HIP_E_ARG;
    const CsrPart &P = A->distributed ? A->loc : A->main;
    double ms = P.plan_ms;
    int64_t b = 0;
    auto add = [&](const CsrPart &Q) {
        if (Q.pk_state > 0) b += 16 * ((int64_t)Q.pk_groups + 4) + 8 * (((int64_t)Q.n_rows + Q.pk_R - 1) / Q.pk_R + 1);
        if (Q.bn_plan) b += (int64_t)binned_plan_bytes(Q);
        if (Q.tl_plan) b += (int64_t)tiled_plan_bytes(Q);
    };
    add(P);
    if (P.rg_state > 0 && P.rg_plan)
        for (const CsrPart &Q : static_cast<const RangePlan *>(P.rg_plan)->parts) { add(Q); ms += Q.plan_ms; }
    if (build_ms) *build_ms = ms;
    if (extra_bytes) *extra_bytes = b;
    return 0;
}

// ---- callbacks -------------------------------------------------------------------------------
// The callback types return void (lcg.h:37-38, clcg.h:40-41): a failure is parked in Ctx::ax_rc, where the solver
// loop picks it up right after the call (driver.hpp: timed_ax / checked_mx) and ends the solve with that code.
static inline void park(int rc) { if (rc && !ctx().ax_rc) ctx().ax_rc = rc; }

void lcg_hip_csr_ax(void *instance, const double *x, double *y, const int n)
{
    lcg_hip_csr *A = static_cast<lcg_hip_csr *>(instance);
    (void)n;
    park(lcg_hip_spmv(A, x, y));
}

void clcg_hip_csr_ax(void *instance, const double *x, double *y, const int n, int layout, int conjugate)
{
    (void)n;
    park(lcg_hip_spmv_op(static_cast<lcg_hip_csr *>(instance), x, y, layout, conjugate));
}

void lcg_hip_jacobi_mx(void *instance, const double *x, double *z, const int n)
{
    park(jacobi_launch(static_cast<lcg_hip_csr *>(instance), x, z, n, ctx().stream));
}

int lcg_hip_spmv_op(lcg_hip_csr_t A, const double *x, double *y, int layout, int conjugate)
{
    if (!A || !x || !y) return LCG_HIP_E_ARG;
    if (!layout && (!conjugate || !A->is_complex)) return lcg_hip_spmv(A, x, y);
    Ctx &c = ctx();
    const CsrPart *P = nullptr;
    int rc = op_part(A, layout, conjugate, &P);
    if (rc) return rc;
    if (A->distributed) return dist_spmv_op(A, *P, x, y);
    return spmv_launch(*P, A->is_complex, A->variant, A->mean_row, x, y, false, c.stream,
                       ax_flag(c));
}


int lcg_hip_spmv(lcg_hip_csr_t A, const double *x, double *y)
{
    if (!A || !x || !y) return LCG_HIP_E_ARG;
    Ctx &c = ctx();
    if (A->distributed) return dist_spmv(A, x, y);
    return spmv_launch(A->main, A->is_complex, A->variant, A->mean_row, x, y, false, c.stream,
                       ax_flag(c));
}

} // extern "C"
