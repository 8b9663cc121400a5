// spmv_pack_lab.hip -- laboratory (not part of the product): does compressing the column indices
// pay?  A block of R rows stores its columns relative to the block's smallest column, six 21-bit
// fields per 16 bytes (2.67 B instead of 4 B per entry: 10.67 instead of 12 B of stream per entry).
// The shipped kernel (through lcg_hip_spmv) and the packed prototype run interleaved on the same
// matrix in one process.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude scripts/spmv_pack_lab.hip -Lliblcg_amd/lib -llcg_hip \
//         -Wl,-rpath,'$ORIGIN/../../liblcg_amd/lib' -o scripts/bin/spmv_pack_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lcg_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int VB = 256;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

// per block of R rows: smallest column, column span, number of 6-entry groups
__global__ __launch_bounds__(64) void k_meta(int n, int R, const int *rowptr, const int *col, int *base, int *ngroups, int *maxspan,
                                             int *maxcnt)
{
    const int b = blockIdx.x;
    const long row0 = (long)b * R;
    const int r1 = (int)min((long)n, row0 + R);
    const int s = rowptr[row0], e = rowptr[r1];
    int lo = 0x7fffffff, hi = 0;
    for (int k = s + threadIdx.x; k < e; k += 64) { const int c = col[k]; lo = min(lo, c); hi = max(hi, c); }
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); }
    if (threadIdx.x == 0) {
        if (e == s) { lo = 0; hi = 0; }
        base[b] = lo; ngroups[b] = (e - s + 5) / 6;
        atomicMax(maxspan, hi - lo); atomicMax(maxcnt, e - s);
    }
}

__global__ __launch_bounds__(VB) void k_pack(int n, int R, const int *rowptr, const int *col, const int *base, const int *pofs,
                                             v4i *packed)
{
    const int b = blockIdx.x;
    const long row0 = (long)b * R;
    const int r1 = (int)min((long)n, row0 + R);
    const int s = rowptr[row0], e = rowptr[r1];
    const int ng = (e - s + 5) / 6, bs = base[b];
    for (int g = threadIdx.x; g < ng; g += VB) {
        u64 w[2] = {0, 0};
        for (int j = 0; j < 6; j++) {
            const int k = s + 6 * g + j;
            const u64 c = k < e ? (u64)(col[k] - bs) : 0;
            w[j / 3] |= c << (21 * (j % 3));
        }
        v4i o; o.x = (int)(unsigned)w[0]; o.y = (int)(unsigned)(w[0] >> 32); o.z = (int)(unsigned)w[1]; o.w = (int)(unsigned)(w[1] >> 32);
        packed[pofs[b] + g] = o;
    }
}

template <int R, int UNR, int PRIO = 0>
__global__ __launch_bounds__(VB) void k_spmv_p(int n, const int *__restrict__ rowptr, const v4i *__restrict__ packed,
                                               const int *__restrict__ pofs, const int *__restrict__ pbase,
                                               const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    constexpr int T = VB / R;
    constexpr int CH = 2240;                            // entries per block at most
    constexpr int NG = (CH + 5) / 6;                    // 374 groups
    constexpr int GR = (NG + VB - 1) / VB;              // 2 rounds of 16-byte group loads
    constexpr int VR = (CH / 2 + 1 + VB - 1) / VB;      // 5 rounds of 16-byte val loads
    __shared__ __attribute__((aligned(16))) double sval[CH + 2];
    __shared__ __attribute__((aligned(16))) int scol[NG * 6];
    double(*sred)[R] = reinterpret_cast<double(*)[R]>(sval);

    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * R;
    const int nrows = min(R, n - row0);
    const int rl = tid % R, j0 = tid / R;
    const int s = rowptr[row0], e = rowptr[row0 + nrows];
    const int cnt = e - s, ng = (cnt + 5) / 6;
    const int po = pofs[blockIdx.x], bs = pbase[blockIdx.x];
    const int bv = s & ~1, cntv = e - bv;

    if (PRIO == 1) __builtin_amdgcn_s_setprio(3);     // issue this block's stream loads ahead of the gathering waves
    v4i pg[GR]; v2d pv[VR];
#pragma unroll
    for (int r = 0; r < GR; r++) {
        const int gi = tid + r * VB;
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
        const int gi = tid + r * VB;
        if (gi < ng) {
            const u64 lo = (u64)(unsigned)pg[r].x | ((u64)(unsigned)pg[r].y << 32);
            const u64 hi = (u64)(unsigned)pg[r].z | ((u64)(unsigned)pg[r].w << 32);
            v2i a, b, c;
            a.x = bs + (int)(lo & 0x1fffff); a.y = bs + (int)((lo >> 21) & 0x1fffff);
            b.x = bs + (int)((lo >> 42) & 0x1fffff); b.y = bs + (int)(hi & 0x1fffff);
            c.x = bs + (int)((hi >> 21) & 0x1fffff); c.y = bs + (int)((hi >> 42) & 0x1fffff);
            v2i *dst = reinterpret_cast<v2i *>(scol + 6 * gi);
            dst[0] = a; dst[1] = b; dst[2] = c;
        }
    }
#pragma unroll
    for (int r = 0; r < VR; r++) {
        const int u = 2 * (tid + r * VB);
        if (u < cntv) *reinterpret_cast<v2d *>(sval + u) = pv[r];
    }
    if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (PRIO == 2) __builtin_amdgcn_s_setprio(3);     // or: let the gathering waves finish first
    int rs = 0, re = 0;
    if (rl < nrows) { rs = rowptr[row0 + rl]; re = rowptr[row0 + rl + 1]; }
    __syncthreads();
    double acc = 0.0;
    int k = rs + j0;
    if (UNR > 8) {      // one predicated batch of UNR slots: no serial tail for rows of up to UNR*T entries
        int c[UNR]; double a[UNR], xv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; q++) {
            const int kk = k + q * T;
            const bool ok = kk < re;
            c[q] = scol[ok ? kk - s : 0];
            a[q] = ok ? sval[kk - bv] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < UNR; q++) xv[q] = x[c[q]];
#pragma unroll
        for (int q = 0; q < UNR; q++) acc = fma(a[q], xv[q], acc);
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
    if (T > 1) {
        sred[j0][rl] = acc;
        __syncthreads();
        if (j0 == 0 && rl < nrows) {
            double v = sred[0][rl];
#pragma unroll
            for (int j = 1; j < T; j++) v += sred[j][rl];
            y[row0 + rl] = v;
        }
    } else if (rl < nrows) {
        y[row0 + rl] = acc;
    }
}

// ---- software-pipelined tiles: a block walks CHUNKS consecutive tiles of R rows.  For the staged
// tile it issues ALL its x gathers first and the stream loads of the NEXT tile behind them: vmcnt
// retires in order, so waiting for the gathers (vmcnt = number of prefetch loads) leaves the
// stream of the next tile in flight during the multiply, the row sums and the y store.
template <int R, int UNR>
__global__ __launch_bounds__(VB) void k_pipe(int n, int CHUNKS, const int *__restrict__ rowptr, const int *__restrict__ col,
                                             const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    constexpr int T = VB / R;
    constexpr int CH = 2240;
    constexpr int NRND = (CH + VB * 4 - 1) / (VB * 4);
    __shared__ __attribute__((aligned(16))) double sval[CH];
    __shared__ __attribute__((aligned(16))) int scol[CH];
    double(*sred)[R] = reinterpret_cast<double(*)[R]>(sval);

    const int tid = threadIdx.x;
    const int rl = tid % R, j0 = tid / R;
    const int ntiles = (n + R - 1) / R;
    const int t0 = blockIdx.x * CHUNKS;
    const int t1 = min(ntiles, t0 + CHUNKS);
    if (t0 >= ntiles) return;

    v4i pc[NRND]; v2d pv[NRND * 2];
    // prologue: tile t0 into registers
    int s_cur = rowptr[t0 * R];
    int s_nxt = rowptr[min(n, (t0 + 1) * R)];
    {
        const int base = s_cur & ~3, cnt = s_nxt - base;
#pragma unroll
        for (int r = 0; r < NRND; r++) {
            const int u = tid * 4 + r * VB * 4;
            const long g = (long)base + (u < cnt ? u : 0);
            pc[r] = *reinterpret_cast<const v4i *>(col + g);
            pv[2 * r] = reinterpret_cast<const v2d *>(val + g)[0];
            pv[2 * r + 1] = reinterpret_cast<const v2d *>(val + g)[1];
        }
    }
    int rs, re;
    rs = rowptr[min(n, t0 * R + rl)]; re = rowptr[min(n, t0 * R + rl + 1)];

    for (int t = t0; t < t1; t++) {
        const int row0 = t * R;
        const int nrows = min(R, n - row0);
        const int base = s_cur & ~3, cnt = s_nxt - base;
        // registers -> LDS
#pragma unroll
        for (int r = 0; r < NRND; r++) {
            const int u = tid * 4 + r * VB * 4;
            if (u < cnt) {
                *reinterpret_cast<v4i *>(scol + u) = pc[r];
                reinterpret_cast<v2d *>(sval + u)[0] = pv[2 * r];
                reinterpret_cast<v2d *>(sval + u)[1] = pv[2 * r + 1];
            }
        }
        // bounds of the next tile (uniform loads) while the stores drain
        const bool more = t + 1 < t1;
        const int s_n2 = more ? rowptr[min(n, (t + 2) * R)] : s_nxt;
        __syncthreads();
        // 1. gathers of this tile (coefficients are re-read from LDS after the wait: LDS has its own counter)
        int c[UNR]; double xv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; q++) {
            const int k = rs + j0 + q * T;
            c[q] = scol[k < re ? k - base : 0];
        }
        // row bounds of the next tile: older than the gathers, so the wait for the gathers covers them
        const int rsn = rowptr[min(n, row0 + R + rl)], ren = rowptr[min(n, row0 + R + rl + 1)];
#pragma unroll
        for (int q = 0; q < UNR; q++) xv[q] = x[c[q]];
        // 2. stream loads of the next tile, queued BEHIND the gathers
        const int nbase = more ? (s_nxt & ~3) : base;
        const int ncnt = more ? s_n2 - nbase : 0;
#pragma unroll
        for (int r = 0; r < NRND; r++) {
            const int u = tid * 4 + r * VB * 4;
            const long g = (long)nbase + (u < ncnt ? u : 0);
            pc[r] = *reinterpret_cast<const v4i *>(col + g);
            pv[2 * r] = reinterpret_cast<const v2d *>(val + g)[0];
            pv[2 * r + 1] = reinterpret_cast<const v2d *>(val + g)[1];
        }
        __builtin_amdgcn_sched_barrier(0);
        // 3. multiply (waits for the gathers only)
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < UNR; q++) {
            const int k = rs + j0 + q * T;
            const double a = k < re ? sval[k - base] : 0.0;
            acc = fma(a, xv[q], acc);
        }
        for (int k = rs + j0 + UNR * T; k < re; k += T) acc = fma(sval[k - base], x[scol[k - base]], acc);   // rows longer than UNR*T
        __syncthreads();
        if (T > 1) {
            sred[j0][rl] = acc;
            __syncthreads();
            if (j0 == 0 && rl < nrows) {
                double v = sred[0][rl];
#pragma unroll
                for (int j = 1; j < T; j++) v += sred[j][rl];
                y[row0 + rl] = v;
            }
            __syncthreads();
        } else if (rl < nrows) {
            y[row0 + rl] = acc;
        }
        s_cur = s_nxt; s_nxt = s_n2; rs = rsn; re = ren;
    }
}


// ---- no LDS at all: T lanes per row, each lane takes 4 consecutive entries per round with ONE
// 16-byte column load and two 16-byte value loads straight from HBM at 4-byte alignment (the
// hardware allows it), so that occupancy is not bound by a staging buffer.
struct __attribute__((packed, aligned(4))) u4i { int v[4]; };
struct __attribute__((packed, aligned(8))) u2d { double v[2]; };
template <int T>
__global__ __launch_bounds__(VB) void k_direct(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                               const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    const int row = (int)(((long)blockIdx.x * VB + threadIdx.x) / T);
    const int j = threadIdx.x % T;
    if (row >= n) return;
    const int rs = rowptr[row], re = rowptr[row + 1];
    double acc = 0.0;
    for (int k = rs + 4 * j; k < re; k += 4 * T) {
        const u4i c = *reinterpret_cast<const u4i *>(col + k);
        const u2d v0 = *reinterpret_cast<const u2d *>(val + k);
        const u2d v1 = *reinterpret_cast<const u2d *>(val + k + 2);
        const int c0 = c.v[0];
        const int c1 = k + 1 < re ? c.v[1] : c0, c2 = k + 2 < re ? c.v[2] : c0, c3 = k + 3 < re ? c.v[3] : c0;
        const double a1 = k + 1 < re ? v0.v[1] : 0.0, a2 = k + 2 < re ? v1.v[0] : 0.0, a3 = k + 3 < re ? v1.v[1] : 0.0;
        const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
        acc = fma(v0.v[0], x0, acc); acc = fma(a1, x1, acc); acc = fma(a2, x2, acc); acc = fma(a3, x3, acc);
    }
#pragma unroll
    for (int off = T / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, T);
    if (j == 0) y[row] = acc;
}

// ---- 18-bit fields, seven per 16 bytes (2.29 B per entry) -- for blocks spanning < 2^18 columns
__device__ __forceinline__ int f18(u64 lo, u64 hi, int j)
{
    const int sh = 18 * j;
    u64 v;
    if (sh + 18 <= 64) v = lo >> sh;
    else if (sh >= 64) v = hi >> (sh - 64);
    else v = (lo >> sh) | (hi << (64 - sh));
    return (int)(v & 0x3ffff);
}
__global__ __launch_bounds__(VB) void k_pack18(int n, int R, const int *rowptr, const int *col, const int *base, const int *pofs, v4i *packed)
{
    const int b = blockIdx.x;
    const long row0 = (long)b * R;
    const int r1 = (int)min((long)n, row0 + R);
    const int s = rowptr[row0], e = rowptr[r1];
    const int ng = (e - s + 6) / 7, bs = base[b];
    for (int g = threadIdx.x; g < ng; g += VB) {
        u64 lo = 0, hi = 0;
        for (int j = 0; j < 7; j++) {
            const int k = s + 7 * g + j;
            const u64 c = k < e ? (u64)(col[k] - bs) : 0;
            const int sh = 18 * j;
            if (sh < 64) { lo |= c << sh; if (sh + 18 > 64) hi |= c >> (64 - sh); }
            else hi |= c << (sh - 64);
        }
        v4i o; o.x = (int)(unsigned)lo; o.y = (int)(unsigned)(lo >> 32); o.z = (int)(unsigned)hi; o.w = (int)(unsigned)(hi >> 32);
        packed[pofs[b] + g] = o;
    }
}
template <int R, int UNR>
__global__ __launch_bounds__(VB) void k_spmv_p18(int n, const int *__restrict__ rowptr, const v4i *__restrict__ packed,
                                                 const int *__restrict__ pofs, const int *__restrict__ pbase,
                                                 const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    constexpr int T = VB / R;
    constexpr int CH = 2240;
    constexpr int NG = (CH + 6) / 7;                    // 320 groups
    constexpr int GR = (NG + VB - 1) / VB;
    constexpr int VR = (CH / 2 + 1 + VB - 1) / VB;
    __shared__ __attribute__((aligned(16))) double sval[CH + 2];
    __shared__ __attribute__((aligned(16))) int scol[NG * 7 + 1];
    double(*sred)[R] = reinterpret_cast<double(*)[R]>(sval);
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * R;
    const int nrows = min(R, n - row0);
    const int rl = tid % R, j0 = tid / R;
    const int s = rowptr[row0], e = rowptr[row0 + nrows];
    const int cnt = e - s, ng = (cnt + 6) / 7;
    const int po = pofs[blockIdx.x], bs = pbase[blockIdx.x];
    const int bv = s & ~1, cntv = e - bv;
    v4i pg[GR]; v2d pv[VR];
#pragma unroll
    for (int r = 0; r < GR; r++) { const int gi = tid + r * VB; pg[r] = packed[po + (gi < ng ? gi : 0)]; }
#pragma unroll
    for (int r = 0; r < VR; r++) { const int u = 2 * (tid + r * VB); pv[r] = *reinterpret_cast<const v2d *>(val + (long)bv + (u < cntv ? u : 0)); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < GR; r++) {
        const int gi = tid + r * VB;
        if (gi < ng) {
            const u64 lo = (u64)(unsigned)pg[r].x | ((u64)(unsigned)pg[r].y << 32);
            const u64 hi = (u64)(unsigned)pg[r].z | ((u64)(unsigned)pg[r].w << 32);
#pragma unroll
            for (int j = 0; j < 7; j++) scol[7 * gi + j] = bs + f18(lo, hi, j);
        }
    }
#pragma unroll
    for (int r = 0; r < VR; r++) { const int u = 2 * (tid + r * VB); if (u < cntv) *reinterpret_cast<v2d *>(sval + u) = pv[r]; }
    int rs = 0, re = 0;
    if (rl < nrows) { rs = rowptr[row0 + rl]; re = rowptr[row0 + rl + 1]; }
    __syncthreads();
    double acc = 0.0;
    int k = rs + j0;
    {
        int c[UNR]; double a[UNR], xv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; q++) { const int kk = k + q * T; const bool ok = kk < re; c[q] = scol[ok ? kk - s : 0]; a[q] = sval[ok ? kk - bv : 0]; }
#pragma unroll
        for (int q = 0; q < UNR; q++) xv[q] = x[c[q]];
#pragma unroll
        for (int q = 0; q < UNR; q++) acc = (k + q * T < re) ? fma(a[q], xv[q], acc) : acc;
        k += UNR * T;
    }
    for (; k < re; k += T) acc = fma(sval[k - bv], x[scol[k - s]], acc);
    __syncthreads();
    sred[j0][rl] = acc;
    __syncthreads();
    if (j0 == 0 && rl < nrows) {
        double v = sred[0][rl];
#pragma unroll
        for (int j = 1; j < T; j++) v += sred[j][rl];
        y[row0 + rl] = v;
    }
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 10000000;
    const long band = argc > 2 ? atol(argv[2]) : 131072;
    const int rounds = argc > 3 ? atoi(argv[3]) : 15;
    constexpr int R = 64;
    lcg_hip_csr_t A;
    if (lcg_hip_init(0) || lcg_hip_csr_generate(&A, n, 16, band, 1, 1, 0.01, 0, n)) { printf("gen failed: %s\n", lcg_hip_last_error()); return 1; }
    const int *rowptr, *col; const double *val;
    lcg_hip_csr_arrays(A, &rowptr, &col, &val);
    const long nnz = lcg_hip_csr_nnz(A);
    hipStream_t s = (hipStream_t)lcg_hip_get_stream();
    double *x, *y, *yref;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&yref, n * 8));
    lcg_hip_gen_xtrue(n, 1, 0, n, x);
    lcg_hip_synchronize();

    const int nb = (int)((n + R - 1) / R);
    int *base, *ngr, *pofs, *stat;
    CK(hipMalloc(&base, 4L * nb)); CK(hipMalloc(&ngr, 4L * nb)); CK(hipMalloc(&pofs, 4L * nb)); CK(hipMalloc(&stat, 8));
    CK(hipMemsetAsync(stat, 0, 8, s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_meta, dim3(nb), dim3(64), 0, s, (int)n, R, rowptr, col, base, ngr, stat, stat + 1);
    std::vector<int> hng(nb), hpo(nb);
    int hstat[2];
    CK(hipMemcpyAsync(hng.data(), ngr, 4L * nb, hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(hstat, stat, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    long tot = 0;
    for (int b = 0; b < nb; b++) { hpo[b] = (int)tot; tot += hng[b]; }
    printf("n=%ld band=%ld nnz=%ld blocks=%d max column span %d (limit %d) max slice %d groups %ld (%.3f B/entry)\n", n, band, nnz, nb,
           hstat[0], 1 << 21, hstat[1], tot, 16.0 * tot / nnz);
    if (hstat[0] >= (1 << 21) || hstat[1] > 2240) { printf("not eligible\n"); return 0; }
    v4i *packed; CK(hipMalloc(&packed, 16L * tot));
    CK(hipMemcpyAsync(pofs, hpo.data(), 4L * nb, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_pack, dim3(nb), dim3(VB), 0, s, (int)n, R, rowptr, col, base, pofs, packed);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float bms; CK(hipEventElapsedTime(&bms, e0, e1));
    printf("pack build: %.1f ms\n", bms);

    // 18-bit build (same base; groups of 7)
    v4i *packed18 = nullptr; int *pofs18 = nullptr;
    const bool ok18 = hstat[0] < (1 << 18);
    if (ok18) {
        std::vector<int> hpo18(nb); long tot18 = 0;
        std::vector<int> rp_h(n + 1);
        CK(hipMemcpy(rp_h.data(), rowptr, 4L * (n + 1), hipMemcpyDeviceToHost));
        for (int b = 0; b < nb; b++) { hpo18[b] = (int)tot18; const long r1 = std::min<long>(n, (long)(b + 1) * R); tot18 += (rp_h[r1] - rp_h[(long)b * R] + 6) / 7; }
        CK(hipMalloc(&packed18, 16L * (tot18 + 4))); CK(hipMalloc(&pofs18, 4L * nb));
        CK(hipMemcpyAsync(pofs18, hpo18.data(), 4L * nb, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pack18, dim3(nb), dim3(VB), 0, s, (int)n, R, rowptr, col, base, pofs18, packed18);
        CK(hipStreamSynchronize(s));
        printf("18-bit groups %ld (%.3f B/entry)\n", tot18, 16.0 * tot18 / nnz);
    } else printf("span too wide for 18 bits\n");
    const double bytes = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    struct Var { const char *name; int id; std::vector<double> ms; double dev; };
    std::vector<Var> vs = {{"shipped (lcg_hip_spmv)", 0, {}, 0}, {"packed unr4", 1, {}, 0}, {"packed unr8", 2, {}, 0}, {"packed pred9", 3, {}, 0}, {"packed18 pred9", 5, {}, 0}, {"packed pred9 prio-load", 6, {}, 0}, {"packed pred9 prio-gather", 7, {}, 0},
                           };
    auto run = [&](int id) {
        if (id >= 50000) {
            const int T = id - 50000;
            const int g = (int)(((long)n * T + VB - 1) / VB);
            if (T == 2) hipLaunchKernelGGL((k_direct<2>), dim3(g), dim3(VB), 0, s, (int)n, rowptr, col, val, x, y);
            else if (T == 4) hipLaunchKernelGGL((k_direct<4>), dim3(g), dim3(VB), 0, s, (int)n, rowptr, col, val, x, y);
            else if (T == 8) hipLaunchKernelGGL((k_direct<8>), dim3(g), dim3(VB), 0, s, (int)n, rowptr, col, val, x, y);
            else hipLaunchKernelGGL((k_direct<16>), dim3(g), dim3(VB), 0, s, (int)n, rowptr, col, val, x, y);
            return;
        }
        if (id >= 100) {
            const int unr = id / 1000, ch = id % 1000;
            const int g = (nb + ch - 1) / ch;
            if (unr == 8) hipLaunchKernelGGL((k_pipe<R, 8>), dim3(g), dim3(VB), 0, s, (int)n, ch, rowptr, col, val, x, y);
            else if (unr == 9) hipLaunchKernelGGL((k_pipe<R, 9>), dim3(g), dim3(VB), 0, s, (int)n, ch, rowptr, col, val, x, y);
            else hipLaunchKernelGGL((k_pipe<R, 10>), dim3(g), dim3(VB), 0, s, (int)n, ch, rowptr, col, val, x, y);
            return;
        }
        if (id == 0) lcg_hip_spmv(A, x, y);
        else if (id == 1) hipLaunchKernelGGL((k_spmv_p<R, 4>), dim3(nb), dim3(VB), 0, s, (int)n, rowptr, packed, pofs, base, val, x, y);
        else if (id == 2) hipLaunchKernelGGL((k_spmv_p<R, 8>), dim3(nb), dim3(VB), 0, s, (int)n, rowptr, packed, pofs, base, val, x, y);
        else if (id == 5) { if (ok18) hipLaunchKernelGGL((k_spmv_p18<R, 9>), dim3(nb), dim3(VB), 0, s, (int)n, rowptr, packed18, pofs18, base, val, x, y); else lcg_hip_spmv(A, x, y); }
        else if (id == 7) hipLaunchKernelGGL((k_spmv_p<R, 9, 2>), dim3(nb), dim3(VB), 0, s, (int)n, rowptr, packed, pofs, base, val, x, y);
        else if (id == 6) hipLaunchKernelGGL((k_spmv_p<R, 9, 1>), dim3(nb), dim3(VB), 0, s, (int)n, rowptr, packed, pofs, base, val, x, y);
        else if (id == 3) hipLaunchKernelGGL((k_spmv_p<R, 9>), dim3(nb), dim3(VB), 0, s, (int)n, rowptr, packed, pofs, base, val, x, y);
        else hipLaunchKernelGGL((k_spmv_p<R, 10>), dim3(nb), dim3(VB), 0, s, (int)n, rowptr, packed, pofs, base, val, x, y);
    };
    run(0); CK(hipStreamSynchronize(s));
    std::vector<double> href(n), hy(n);
    CK(hipMemcpy(href.data(), y, n * 8, hipMemcpyDeviceToHost));
    for (int r = 0; r < rounds + 1; r++) {
        for (auto &v : vs) {
            if (r == 0) CK(hipMemsetAsync(y, 0, n * 8, s));
            CK(hipEventRecord(e0, s));
            run(v.id);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) v.ms.push_back(ms);
            else {
                CK(hipMemcpy(hy.data(), y, n * 8, hipMemcpyDeviceToHost));
                double d = 0; for (long k = 0; k < n; k++) d = std::max(d, std::fabs(hy[k] - href[k]));
                v.dev = d;
            }
        }
    }
    for (auto &v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-26s median %.3f ms  min %.3f ms  -> %.0f GB/s algorithmic (median)  maxdev %.1e\n", v.name, v.ms[v.ms.size() / 2], v.ms[0],
               bytes / (v.ms[v.ms.size() / 2] * 1e-3) / 1e9, v.dev);
    }
    return 0;
}
