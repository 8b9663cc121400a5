// spmv_lab.hip -- A/B laboratory for the CSR A.x kernel (not part of the product).
// Variants run interleaved in ONE process on the same matrix (cdna guide rule 24); prints median
// and min time per variant and the max deviation from the reference variant.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude scripts/spmv_lab.hip -Lliblcg_amd/lib -llcg_hip \
//         -Wl,-rpath,'$ORIGIN/../../liblcg_amd/lib' -o scripts/bin/spmv_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "lcg_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int VB = 256;
constexpr int CH = 2304;

// ---------------------------------------------------------------- V0: product kernel shape (register staging)
template <int R, bool GATHER, bool XCD>
__global__ __launch_bounds__(VB) void k_base(int n, long nnz, const int *__restrict__ rowptr, const int *__restrict__ col,
                                             const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    constexpr int T = VB / R;
    __shared__ double sval[CH];
    __shared__ int scol[CH];
    __shared__ double sred[T > 1 ? T : 1][R];
    const int tid = threadIdx.x;
    int lb = blockIdx.x;
    if (XCD) { const int per = gridDim.x >> 3; lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3); }
    const int row0 = lb * R;
    if (row0 >= n) return;
    const int nrows = min(R, n - row0);
    const int rl = tid % R, j0 = tid / R;
    const int s = rowptr[row0], e = rowptr[row0 + nrows];
    int rs = 0, re = 0;
    if (rl < nrows) { rs = rowptr[row0 + rl]; re = rowptr[row0 + rl + 1]; }
    double acc = 0.0;
    for (int base = s & ~3; base < e; base += CH) {
        const int cnt = min(CH, e - base);
        for (int u = tid * 4; u < cnt; u += VB * 4) {
            const long g = (long)base + u;
            if (g + 3 < nnz) {
                *reinterpret_cast<int4 *>(scol + u) = *reinterpret_cast<const int4 *>(col + g);
                *reinterpret_cast<double2 *>(sval + u) = *reinterpret_cast<const double2 *>(val + g);
                *reinterpret_cast<double2 *>(sval + u + 2) = *reinterpret_cast<const double2 *>(val + g + 2);
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
            const double a0 = sval[k - base], a1 = sval[k + T - base];
            acc = fma(a0, GATHER ? x[c0] : x[row0 + rl], acc);
            acc = fma(a1, GATHER ? x[c1] : x[row0 + rl], acc);
        }
        if (k < hi) acc = fma(sval[k - base], GATHER ? x[scol[k - base]] : x[row0 + rl], acc);
        __syncthreads();
    }
    if (T > 1) {
        sred[j0][rl] = acc;
        __syncthreads();
        if (j0 == 0 && rl < nrows) {
            double v = sred[0][rl];
#pragma unroll
            for (int j = 1; j < T; j++) v += sred[j][rl];
            y[row0 + rl] = v;
        }
    } else if (rl < nrows) y[row0 + rl] = acc;
}

// ---------------------------------------------------------------- V1: pure stream of val/col (ceiling)
__global__ __launch_bounds__(VB) void k_stream(long nnz, const int *__restrict__ col, const double *__restrict__ val, double *__restrict__ y)
{
    double acc = 0.0;
    const long stride = (long)gridDim.x * VB * 4;
    for (long g = ((long)blockIdx.x * VB + threadIdx.x) * 4; g + 3 < nnz; g += stride) {
        const int4 c = *reinterpret_cast<const int4 *>(col + g);
        const double2 v0 = *reinterpret_cast<const double2 *>(val + g);
        const double2 v1 = *reinterpret_cast<const double2 *>(val + g + 2);
        acc += v0.x * c.x + v0.y * c.y + v1.x * c.z + v1.y * c.w;
    }
    if (acc == 1.2345e-300) y[0] = acc;
}

// ---------------------------------------------------------------- V2: LDS-DMA staging (global_load_lds 16 B/lane)
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

template <int R, bool XCD, int NBUF>
__global__ __launch_bounds__(VB) void k_dma(int n, long nnz, const int *__restrict__ rowptr, const int *__restrict__ col,
                                            const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
                                            int rows_per_block_total)
{
    // one block walks `rows_per_block_total / R` consecutive chunks of R rows, double buffered:
    // the DMA of chunk c+1 is in flight while chunk c is multiplied.
    constexpr int T = VB / R;
    __shared__ __attribute__((aligned(16))) double sval[NBUF][CH];
    __shared__ __attribute__((aligned(16))) int scol[NBUF][CH];
    __shared__ double sred[T > 1 ? T : 1][R];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int lb = blockIdx.x;
    if (XCD) { const int per = gridDim.x >> 3; lb = (blockIdx.x & 7) * per + (blockIdx.x >> 3); }
    const int chunks = rows_per_block_total / R;
    const int first_row = lb * rows_per_block_total;
    if (first_row >= n) return;
    const int rl = tid % R, j0 = tid / R;

    auto issue = [&](int c, int buf) -> int {
        int ops = 0;
        const int row0 = first_row + c * R;
        if (row0 >= n) return 0;
        const int nrows = min(R, n - row0);
        const int s = rowptr[row0], e = rowptr[row0 + nrows];
        const int base = s & ~3;
        const int cnt = min(CH, e - base);
        // each wave-instruction moves 1 KiB: 256 ints / 128 doubles
        for (int u = wave * 256; u < cnt; u += 4 * 256) {       // col: 4 ints per lane
            const long g = (long)base + u + lane * 4;
            const long gc = g + 3 < nnz ? g : (nnz - 4 > 0 ? ((nnz - 4) & ~3L) : 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(col + gc), (lds_void *)(&scol[buf][u]), 16, 0, 0);
            ops++;
        }
        for (int u = wave * 128; u < cnt; u += 4 * 128) {       // val: 2 doubles per lane
            const long g = (long)base + u + lane * 2;
            const long gc = g + 1 < nnz ? g : (nnz - 2 > 0 ? ((nnz - 2) & ~1L) : 0);
            __builtin_amdgcn_global_load_lds((glb_void *)(val + gc), (lds_void *)(&sval[buf][u]), 16, 0, 0);
            ops++;
        }
        return ops;
    };

    issue(0, 0);
    for (int c = 0; c < chunks; c++) {
        const int buf = NBUF > 1 ? (c & 1) : 0;
        const int row0 = first_row + c * R;
        if (row0 >= n) break;
        int younger = 0;
        if (NBUF > 1 && c + 1 < chunks) younger = issue(c + 1, buf ^ 1);
        // wait for chunk c; exactly `younger` DMA ops of chunk c+1 (wave-uniform, <= 8) stay in flight
        switch (__builtin_amdgcn_readfirstlane(younger)) {
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        __syncthreads();
        const int nrows = min(R, n - row0);
        const int s = rowptr[row0];
        const int base = s & ~3;
        int rs = 0, re = 0;
        if (rl < nrows) { rs = rowptr[row0 + rl]; re = rowptr[row0 + rl + 1]; }
        const int hi = min(re, base + CH);
        double acc = 0.0;
        int k = rs + j0;
        for (; k + T < hi; k += 2 * T) {
            const int c0 = scol[buf][k - base], c1 = scol[buf][k + T - base];
            const double a0 = sval[buf][k - base], a1 = sval[buf][k + T - base];
            acc = fma(a0, x[c0], acc);
            acc = fma(a1, x[c1], acc);
        }
        if (k < hi) acc = fma(sval[buf][k - base], x[scol[buf][k - base]], acc);
        if (T > 1) {
            sred[j0][rl] = acc;
            __syncthreads();
            if (j0 == 0 && rl < nrows) {
                double v = sred[0][rl];
#pragma unroll
                for (int j = 1; j < T; j++) v += sred[j][rl];
                y[row0 + rl] = v;
            }
        } else if (rl < nrows) y[row0 + rl] = acc;
        __syncthreads();    // buffer `buf` is free for chunk c+2
    }
}

// ---------------------------------------------------------------- V3: register staging, tunable window / nt loads / gather depth
typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <int R, int CHN, bool NT, int UNR>
__global__ __launch_bounds__(VB) void k_v2(int n, long nnz, const int *__restrict__ rowptr, const int *__restrict__ col,
                                           const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y)
{
    constexpr int T = VB / R;
    __shared__ __attribute__((aligned(16))) double sval[CHN];
    __shared__ __attribute__((aligned(16))) int scol[CHN];
    __shared__ double sred[T > 1 ? T : 1][R];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * R;
    if (row0 >= n) return;
    const int nrows = min(R, n - row0);
    const int rl = tid % R, j0 = tid / R;
    const int s = rowptr[row0], e = rowptr[row0 + nrows];
    int rs = 0, re = 0;
    if (rl < nrows) { rs = rowptr[row0 + rl]; re = rowptr[row0 + rl + 1]; }
    double acc = 0.0;
    for (int base = s & ~3; base < e; base += CHN) {
        const int cnt = min(CHN, e - base);
        for (int u = tid * 4; u < cnt; u += VB * 4) {
            const long g = (long)base + u;
            if (g + 3 < nnz) {
                const v4i *pc = reinterpret_cast<const v4i *>(col + g);
                const v2d *pv = reinterpret_cast<const v2d *>(val + g);
                v4i c4; v2d v0, v1;
                if (NT) { c4 = __builtin_nontemporal_load(pc); v0 = __builtin_nontemporal_load(pv); v1 = __builtin_nontemporal_load(pv + 1); }
                else { c4 = *pc; v0 = *pv; v1 = pv[1]; }
                *reinterpret_cast<v4i *>(scol + u) = c4;
                *reinterpret_cast<v2d *>(sval + u) = v0;
                *reinterpret_cast<v2d *>(sval + u + 2) = v1;
            } else {
                for (int q = 0; q < 4 && g + q < nnz; q++) { scol[u + q] = col[g + q]; sval[u + q] = val[g + q]; }
            }
        }
        __syncthreads();
        const int lo = max(rs, base), hi = min(re, base + cnt);
        int k = rs + j0;
        if (k < lo) k += ((lo - k + T - 1) / T) * T;
        for (; k + (UNR - 1) * T < hi; k += UNR * T) {
            int c[UNR]; double a[UNR], xv[UNR];
#pragma unroll
            for (int q = 0; q < UNR; q++) { c[q] = scol[k + q * T - base]; a[q] = sval[k + q * T - base]; }
#pragma unroll
            for (int q = 0; q < UNR; q++) xv[q] = x[c[q]];
#pragma unroll
            for (int q = 0; q < UNR; q++) acc = fma(a[q], xv[q], acc);
        }
        for (; k < hi; k += T) acc = fma(sval[k - base], x[scol[k - base]], acc);
        __syncthreads();
    }
    if (T > 1) {
        sred[j0][rl] = acc;
        __syncthreads();
        if (j0 == 0 && rl < nrows) {
            double v = sred[0][rl];
#pragma unroll
            for (int j = 1; j < T; j++) v += sred[j][rl];
            y[row0 + rl] = v;
        }
    } else if (rl < nrows) y[row0 + rl] = acc;
}

// ---------------------------------------------------------------- V4: register-prefetched chunks, sred aliased on the staging buffer
// A block walks `chunks` consecutive groups of R rows.  The global loads of group c+1 are issued into
// registers BEFORE group c is multiplied, so HBM latency overlaps the gather phase inside the block
// (on top of the overlap between resident blocks); LDS stays single-buffered.
template <int R, int CHN, int UNR, int LB = 1>
__global__ __launch_bounds__(VB, LB) void k_v3(int n, long nnz, const int *__restrict__ rowptr, const int *__restrict__ col,
                                           const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
                                           int chunks)
{
    constexpr int T = VB / R;
    constexpr int NRND = (CHN + VB * 4 - 1) / (VB * 4);
    __shared__ __attribute__((aligned(16))) double sval[CHN];
    __shared__ __attribute__((aligned(16))) int scol[CHN];
    double (*sred)[R] = reinterpret_cast<double (*)[R]>(sval);      // T*R doubles <= CHN
    const int tid = threadIdx.x;
    const int rl = tid % R, j0 = tid / R;
    const int first = blockIdx.x * chunks * R;
    if (first >= n) return;

    v4i pc[NRND]; v2d pv0[NRND], pv1[NRND];
    int nbase = 0, ncnt = 0;
    auto prefetch = [&](int row0) {
        const int nrows = min(R, n - row0);
        const int s = rowptr[row0], e = rowptr[row0 + nrows];
        nbase = s & ~3; ncnt = min(CHN, e - nbase);
#pragma unroll
        for (int r = 0; r < NRND; r++) {
            const int u = tid * 4 + r * VB * 4;
            const long g = (long)nbase + u;
            if (u < ncnt) {
                if (g + 3 < nnz) {
                    pc[r] = *reinterpret_cast<const v4i *>(col + g);
                    pv0[r] = *reinterpret_cast<const v2d *>(val + g);
                    pv1[r] = *reinterpret_cast<const v2d *>(val + g + 2);
                } else {
                    v4i c = {0, 0, 0, 0}; v2d a = {0, 0}, b = {0, 0};
                    if (g < nnz) { c.x = col[g]; a.x = val[g]; }
                    if (g + 1 < nnz) { c.y = col[g + 1]; a.y = val[g + 1]; }
                    if (g + 2 < nnz) { c.z = col[g + 2]; b.x = val[g + 2]; }
                    pc[r] = c; pv0[r] = a; pv1[r] = b;
                }
            }
        }
    };
    prefetch(first);
    for (int c = 0; c < chunks; c++) {
        const int row0 = first + c * R;
        if (row0 >= n) break;
        const int nrows = min(R, n - row0);
        const int base = nbase, cnt = ncnt;
        // commit the prefetched registers to LDS
#pragma unroll
        for (int r = 0; r < NRND; r++) {
            const int u = tid * 4 + r * VB * 4;
            if (u < cnt) {
                *reinterpret_cast<v4i *>(scol + u) = pc[r];
                *reinterpret_cast<v2d *>(sval + u) = pv0[r];
                *reinterpret_cast<v2d *>(sval + u + 2) = pv1[r];
            }
        }
        int rs = 0, re = 0;
        if (rl < nrows) { rs = rowptr[row0 + rl]; re = rowptr[row0 + rl + 1]; }
        __syncthreads();
        if (c + 1 < chunks && row0 + R < n) prefetch(row0 + R);      // next group's HBM loads fly during the gathers
        const int hi = min(re, base + cnt);
        double acc = 0.0;
        int k = rs + j0;
        for (; k + (UNR - 1) * T < hi; k += UNR * T) {
            int cc[UNR]; double a[UNR], xv[UNR];
#pragma unroll
            for (int q = 0; q < UNR; q++) { cc[q] = scol[k + q * T - base]; a[q] = sval[k + q * T - base]; }
#pragma unroll
            for (int q = 0; q < UNR; q++) xv[q] = x[cc[q]];
#pragma unroll
            for (int q = 0; q < UNR; q++) acc = fma(a[q], xv[q], acc);
        }
        for (; k < hi; k += T) acc = fma(sval[k - base], x[scol[k - base]], acc);
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
        } else if (rl < nrows) y[row0 + rl] = acc;
    }
}

struct Variant { std::string name; std::function<void()> run; std::vector<double> ms; };

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 10000000;
    const long band = argc > 2 ? atol(argv[2]) : 131072;
    const int rounds = argc > 3 ? atoi(argv[3]) : 15;
    lcg_hip_csr_t A;
    if (lcg_hip_init(0) || lcg_hip_csr_generate(&A, n, 16, band, 1, 1, 0.01, 0, n)) { printf("gen failed: %s\n", lcg_hip_last_error()); return 1; }
    const int *rowptr, *col; const double *val;
    lcg_hip_csr_arrays(A, &rowptr, &col, &val);
    const long nnz = lcg_hip_csr_nnz(A);
    double *x, *y, *yref;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&yref, n * 8));
    lcg_hip_gen_xtrue(n, 1, 0, n, x);
    lcg_hip_synchronize();
    hipStream_t s; CK(hipStreamCreate(&s));
    const double bytes = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    printf("n=%ld band=%ld nnz=%ld algorithmic bytes %.3f GB\n", n, band, nnz, bytes / 1e9);

    std::vector<Variant> vs;
    auto grid8 = [&](int R) { return (unsigned)((((n + R - 1) / R + 7) / 8) * 8); };
    vs.push_back({"base R64", [&] { hipLaunchKernelGGL((k_base<64, true, false>), dim3((n + 63) / 64), dim3(VB), 0, s, (int)n, nnz, rowptr, col, val, x, y); }, {}});
    vs.push_back({"base R64 xcd", [&] { hipLaunchKernelGGL((k_base<64, true, true>), dim3(grid8(64)), dim3(VB), 0, s, (int)n, nnz, rowptr, col, val, x, y); }, {}});
    vs.push_back({"base R64 nogather", [&] { hipLaunchKernelGGL((k_base<64, false, false>), dim3((n + 63) / 64), dim3(VB), 0, s, (int)n, nnz, rowptr, col, val, x, y); }, {}});
    vs.push_back({"stream val/col only", [&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(VB), 0, s, nnz, col, val, y); }, {}});
    vs.push_back({"stream val/col 8192 blocks", [&] { hipLaunchKernelGGL(k_stream, dim3(8192), dim3(VB), 0, s, nnz, col, val, y); }, {}});
    vs.push_back({"dma R64 1buf 1chunk", [&] { hipLaunchKernelGGL((k_dma<64, false, 1>), dim3((n + 63) / 64), dim3(VB), 0, s, (int)n, nnz, rowptr, col, val, x, y, 64); }, {}});
    vs.push_back({"dma R64 1buf 1chunk xcd", [&] { hipLaunchKernelGGL((k_dma<64, true, 1>), dim3(grid8(64)), dim3(VB), 0, s, (int)n, nnz, rowptr, col, val, x, y, 64); }, {}});
#define V2(NAME, R, CHN, NT, UNR) \
    vs.push_back({NAME, [&] { hipLaunchKernelGGL((k_v2<R, CHN, NT, UNR>), dim3((n + R - 1) / R), dim3(VB), 0, s, (int)n, nnz, rowptr, col, val, x, y); }, {}});
    V2("v2 R64 ch2304 unr2", 64, 2304, false, 2)
    V2("v2 R64 ch2304 unr4", 64, 2304, false, 4)
    V2("v2 R64 ch2304 unr8", 64, 2304, false, 8)
    V2("v2 R64 ch2304 nt unr2", 64, 2304, true, 2)
    V2("v2 R64 ch2304 nt unr4", 64, 2304, true, 4)
    V2("v2 R32 ch1152 unr4", 32, 1152, false, 4)
    V2("v2 R32 ch1152 nt unr4", 32, 1152, true, 4)
    V2("v2 R128 ch4352 unr4", 128, 4352, false, 4)
    V2("v2 R128 ch4352 nt unr4", 128, 4352, true, 4)
    V2("v2 R64 ch2176 nt unr4", 64, 2176, true, 4)
#define V3(NAME, R, CHN, UNR, CHUNKS) \
    vs.push_back({NAME, [&] { hipLaunchKernelGGL((k_v3<R, CHN, UNR>), dim3((n + (R * CHUNKS) - 1) / (R * CHUNKS)), dim3(VB), 0, s, (int)n, nnz, rowptr, col, val, x, y, CHUNKS); }, {}});
    V3("v3 R64 ch2240 unr8 1chunk", 64, 2240, 8, 1)
    V3("v3 R64 ch2240 unr8 2chunks", 64, 2240, 8, 2)
    V3("v3 R64 ch2240 unr8 4chunks", 64, 2240, 8, 4)
    V3("v3 R64 ch2240 unr8 8chunks", 64, 2240, 8, 8)
    V3("v3 R64 ch2240 unr4 4chunks", 64, 2240, 4, 4)
    V3("v3 R64 ch2304 unr8 4chunks", 64, 2304, 8, 4)
    V3("v3 R64 ch2240 unr8 32chunks", 64, 2240, 8, 32)
#define V3L(NAME, R, CHN, UNR, LB) \
    vs.push_back({NAME, [&] { hipLaunchKernelGGL((k_v3<R, CHN, UNR, LB>), dim3((n + R - 1) / R), dim3(VB), 0, s, (int)n, nnz, rowptr, col, val, x, y, 1); }, {}});
    V3L("v3 R64 ch2240 unr8 lb6", 64, 2240, 8, 6)
    V3L("v3 R64 ch2240 unr4 lb6", 64, 2240, 4, 6)
    V3L("v3 R64 ch2240 unr4 lb1", 64, 2240, 4, 1)
    V3L("v3 R64 ch2240 unr12 lb1", 64, 2240, 12, 1)
    V3L("v3 R32 ch1120 unr4 lb8", 32, 1120, 4, 8)
    V3L("v3 R32 ch1120 unr8 lb8", 32, 1120, 8, 8)
    V3L("v3 R32 ch1120 unr4 lb1", 32, 1120, 4, 1)
    V3L("v3 R128 ch4352 unr8 lb1", 128, 4352, 8, 1)
    // reference result
    vs[0].run(); CK(hipStreamSynchronize(s));
    CK(hipMemcpy(yref, y, n * 8, hipMemcpyDeviceToDevice));
    std::vector<double> href(n), hy(n);
    CK(hipMemcpy(href.data(), yref, n * 8, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> dev(vs.size(), 0.0);
    for (int r = 0; r < rounds + 1; r++) {
        for (size_t i = 0; i < vs.size(); i++) {
            if (r == 0) CK(hipMemsetAsync(y, 0, n * 8, s));
            CK(hipEventRecord(e0, s));
            vs[i].run();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) vs[i].ms.push_back(ms);
            else if (vs[i].name.find("stream") == std::string::npos && vs[i].name.find("nogather") == std::string::npos) {
                CK(hipMemcpy(hy.data(), y, n * 8, hipMemcpyDeviceToHost));
                double d = 0; for (long k = 0; k < n; k++) d = std::max(d, std::fabs(hy[k] - href[k]));
                dev[i] = d;
            }
        }
    }
    for (size_t i = 0; i < vs.size(); i++) {
        auto &m = vs[i].ms; std::sort(m.begin(), m.end());
        printf("%-32s median %.3f ms  min %.3f ms  -> %.0f GB/s (median)  maxdev %.1e\n", vs[i].name.c_str(), m[m.size() / 2], m[0],
               bytes / (m[m.size() / 2] * 1e-3) / 1e9, dev[i]);
    }
    return 0;
}
