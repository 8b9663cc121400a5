// csr_build.hip -- how a CSR handle comes to be and what is derived from it once: COO -> CSR ingest on the device, the reciprocal
// diagonal (Jacobi), op(A) = A^T / A^H / conj(A) as a CSR of its own, the device scan they share, and the on-device generators of
// the benchmark systems (BASELINE configs 2-5; CPU twin: oracle/csr_oracle.c).  The A.x kernels and the choice among them: csr.hip.
#include <algorithm>
#include <cstring>
#include <functional>
#include <numeric>

#include "devcommon.hpp"

namespace lcgh {

// ------------------------------------------------------------------------- Jacobi / diagonal
// algebra_cuda.cu:40-57: scan the row for col == row (here with the row's GLOBAL index).
template <class V>
__global__ void k_diag(int n, long row0, const int *rowptr, const int *col, const V *val, V *diag, V *inv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V d = vzero(V());
    for (int k = rowptr[i]; k < rowptr[i + 1]; k++)
        if (col[k] == row0 + i) { d = val[k]; break; }
    if (diag) diag[i] = d;
    if (inv) {
        if constexpr (sizeof(V) == 8) inv[i] = 1.0 / d;
        else inv[i] = cdiv(make_double2(1.0, 0.0), d);
    }
}

struct OpMul {      // c = a .* b      (lcg_vecMvecD_element_wise, algebra_cuda.cu:59-67)
    static constexpr int NR = 0, SKIP = SKIP_NEVER;
    DevState *st; const double *a, *b; double *c;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *) { st_(c, i, vmul(ld<T>(a, i), ld<T>(b, i))); }
};
struct OpDiv {      // c = a ./ b      (lcg_vecDvecD_element_wise, algebra_cuda.cu:69-77)
    static constexpr int NR = 0, SKIP = SKIP_NEVER;
    DevState *st; const double *a, *b; double *c;
    __device__ void prep() {}
    template <class T> __device__ void apply(long i, double *);
};
template <> __device__ void OpDiv::apply<double>(long i, double *) { c[i] = a[i] / b[i]; }
template <> __device__ void OpDiv::apply<double2>(long i, double *)
{
    const double2 x = ld<double2>(a, i), y = ld<double2>(b, i);
    st_(c, i, make_double2(x.x / y.x, x.y / y.y));
}
__global__ void k_cmul(long n, const double2 *a, const double2 *b, double2 *c)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        c[i] = cmul(a[i], b[i]);
}
__global__ void k_cdiv(long n, const double2 *a, const double2 *b, double2 *c)
{   // vecDvecZ_element_wise_device, lcg_complex_cuda.cu:95-103 (cuCdiv)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        c[i] = cdiv(a[i], b[i]);
}

template <class Op> static int launch_vec(Op op, long n, uintptr_t align_or, hipStream_t s, double *partials)
{
    const bool v2 = (align_or & 15) == 0;
    const int g = grid_for(v2 ? (n + 1) / 2 : n);
    if (v2) hipLaunchKernelGGL((k_vec<Op, true>), dim3(g), dim3(VB), 0, s, op, n, partials);
    else hipLaunchKernelGGL((k_vec<Op, false>), dim3(g), dim3(VB), 0, s, op, n, partials);
    HIPCHK(hipGetLastError());
    return g;
}

int jacobi_launch(const lcg_hip_csr *A, const double *x, double *z, int n, hipStream_t s)
{
    if (!A->invdiag) return fail(hipErrorInvalidValue, "lcg_hip_csr_build_jacobi() was not called", __FILE__, __LINE__);
    if (A->is_complex) {
        hipLaunchKernelGGL(k_cmul, dim3(grid_for(n)), dim3(VB), 0, s, (long)n,
                           reinterpret_cast<const double2 *>(A->invdiag), reinterpret_cast<const double2 *>(x),
                           reinterpret_cast<double2 *>(z));
        HIPCHK(hipGetLastError());
        return 0;
    }
    int g = launch_vec(OpMul{nullptr, A->invdiag, x, z}, n, (uintptr_t)A->invdiag | (uintptr_t)x | (uintptr_t)z, s, nullptr);
    return g < 0 ? g : 0;
}

// ------------------------------------------------------------------------------ device scan
// exclusive scan of int counts -> rowptr[n+1] (three passes, 4096 items per block)
constexpr int SCAN_ITEMS = 16;
__global__ __launch_bounds__(VB) void k_scan_local(int n, const int *in, int *out, int *block_sums)
{
    __shared__ int sh[VB];
    const int base = blockIdx.x * VB * SCAN_ITEMS + threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS], sum = 0;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; q++) { v[q] = base + q < n ? in[base + q] : 0; sum += v[q]; }
    sh[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < VB; off <<= 1) {
        int t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int run = sh[threadIdx.x] - sum;
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; q++) { if (base + q < n) out[base + q] = run; run += v[q]; }
    if (threadIdx.x == VB - 1) block_sums[blockIdx.x] = sh[VB - 1];
}
__global__ void k_scan_blocks(int nb, int *block_sums, int *total_out)
{   // one thread: nb is at most a few thousand
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < nb; b++) { int v = block_sums[b]; block_sums[b] = run; run += v; }
        *total_out = run;
    }
}
__global__ __launch_bounds__(VB) void k_scan_add(int n, int *out, const int *block_sums, const int *total)
{
    const int base = blockIdx.x * VB * SCAN_ITEMS + threadIdx.x * SCAN_ITEMS;
    const int add = block_sums[blockIdx.x];
#pragma unroll
    for (int q = 0; q < SCAN_ITEMS; q++) if (base + q < n) out[base + q] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

int device_exclusive_scan(int n, const int *counts, int *rowptr, hipStream_t s, long *total)
{
    const int nb = (n + VB * SCAN_ITEMS - 1) / (VB * SCAN_ITEMS);
    int *bs = nullptr;
    HIPCHK(hipMalloc(&bs, sizeof(int) * (nb + 1)));
    hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(VB), 0, s, n, counts, rowptr, bs);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1), 0, s, nb, bs, bs + nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(VB), 0, s, n, rowptr, bs, bs + nb);
    int tot = 0;
    hipError_t e = hipMemcpyAsync(&tot, bs + nb, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(bs);
    if (e != hipSuccess) return fail(e, "scan", __FILE__, __LINE__);
    *total = tot;
    return 0;
}

// ------------------------------------------------------------------------ synthetic family
// Bit-for-bit twin of oracle/csr_oracle.c (orc_gen_*): integer hashing, ascending columns,
// diagonal = sum of |off-diagonals| in column order + shift.
struct GenParams {
    long n; int npairs; long a[16], ainv[16], c[16]; int banded /* pattern 0 | 1 | 2 */, symmetric; unsigned long long seed; double shift;
    int wb_log2;
};

__host__ __device__ inline unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline unsigned long long mix3(unsigned long long a, unsigned long long b, unsigned long long seed)
{
    return splitmix64(splitmix64(a ^ seed) + b * 0xD6E8FEB86659FD93ull);
}
__host__ __device__ inline double unit_open0(unsigned long long h) { return (double)((h >> 11) + 1) * (1.0 / 9007199254740992.0); }
__host__ __device__ inline double unit_open1(unsigned long long h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

__device__ inline unsigned long long mulmod(unsigned long long a, unsigned long long b, unsigned long long n)
{   // a, b < n < 2^31 in practice; stay exact for anything below 2^63 via 128-bit product
    return (unsigned long long)(((unsigned __int128)a * b) % n);
}

// pattern 2 (row-random band): keyed bijection of [0, 2^L), twin of blk_fwd / blk_inv in oracle/csr_oracle.c
__device__ inline unsigned long long inv_pow2(unsigned long long a)
{
    unsigned long long x = a;
    for (int it = 0; it < 6; it++) x *= 2 - a * x;
    return x;
}
__device__ inline void blk_keys(const GenParams &g, int k, long b, unsigned long long *a1, unsigned long long *c1, unsigned long long *a2)
{
    const unsigned long long mask = (1ull << g.wb_log2) - 1;
    const unsigned long long key = mix3((unsigned long long)g.c[k], (unsigned long long)b, g.seed);
    *a1 = (key | 1) & mask; *c1 = (key >> 21) & mask; *a2 = ((key >> 42) | 1) & mask;
    if (g.wb_log2 == 0) { *a1 = 1; *a2 = 1; }
}
__device__ inline unsigned long long blk_fwd(const GenParams &g, int k, long b, unsigned long long u)
{
    const int L = g.wb_log2, sh = (L + 1) / 2;
    const unsigned long long mask = (1ull << L) - 1;
    unsigned long long a1, c1, a2; blk_keys(g, k, b, &a1, &c1, &a2);
    unsigned long long x = (u * a1 + c1) & mask;
    if (sh) x ^= x >> sh;
    x = (x * a2) & mask;
    if (sh) x ^= x >> sh;
    return x;
}
__device__ inline unsigned long long blk_inv(const GenParams &g, int k, long b, unsigned long long y)
{
    const int L = g.wb_log2, sh = (L + 1) / 2;
    const unsigned long long mask = (1ull << L) - 1;
    unsigned long long a1, c1, a2; blk_keys(g, k, b, &a1, &c1, &a2);
    unsigned long long x = y;
    if (sh) x ^= x >> sh;
    x = (x * inv_pow2(a2)) & mask;
    if (sh) x ^= x >> sh;
    return ((x - c1) * inv_pow2(a1)) & mask;
}

__device__ int gen_row_cols(const GenParams &g, long i, long *out)
{
    int cnt = 0;
    for (int k = 0; k < g.npairs; k++) {
        long j[2];
        if (g.banded == 2) {
            const int L = g.wb_log2;
            const long b = i >> L;
            const unsigned long long u = (unsigned long long)i & ((1ull << L) - 1);
            j[0] = ((b + 1) << L) + (long)blk_fwd(g, k, b, u);
            j[1] = b >= 1 ? ((b - 1) << L) + (long)blk_inv(g, k, b - 1, u) : -1;
        } else if (g.banded) { j[0] = i + g.c[k]; j[1] = i - g.c[k]; }
        else {
            j[0] = (long)((mulmod((unsigned long long)g.a[k], (unsigned long long)i, (unsigned long long)g.n) + (unsigned long long)g.c[k]) % (unsigned long long)g.n);
            long d = i - g.c[k]; if (d < 0) d += g.n;
            j[1] = (long)mulmod((unsigned long long)g.ainv[k], (unsigned long long)d, (unsigned long long)g.n);
        }
        for (int e = 0; e < 2; e++) {
            const long cc = j[e];
            if (cc < 0 || cc >= g.n || cc == i) continue;
            bool dup = false;
            for (int t = 0; t < cnt; t++) if (out[t] == cc) { dup = true; break; }
            if (dup) continue;
            int pos = cnt;
            while (pos > 0 && out[pos - 1] > cc) { out[pos] = out[pos - 1]; pos--; }
            out[pos] = cc; cnt++;
        }
    }
    return cnt;
}

__global__ void k_gen_count(GenParams g, long r0, long r1, int *counts)
{
    const long i = r0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r1) return;
    long tmp[32];
    counts[i - r0] = gen_row_cols(g, i, tmp) + 1;
}

__global__ void k_gen_fill(GenParams g, long r0, long r1, const int *rowptr, int *col, double *val)
{
    const long i = r0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r1) return;
    long tmp[32];
    const int cnt = gen_row_cols(g, i, tmp);
    const int base = rowptr[i - r0];
    int w = 0, dpos = -1;
    double sum = 0.0;
    for (int t = 0; t < cnt; t++) {
        if (dpos < 0 && tmp[t] > i) { dpos = base + w; w++; }
        const long j = tmp[t];
        const unsigned long long h = g.symmetric ? mix3((unsigned long long)(i < j ? i : j), (unsigned long long)(i < j ? j : i), g.seed)
                                                 : mix3((unsigned long long)i, (unsigned long long)j, g.seed);
        const double v = -unit_open0(h);
        col[base + w] = (int)j; val[base + w] = v; w++;
        sum += -v;
    }
    if (dpos < 0) { dpos = base + w; w++; }
    col[dpos] = (int)i;
    val[dpos] = sum + g.shift;
}

__global__ void k_gen_xtrue(unsigned long long seed, long r0, long r1, double *x)
{
    const long i = r0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < r1) x[i - r0] = unit_open1(mix3((unsigned long long)i, 0x7265757274ull, seed));
}

static long gcd64(long a, long b) { while (b) { long t = a % b; a = b; b = t; } return a; }
static long modinv(long a, long n)
{
    long t = 0, nt = 1, r = n, nr = a % n;
    while (nr) { long q = r / nr, tmp = t - q * nt; t = nt; nt = tmp; tmp = r - q * nr; r = nr; nr = tmp; }
    return t < 0 ? t + n : t;
}

static void gen_init(GenParams &g, long n, int npairs, int pattern, long band, int symmetric, unsigned long long seed, double shift)
{   // same draws as orc_gen_init_ex
    std::memset(&g, 0, sizeof g);
    g.n = n; g.npairs = npairs > 16 ? 16 : npairs; g.banded = pattern; g.symmetric = symmetric; g.seed = seed; g.shift = shift;
    unsigned long long s = splitmix64(seed ^ 0xA5A5A5A55A5A5A5Aull);
    if (band > n - 1) band = n - 1;
    if (band < 1) band = 1;
    if (pattern == 2) {
        int L = 0;
        while (L < 30 && (2L << L) <= band / 2) L++;
        g.wb_log2 = L;
        for (int k = 0; k < g.npairs; k++) { s = splitmix64(s); g.a[k] = 1; g.ainv[k] = 1; g.c[k] = (long)(s >> 1); }
        return;
    }
    for (int k = 0; k < g.npairs; k++) {
        if (g.banded) {
            long c = 1;
            for (int tries = 0; tries < 64; tries++) {
                s = splitmix64(s);
                c = (k == 0 || band < 2) ? 1 : 2 + (long)(s % (unsigned long long)(band - 1));
                bool dup = false;
                for (int j = 0; j < k; j++) if (g.c[j] == c) dup = true;
                if (!dup) break;
            }
            g.a[k] = 1; g.ainv[k] = 1; g.c[k] = c;
        } else {
            long a;
            do { s = splitmix64(s); a = 2 + (long)(s % (unsigned long long)(n > 3 ? n - 2 : 1)); } while (gcd64(a, n) != 1);
            s = splitmix64(s);
            g.a[k] = a; g.ainv[k] = modinv(a, n); g.c[k] = (long)(s % (unsigned long long)n);
        }
    }
}

// 5-point Laplacian, grid nx*ny, row-major numbering, diag 4 / off-diag -1
__global__ void k_lap_count(int nx, int ny, long r0, long r1, int *counts)
{
    const long i = r0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r1) return;
    const int ix = (int)(i % nx), iy = (int)(i / nx);
    counts[i - r0] = 1 + (ix > 0) + (ix < nx - 1) + (iy > 0) + (iy < ny - 1);
}
__global__ void k_lap_fill(int nx, int ny, long r0, long r1, const int *rowptr, int *col, double *val)
{
    const long i = r0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= r1) return;
    const int ix = (int)(i % nx), iy = (int)(i / nx);
    int k = rowptr[i - r0];
    if (iy > 0) { col[k] = (int)(i - nx); val[k++] = -1.0; }
    if (ix > 0) { col[k] = (int)(i - 1); val[k++] = -1.0; }
    col[k] = (int)i; val[k++] = 4.0;
    if (ix < nx - 1) { col[k] = (int)(i + 1); val[k++] = -1.0; }
    if (iy < ny - 1) { col[k] = (int)(i + nx); val[k++] = -1.0; }
}

// ------------------------------------------------------------------------------ COO ingest
__global__ void k_coo_sorted(long nnz, const int *row, int *unsorted_flag)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k + 1 < nnz && row[k] > row[k + 1]) *unsorted_flag = 1;
}
__global__ void k_coo_rowptr(long nnz, int n, const int *row, int *rowptr)
{   // row-sorted COO: rowptr[r] = first k with row[k] >= r
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nnz) return;
    const int prev = k == 0 ? -1 : row[k - 1];
    const int cur = k == nnz ? n : row[k];
    for (int r = prev + 1; r <= cur; r++) rowptr[r] = (int)k;
}

__global__ void k_coo_count(long nnz, int n, const int *row, int *cnt, int *bad)
{
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x) {
        const int r = row[k];
        if (r < 0 || r >= n) *bad = 1; else atomicAdd(&cnt[r], 1);
    }
}
__global__ void k_coo_place(long nnz, const int *row, const int *rowptr, int *next, int *perm)
{
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x) {
        const int r = row[k];
        perm[rowptr[r] + atomicAdd(&next[r], 1)] = (int)k;
    }
}
__global__ void k_perm_sort(int n, const int *rowptr, int *perm)
{   // ascending input position inside each row = the order a stable sort by row would give
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = rowptr[i], e = rowptr[i + 1];
    for (int a = s + 1; a < e; a++) {
        const int v = perm[a];
        int b = a - 1;
        while (b >= s && perm[b] > v) { perm[b + 1] = perm[b]; b--; }
        perm[b + 1] = v;
    }
}
template <class V>
__global__ void k_coo_gather(long nnz, const int *perm, const int *colin, const V *valin, int *col, V *val)
{
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x) {
        const int src = perm[k];
        col[k] = colin[src]; val[k] = valin[src];
    }
}

int alloc_part(CsrPart &P, int n_rows, long nnz, bool cplx)
{
    P.n_rows = n_rows; P.nnz = nnz; P.owned = true;
    HIPCHK(hipMalloc(&P.rowptr, sizeof(int) * ((size_t)n_rows + 1)));
    HIPCHK(hipMalloc(&P.col, sizeof(int) * (size_t)std::max<long>(nnz, 1) + 64));
    HIPCHK(hipMalloc(&P.val, sizeof(double) * (cplx ? 2 : 1) * (size_t)std::max<long>(nnz, 1) + 64));
    P.padded = true;
    return 0;
}


// ------------------------------------------------------------- op(A): A^T, A^H, conj(A)
// The reference's complex callback carries (layout, conjugate) (clcg.h:40-41); BiCG asks for
// A^H.x (clcg.cpp:187).  op(A) is materialised once as its own CSR (counting pass, scan,
// scatter, then a per-row sort so the summation order -- and with it the result -- does not
// depend on the order in which the scatter's atomics happened to land) and then multiplied by
// the same A.x kernels.
__global__ void k_tr_count(long nnz, const int *col, int *cnt)
{
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x)
        atomicAdd(&cnt[col[k]], 1);
}
template <class V>
__global__ void k_tr_fill(int n, const int *rowptr, const int *col, const V *val, const int *rpT, int *next, int *colT,
                          V *valT, int conj)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = rowptr[i]; k < rowptr[i + 1]; k++) {
        const int c = col[k];
        const int pos = rpT[c] + atomicAdd(&next[c], 1);
        colT[pos] = i;
        V v = val[k];
        if constexpr (sizeof(V) == 16) { if (conj) v.y = -v.y; }
        valT[pos] = v;
    }
}
__device__ __forceinline__ bool val_after(double a, double b) { return a > b; }
__device__ __forceinline__ bool val_after(double2 a, double2 b) { return a.x > b.x || (a.x == b.x && a.y > b.y); }
template <class V>
__global__ void k_row_sort(int n, const int *rowptr, int *col, V *val)
{   // insertion sort by (column, value) inside each row (rows are short).  The value tie-break
    // gives duplicate (row, col) entries a fixed order too, whatever order the scatter left.
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = rowptr[i], e = rowptr[i + 1];
    for (int a = s + 1; a < e; a++) {
        const int c = col[a]; const V v = val[a];
        int b = a - 1;
        while (b >= s && (col[b] > c || (col[b] == c && val_after(val[b], v)))) { col[b + 1] = col[b]; val[b + 1] = val[b]; b--; }
        col[b + 1] = c; val[b + 1] = v;
    }
}
__global__ void k_conj_copy(long nnz, const double2 *in, double2 *out)
{
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x)
        out[k] = make_double2(in[k].x, -in[k].y);
}

// part of `A` that realises op(A); built on first use.  layout/conjugate as in algebra.h:31-50.
int op_part(lcg_hip_csr *A, int layout, int conjugate, const CsrPart **out)
{
    if (!A->is_complex) conjugate = 0;
    const int idx = (layout ? 2 : 0) + (conjugate ? 1 : 0);
    if (idx == 0) { *out = &A->main; return 0; }
    CsrPart &T = A->op[idx];
    if (T.rowptr) { *out = &T; return 0; }
    // Sharded rows: this rank holds rows [row0, row0 + n) of A with GLOBAL columns.  Its share of op(A).x = A^T.x (A^H.x) is
    // (A_r)^T . x_r -- a vector of the matrix's full height, to which every rank contributes and of which every rank keeps
    // its own row block (comm.hip: dist_spmv_op, a reduce-scatter).  (A_r)^T is materialised like the unsharded transpose:
    // rows = the global columns, padded to ranks x rows-per-rank so that the reduce-scatter's blocks are equal; columns =
    // this rank's local rows.  conj(A) alone keeps the row split and is not offered on a sharded matrix.
    if (A->distributed && !layout) return fail(hipErrorInvalidValue, "conj(A).x is not available on a sharded matrix (A^T and A^H are)", __FILE__, __LINE__);
    if (!A->distributed && A->n_cols != A->n_rows) return fail(hipErrorInvalidValue, "op(A) needs a square matrix", __FILE__, __LINE__);
    Ctx &c = ctx();
    const int n = A->n_rows;                                    // rows of the source part
    long nt = n;                                                // rows of the transposed part
    if (A->distributed) {
        const long P = (A->n_global + A->rows_per_rank - 1) / A->rows_per_rank;
        nt = P * A->rows_per_rank;
        if (nt > 0x7fffffffL) return fail(hipErrorInvalidValue, "op(A): padded height exceeds int32", __FILE__, __LINE__);
    }
    const long nnz = A->main.nnz;
    int rc = alloc_part(T, (int)nt, nnz, A->is_complex);
    if (rc) return rc;
    T.n_cols = n;
    if (!layout) {      // conj(A): same structure
        HIPCHK(hipMemcpyAsync(T.rowptr, A->main.rowptr, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToDevice, c.stream));
        HIPCHK(hipMemcpyAsync(T.col, A->main.col, sizeof(int) * (size_t)nnz, hipMemcpyDeviceToDevice, c.stream));
        hipLaunchKernelGGL(k_conj_copy, dim3(1024), dim3(VB), 0, c.stream, nnz, reinterpret_cast<const double2 *>(A->main.val),
                           reinterpret_cast<double2 *>(T.val));
    } else {
        int *cnt = nullptr;
        HIPCHK(hipMalloc(&cnt, sizeof(int) * (size_t)nt));
        HIPCHK(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)nt, c.stream));
        hipLaunchKernelGGL(k_tr_count, dim3(1024), dim3(VB), 0, c.stream, nnz, A->main.col, cnt);
        long total = 0;
        rc = device_exclusive_scan((int)nt, cnt, T.rowptr, c.stream, &total);
        if (rc || total != nnz) { hipFree(cnt); return rc ? rc : fail(hipErrorUnknown, "transpose count", __FILE__, __LINE__); }
        HIPCHK(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)nt, c.stream));
        const unsigned g = (unsigned)((n + VB - 1) / VB), gt = (unsigned)((nt + VB - 1) / VB);
        if (A->is_complex) {
            hipLaunchKernelGGL((k_tr_fill<double2>), dim3(g), dim3(VB), 0, c.stream, n, A->main.rowptr, A->main.col,
                               reinterpret_cast<const double2 *>(A->main.val), T.rowptr, cnt, T.col, reinterpret_cast<double2 *>(T.val), conjugate);
            hipLaunchKernelGGL((k_row_sort<double2>), dim3(gt), dim3(VB), 0, c.stream, (int)nt, T.rowptr, T.col, reinterpret_cast<double2 *>(T.val));
        } else {
            hipLaunchKernelGGL((k_tr_fill<double>), dim3(g), dim3(VB), 0, c.stream, n, A->main.rowptr, A->main.col, A->main.val,
                               T.rowptr, cnt, T.col, T.val, 0);
            hipLaunchKernelGGL((k_row_sort<double>), dim3(gt), dim3(VB), 0, c.stream, (int)nt, T.rowptr, T.col, T.val);
        }
        hipError_t e = hipStreamSynchronize(c.stream);
        hipFree(cnt);
        if (e != hipSuccess) return fail(e, "transpose build", __FILE__, __LINE__);
    }
    HIPCHK(hipGetLastError());
    *out = &T;
    return 0;
}

} // namespace lcgh

using namespace lcgh;

extern "C" {

int lcg_hip_csr_from_coo(lcg_hip_csr_t *out, int n, int64_t nnz, const int *row, const int *col, const double *val,
                         int is_complex, int mem)
{
    if (!out || n <= 0 || nnz <= 0 || nnz > 0x7fffffffLL || !row || !col || !val) return LCG_HIP_E_ARG;
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    const size_t vw = is_complex ? 2 : 1;
    lcg_hip_csr *A = new lcg_hip_csr();
    A->n_rows = n; A->n_cols = n; A->is_complex = is_complex != 0; A->mean_row = (double)nnz / n;
    rc = alloc_part(A->main, n, nnz, A->is_complex);
    if (rc) { delete A; return rc; }
    A->main.n_cols = n;
    int *d_row = nullptr, *d_flag = nullptr;
    const hipMemcpyKind kind = mem == LCG_HIP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    auto bail = [&](int code) { if (d_row) hipFree(d_row); if (d_flag) hipFree(d_flag); free_part(A->main); delete A; return code; };
    if (hipMalloc(&d_row, sizeof(int) * (size_t)nnz) != hipSuccess || hipMalloc(&d_flag, sizeof(int)) != hipSuccess)
        return bail(fail(hipErrorOutOfMemory, "coo staging", __FILE__, __LINE__));
    hipError_t e = hipMemcpyAsync(d_row, row, sizeof(int) * (size_t)nnz, kind, c.stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, sizeof(int), c.stream);
    if (e != hipSuccess) return bail(fail(e, "coo upload", __FILE__, __LINE__));
    const unsigned gb = (unsigned)((nnz + 1 + VB - 1) / VB);
    hipLaunchKernelGGL(k_coo_sorted, dim3(gb), dim3(VB), 0, c.stream, (long)nnz, d_row, d_flag);
    int unsorted = 0;
    e = hipMemcpyAsync(&unsorted, d_flag, sizeof(int), hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    if (e != hipSuccess) return bail(fail(e, "coo sortedness", __FILE__, __LINE__));
    if (!unsorted) {
        // the bundled files are row-major sorted: entries stay in place
        hipLaunchKernelGGL(k_coo_rowptr, dim3(gb), dim3(VB), 0, c.stream, (long)nnz, n, d_row, A->main.rowptr);
        e = hipMemcpyAsync(A->main.col, col, sizeof(int) * (size_t)nnz, kind, c.stream);
        if (e == hipSuccess) e = hipMemcpyAsync(A->main.val, val, sizeof(double) * vw * (size_t)nnz, kind, c.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
        if (e != hipSuccess) return bail(fail(e, "coo copy", __FILE__, __LINE__));
    } else {
        // unsorted input: stable counting sort by row on the device.  Entries are scattered into
        // their rows with atomics (arbitrary order), then each row's slots are sorted by the
        // entry's position in the input, which restores the input order inside every row --
        // the same result as a stable host sort, whatever order the atomics landed in.
        int *cnt = nullptr, *perm = nullptr, *d_colin = nullptr;
        double *d_valin = nullptr;
        auto bail2 = [&](int code) {
            if (cnt) hipFree(cnt); if (perm) hipFree(perm);
            if (mem != LCG_HIP_MEM_DEVICE) { if (d_colin) hipFree(d_colin); if (d_valin) hipFree(d_valin); }
            return bail(code);
        };
        if (hipMalloc(&cnt, sizeof(int) * (size_t)n) != hipSuccess || hipMalloc(&perm, sizeof(int) * (size_t)nnz) != hipSuccess)
            return bail2(fail(hipErrorOutOfMemory, "coo sort workspace", __FILE__, __LINE__));
        if (mem == LCG_HIP_MEM_DEVICE) { d_colin = const_cast<int *>(col); d_valin = const_cast<double *>(val); }
        else {
            if (hipMalloc(&d_colin, sizeof(int) * (size_t)nnz) != hipSuccess || hipMalloc(&d_valin, sizeof(double) * vw * (size_t)nnz) != hipSuccess)
                return bail2(fail(hipErrorOutOfMemory, "coo staging", __FILE__, __LINE__));
            e = hipMemcpyAsync(d_colin, col, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice, c.stream);
            if (e == hipSuccess) e = hipMemcpyAsync(d_valin, val, sizeof(double) * vw * (size_t)nnz, hipMemcpyHostToDevice, c.stream);
            if (e != hipSuccess) return bail2(fail(e, "coo upload", __FILE__, __LINE__));
        }
        e = hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)n, c.stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_flag, 0, sizeof(int), c.stream);
        if (e != hipSuccess) return bail2(fail(e, "coo sort init", __FILE__, __LINE__));
        hipLaunchKernelGGL(k_coo_count, dim3(1024), dim3(VB), 0, c.stream, (long)nnz, n, d_row, cnt, d_flag);
        int bad = 0;
        e = hipMemcpyAsync(&bad, d_flag, sizeof(int), hipMemcpyDeviceToHost, c.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
        if (e != hipSuccess) return bail2(fail(e, "coo count", __FILE__, __LINE__));
        if (bad) return bail2(LCG_HIP_E_ARG);                          // a row index outside [0, n)
        long total = 0;
        int rc2 = device_exclusive_scan(n, cnt, A->main.rowptr, c.stream, &total);
        if (rc2 || total != nnz) return bail2(rc2 ? rc2 : LCG_HIP_E_ARG);
        e = hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)n, c.stream);
        if (e != hipSuccess) return bail2(fail(e, "coo sort", __FILE__, __LINE__));
        hipLaunchKernelGGL(k_coo_place, dim3(1024), dim3(VB), 0, c.stream, (long)nnz, d_row, A->main.rowptr, cnt, perm);
        hipLaunchKernelGGL(k_perm_sort, dim3((unsigned)((n + VB - 1) / VB)), dim3(VB), 0, c.stream, n, A->main.rowptr, perm);
        if (is_complex)
            hipLaunchKernelGGL((k_coo_gather<double2>), dim3(1024), dim3(VB), 0, c.stream, (long)nnz, perm, d_colin,
                               reinterpret_cast<const double2 *>(d_valin), A->main.col, reinterpret_cast<double2 *>(A->main.val));
        else
            hipLaunchKernelGGL((k_coo_gather<double>), dim3(1024), dim3(VB), 0, c.stream, (long)nnz, perm, d_colin, d_valin,
                               A->main.col, A->main.val);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
        hipFree(cnt); hipFree(perm);
        if (mem != LCG_HIP_MEM_DEVICE) { hipFree(d_colin); hipFree(d_valin); }
        if (e != hipSuccess) return bail(fail(e, "coo device sort", __FILE__, __LINE__));
    }
    hipFree(d_row); hipFree(d_flag);
    *out = A;
    return 0;
}

int lcg_hip_csr_build_jacobi(lcg_hip_csr_t A, double *diag_out)
{
    if (!A) return LCG_HIP_E_ARG;
    Ctx &c = ctx();
    const size_t w = A->is_complex ? 2 : 1;
    if (!A->invdiag) HIPCHK(hipMalloc(&A->invdiag, sizeof(double) * w * (size_t)A->n_rows));
    const unsigned g = (unsigned)((A->n_rows + VB - 1) / VB);
    if (A->is_complex)
        hipLaunchKernelGGL((k_diag<double2>), dim3(g), dim3(VB), 0, c.stream, A->n_rows, (long)A->row0, A->main.rowptr, A->main.col,
                           reinterpret_cast<const double2 *>(A->main.val), reinterpret_cast<double2 *>(diag_out),
                           reinterpret_cast<double2 *>(A->invdiag));
    else
        hipLaunchKernelGGL((k_diag<double>), dim3(g), dim3(VB), 0, c.stream, A->n_rows, (long)A->row0, A->main.rowptr, A->main.col,
                           A->main.val, diag_out, A->invdiag);
    HIPCHK(hipGetLastError());
    return 0;
}


int lcg_hip_vecmul(int n, const double *a, const double *b, double *out)
{
    int rc = ensure_init(); if (rc) return rc;
    int g = launch_vec(OpMul{nullptr, a, b, out}, n, (uintptr_t)a | (uintptr_t)b | (uintptr_t)out, ctx().stream, nullptr);
    return g < 0 ? g : 0;
}
int lcg_hip_vecdiv(int n, const double *a, const double *b, double *out)
{
    int rc = ensure_init(); if (rc) return rc;
    int g = launch_vec(OpDiv{nullptr, a, b, out}, n, (uintptr_t)a | (uintptr_t)b | (uintptr_t)out, ctx().stream, nullptr);
    return g < 0 ? g : 0;
}
int clcg_hip_vecdiv(int n, const double *a, const double *b, double *out)
{
    int rc = ensure_init(); if (rc) return rc;
    hipLaunchKernelGGL(k_cdiv, dim3(grid_for(n)), dim3(VB), 0, ctx().stream, (long)n, reinterpret_cast<const double2 *>(a),
                       reinterpret_cast<const double2 *>(b), reinterpret_cast<double2 *>(out));
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- generators ----------------------------------------------------------------------------------
static int finish_generated(lcg_hip_csr *A, int nloc, long r0, int *counts, hipStream_t s,
                            const std::function<void(const int *, int *, double *)> &fill)
{
    long total = 0;
    int *rowptr = nullptr;
    hipError_t e = hipMalloc(&rowptr, sizeof(int) * ((size_t)nloc + 1));
    if (e != hipSuccess) { hipFree(counts); return fail(e, "rowptr", __FILE__, __LINE__); }
    int rc = device_exclusive_scan(nloc, counts, rowptr, s, &total);
    hipFree(counts);
    if (rc) { hipFree(rowptr); return rc; }
    A->main.n_rows = nloc; A->main.nnz = total; A->main.owned = true; A->main.rowptr = rowptr;
    e = hipMalloc(&A->main.col, sizeof(int) * (size_t)total + 64);
    if (e == hipSuccess) e = hipMalloc(&A->main.val, sizeof(double) * (size_t)total + 64);
    A->main.padded = true;
    if (e != hipSuccess) return fail(e, "generated arrays", __FILE__, __LINE__);
    fill(rowptr, A->main.col, A->main.val);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    A->mean_row = (double)total / nloc;
    A->row0 = r0;
    A->main.n_cols = A->n_cols;
    return 0;
}

int lcg_hip_csr_generate(lcg_hip_csr_t *out, int64_t n, int npairs, int64_t band, int symmetric, uint64_t seed,
                         double diag_shift, int64_t r0, int64_t r1)
{
    return lcg_hip_csr_generate_ex(out, n, npairs, band > 0 ? LCG_HIP_GEN_DIAGONALS : LCG_HIP_GEN_SCRAMBLED, band, symmetric, seed,
                                   diag_shift, r0, r1);
}

int lcg_hip_csr_generate_ex(lcg_hip_csr_t *out, int64_t n, int npairs, int pattern, int64_t band, int symmetric, uint64_t seed,
                            double diag_shift, int64_t r0, int64_t r1)
{
    if (!out || n <= 1 || n > 0x7fffffffLL || r0 < 0 || r1 > n || r1 <= r0 || npairs < 1 || pattern < 0 || pattern > 2) return LCG_HIP_E_ARG;
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    GenParams g; gen_init(g, n, npairs, pattern, band, symmetric, seed, diag_shift);
    const int nloc = (int)(r1 - r0);
    int *counts = nullptr;
    HIPCHK(hipMalloc(&counts, sizeof(int) * (size_t)nloc));
    const unsigned gb = (unsigned)((nloc + VB - 1) / VB);
    hipLaunchKernelGGL(k_gen_count, dim3(gb), dim3(VB), 0, c.stream, g, (long)r0, (long)r1, counts);
    lcg_hip_csr *A = new lcg_hip_csr();
    A->n_rows = nloc; A->n_cols = (int)n; A->is_complex = false;
    rc = finish_generated(A, nloc, r0, counts, c.stream, [&](const int *rp, int *col, double *val) {
        hipLaunchKernelGGL(k_gen_fill, dim3(gb), dim3(VB), 0, c.stream, g, (long)r0, (long)r1, rp, col, val);
    });
    if (rc) { free_part(A->main); delete A; return rc; }
    *out = A;
    return 0;
}

int lcg_hip_gen_xtrue(int64_t n, uint64_t seed, int64_t r0, int64_t r1, double *x_dev)
{
    if (r0 < 0 || r1 > n || r1 <= r0 || !x_dev) return LCG_HIP_E_ARG;
    int rc = ensure_init(); if (rc) return rc;
    const unsigned gb = (unsigned)((r1 - r0 + VB - 1) / VB);
    hipLaunchKernelGGL(k_gen_xtrue, dim3(gb), dim3(VB), 0, ctx().stream, (unsigned long long)seed, (long)r0, (long)r1, x_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

int lcg_hip_csr_laplace2d(lcg_hip_csr_t *out, int nx, int ny, int64_t r0, int64_t r1)
{
    const int64_t n = (int64_t)nx * ny;
    if (!out || nx < 1 || ny < 1 || n > 0x7fffffffLL || r0 < 0 || r1 > n || r1 <= r0) return LCG_HIP_E_ARG;
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    const int nloc = (int)(r1 - r0);
    int *counts = nullptr;
    HIPCHK(hipMalloc(&counts, sizeof(int) * (size_t)nloc));
    const unsigned gb = (unsigned)((nloc + VB - 1) / VB);
    hipLaunchKernelGGL(k_lap_count, dim3(gb), dim3(VB), 0, c.stream, nx, ny, (long)r0, (long)r1, counts);
    lcg_hip_csr *A = new lcg_hip_csr();
    A->n_rows = nloc; A->n_cols = (int)n; A->is_complex = false;
    rc = finish_generated(A, nloc, r0, counts, c.stream, [&](const int *rp, int *col, double *val) {
        hipLaunchKernelGGL(k_lap_fill, dim3(gb), dim3(VB), 0, c.stream, nx, ny, (long)r0, (long)r1, rp, col, val);
    });
    if (rc) { free_part(A->main); delete A; return rc; }
    *out = A;
    return 0;
}

} // extern "C"
