// colslab_model: the LAST alternative for scattered columns that round 4 left as an estimate ("an L2-resident x -- slabs of
// columns, row sums in LDS -- was priced: 3.3e8 gathers x 128-byte lines = 42 GB through the L2s ... = 1.4 ms before the stream"), MEASURED.
// A TRAFFIC MODEL at full size and full parallelism, no plan builder:
//   the columns are cut into slabs of S doubles of x (S x 8 B = 0.5 .. 4 MB: what one XCD's 4 MB L2 can keep), the entries of a
//   (row chunk, slab) pair are stored together in row order as 12 bytes -- the value and one 32-bit word holding the column inside the
//   slab and the row inside the chunk --, streamed once (non-temporal); x is GATHERED from the slab (from the L2, if the plan works);
//   form A: a workgroup per chunk of R rows keeps the R row sums in LDS (ds_add_f64) while it walks the slabs in order -- every
//           workgroup of an XCD walks them in the same order, so at any time an XCD gathers from one or two slabs -- and writes y once;
//   form B: (the form VERDICT r4 describes) slab-major: a workgroup per (slab, chunk), y read, added to and written back per slab.
// Same entry count and x length as the 10M-row scrambled system of bench.py (3.3e8 entries, 1e7 columns; ~0.8 entries per row and slab).
// To be read against the two-pass binned product it would replace: 1725-1850 us per product on the same boxes (28.5 B per entry streamed).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics scripts/colslab_model.hip -o scripts/bin/colslab_model
//   scripts/bin/colslab_model [entries=330000000] [x_doubles=10000000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int TB = 256;
constexpr int UN = 8;

__device__ __forceinline__ unsigned hash32(unsigned long long v)
{
    v ^= v >> 33; v *= 0xff51afd7ed558ccdULL; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ULL; v ^= v >> 33;
    return (unsigned)v;
}

// the packed words: entry j of pair (chunk c, slab s) -> column hash in [0, S), row = j * R / per_pair (ascending inside the pair)
__global__ void k_fill(unsigned *pk, double *val, long n_entries, long per_pair, int R, unsigned smask, int row_shift)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += (long)gridDim.x * blockDim.x) {
        const long j = e % per_pair;
        const unsigned row = (unsigned)((j * R) / per_pair);
        pk[e] = (hash32((unsigned long long)e) & smask) | (row << row_shift);
        val[e] = 1.0 + 1e-9 * (double)(e & 1023);
    }
}
__global__ void k_fill_x(double *x, long n) { for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] = 1.0 + 1e-6 * (double)(i & 4095); }

// one pair's entries into the LDS sums
__device__ __forceinline__ void pair_into(const double *__restrict__ xs, const double *__restrict__ val, const unsigned *__restrict__ pk, long e0, long e1,
                                          double *sums, unsigned smask, int row_shift)
{
    for (long e = e0 + threadIdx.x; e < e1; e += (long)UN * TB) {
        unsigned w[UN]; double a[UN], xv[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) {
            const long eu = e + (long)u * TB;
            const long ec = eu < e1 ? eu : e;
            w[u] = __builtin_nontemporal_load(pk + ec);
            a[u] = eu < e1 ? __builtin_nontemporal_load(val + ec) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UN; u++) xv[u] = xs[w[u] & smask];
#pragma unroll
        for (int u = 0; u < UN; u++) unsafeAtomicAdd(&sums[w[u] >> row_shift], a[u] * xv[u]);
    }
}

// form A: workgroup = chunk, slabs walked in order, y written once
__global__ __launch_bounds__(TB) void k_form_a(const double *__restrict__ x, long S, int NS, const double *__restrict__ val, const unsigned *__restrict__ pk,
                                               double *__restrict__ y, long per_pair, int R, unsigned smask, int row_shift)
{
    extern __shared__ double sums[];
    for (int i = threadIdx.x; i < R; i += TB) sums[i] = 0.0;
    __syncthreads();
    const long c = blockIdx.x;
    for (int s = 0; s < NS; s++) {
        const long e0 = (c * NS + s) * per_pair;
        pair_into(x + (long)s * S, val, pk, e0, e0 + per_pair, sums, smask, row_shift);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += TB) y[c * R + i] = sums[i];
}

// form B: slab-major, workgroup = (slab, chunk), y read-modify-written per slab
__global__ __launch_bounds__(TB) void k_form_b(const double *__restrict__ x, long S, int NS, long nchunks, const double *__restrict__ val, const unsigned *__restrict__ pk,
                                               double *__restrict__ y, long per_pair, int R, unsigned smask, int row_shift)
{
    extern __shared__ double sums[];
    const long s = blockIdx.x / nchunks, c = blockIdx.x % nchunks;
    for (int i = threadIdx.x; i < R; i += TB) sums[i] = s == 0 ? 0.0 : y[c * R + i];
    __syncthreads();
    const long e0 = (s * nchunks + c) * per_pair;
    pair_into(x + s * S, val, pk, e0, e0 + per_pair, sums, smask, row_shift);
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += TB) y[c * R + i] = sums[i];
}

int main(int argc, char **argv)
{
    const long entries = argc > 1 ? atol(argv[1]) : 330000000L;
    const long ncols = argc > 2 ? atol(argv[2]) : 10000000L;
    const long nrows = ncols;
    double *x, *val, *y; unsigned *pk;
    CK(hipMalloc(&x, sizeof(double) * (size_t)(ncols + (1 << 20))));
    CK(hipMalloc(&val, sizeof(double) * (size_t)entries));
    CK(hipMalloc(&pk, sizeof(unsigned) * (size_t)entries));
    CK(hipMalloc(&y, sizeof(double) * (size_t)(nrows + 16384)));
    hipLaunchKernelGGL(k_fill_x, dim3(4096), dim3(256), 0, 0, x, ncols + (1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("entries %ld, x %ld doubles; 12 B per entry streamed = %.2f GB + x + y; to beat: the two-pass binned product, 1725-1850 us\n", entries, ncols, 12.0 * entries / 1e9);
    for (int form = 0; form < 2; form++)
        for (long S : {65536L, 131072L, 262144L, 524288L})
            for (int R : {2048, 4096, 8192}) {
                const int NS = (int)((ncols + S - 1) / S);
                const long nchunks = (nrows + R - 1) / R;
                const long per_pair = entries / (nchunks * NS);
                const long used = per_pair * nchunks * NS;
                int sbits = 0; while ((1L << sbits) < S) sbits++;
                int rbits = 0; while ((1 << rbits) < R) rbits++;
                if (sbits + rbits > 32) continue;
                hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, pk, val, used, per_pair, R, (unsigned)(S - 1), sbits);
                CK(hipDeviceSynchronize());
                float best = 1e30f, sum = 0.f;
                const int reps = 4;
                for (int r = 0; r < reps + 1; r++) {
                    CK(hipEventRecord(e0, 0));
                    if (form == 0)
                        hipLaunchKernelGGL(k_form_a, dim3((unsigned)nchunks), dim3(TB), sizeof(double) * R, 0, x, S, NS, val, pk, y, per_pair, R, (unsigned)(S - 1), sbits);
                    else
                        hipLaunchKernelGGL(k_form_b, dim3((unsigned)(nchunks * NS)), dim3(TB), sizeof(double) * R, 0, x, S, NS, nchunks, val, pk, y, per_pair, R, (unsigned)(S - 1), sbits);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r > 0) { best = std::min(best, ms); sum += ms; }
                }
                const double us = 1e3 * sum / reps;
                const double ybytes = form == 0 ? 8.0 * nrows : 16.0 * nrows * NS;
                printf("form %c  slab %7ld columns (%4.1f MB of x) x %3d slabs, chunk %5d rows (%2d KB of LDS), %5.1f entries per pair and row-chunk %ld: %8.1f us per product "
                       "(best %8.1f); stream %.2f GB + y %.2f GB -> %.2f TB/s\n", form == 0 ? 'A' : 'B', S, 8.0 * S / 1048576.0, NS, R, R / 128, (double)per_pair, per_pair,
                       us, 1e3 * best, 12.0 * used / 1e9, ybytes / 1e9, (12.0 * used + ybytes) / (us * 1e-6) / 1e12);
                fflush(stdout);
            }
    return 0;
}
