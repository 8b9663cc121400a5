// slab_model: what would an Infinity-Cache slab walk of the binned product (csr_binned.hip) cost?  A TRAFFIC MODEL of the two passes at
// full parallelism -- same bytes per entry, same access shapes, no plan -- run slab by slab for slab sizes from 3M entries to "all":
//   pass 1 (a workgroup per column tile and slab): the tile's 64 KB slice of x into LDS (x: 80 MB, the same for every slab), then
//           per entry 2 B of tile-relative column + 0.5 B of destination, an LDS read, and 8 B written to xg in whole 64-byte
//           granules at scattered places of the SLAB's part of xg (an affine permutation of the slab's granules);
//   pass 2 (a wavefront per 2048 entries): xg 8 B + val 8 B + row 2 B per entry streamed, a sum kept, one double written.
// Read-once streams (columns, values, rows) are as large as the real ones (fresh bytes for every slab: they come from HBM); xg is
// ONE buffer reused by every slab, so that with small slabs its 16 B per entry of round trip can stay in the 256 MB Infinity Cache.
// The question: is (all bytes) / time better with slabs that fit the cache than with one slab = the product as it runs today?
//   hipcc --offload-arch=gfx950 -O3 scripts/slab_model.hip -o scripts/bin/slab_model
//   scripts/bin/slab_model [entries=330000000] [x_doubles=10000000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
constexpr int TILE = 8192;

// one workgroup (512 threads) per (tile, piece): entries [e0, e1) of stream 1, granules of the slab [g_base, g_base + g_n)
__global__ __launch_bounds__(512) void k_p1(const double *__restrict__ x, long ncols, int ntiles, const unsigned *__restrict__ col2, const int *__restrict__ dst,
                                            double *__restrict__ xg, long e_first, long per_item, long e_end, long g_base, long g_n, long perm_a, int nt_stores)
{
    __shared__ __attribute__((aligned(16))) double sx[TILE];
    const int tid = threadIdx.x;
    const int t = blockIdx.x % ntiles;
    const long c0 = (long)t * TILE;
    {
        v2d v[TILE / 2 / 512];
#pragma unroll
        for (int q = 0; q < TILE / 2 / 512; q++) { long i = c0 + 2L * (q * 512 + tid); v[q] = i + 1 < ncols ? *reinterpret_cast<const v2d *>(x + i) : v2d{0.0, 0.0}; }
#pragma unroll
        for (int q = 0; q < TILE / 2 / 512; q++) reinterpret_cast<v2d *>(sx)[q * 512 + tid] = v[q];
    }
    __syncthreads();
    const long e0 = e_first + (long)blockIdx.x * per_item, e1 = min(e_end, e0 + per_item);     // in PAIRS of entries
    constexpr int UN = 8;
    for (long e = e0 + tid; e < e1; e += (long)UN * 512) {
        unsigned lc[UN]; int d[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) { long eu = e + (long)u * 512; long ec = eu < e1 ? eu : e; lc[u] = col2[ec]; d[u] = dst[ec >> 2]; }
        v2d v[UN];
#pragma unroll
        for (int u = 0; u < UN; u++) { v[u].x = sx[lc[u] & (TILE - 1)]; v[u].y = sx[(lc[u] >> 16) & (TILE - 1)]; }
#pragma unroll
        for (int u = 0; u < UN; u++) {
            long eu = e + (long)u * 512;
            if (eu < e1) {
                // the granule this pair belongs to, scattered over the slab's granules (the real plan writes whole granules too)
                // (scattered in runs of 8 granules = 512 B: a (chunk, tile) group of the 10M-row scrambled system holds ~55 entries)
                long g = ((eu >> 2) - (e_first >> 2) + (long)(d[u] & 1)) % g_n;
                const long units = g_n >> 3;
                if (units > 1 && (g >> 3) < units) g = (((g >> 3) * perm_a) % units) * 8 + (g & 7);
                v2d *p = reinterpret_cast<v2d *>(xg) + 4 * (g_base + g) + (eu & 3);
                if (nt_stores) __builtin_nontemporal_store(v[u], p); else *p = v[u];
            }
        }
    }
}

// one wavefront per 2048 entries of the slab
__global__ __launch_bounds__(256) void k_p2(const double *__restrict__ xg, const double *__restrict__ val, const u16 *__restrict__ rows, double *__restrict__ y,
                                            long xg_first, long s_first, long s_end, int nt_xg)
{
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const long chunk = (long)blockIdx.x * 4 + w;
    const long p0 = chunk * 2048, p1 = min(s_end - s_first, p0 + 2048);
    double acc = 0.0;
    for (long p = p0; p < p1; p += 1024) {
        v2d xa[4], va[4], xb[4], vb[4];
        const long pb = p + 512 < p1 ? p + 512 : p;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const v2d *px = reinterpret_cast<const v2d *>(xg + xg_first + p + i * 128 + 2 * l), *pxb = reinterpret_cast<const v2d *>(xg + xg_first + pb + i * 128 + 2 * l);
            xa[i] = nt_xg ? __builtin_nontemporal_load(px) : *px;
            va[i] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(val + s_first + p + i * 128 + 2 * l));
            xb[i] = nt_xg ? __builtin_nontemporal_load(pxb) : *pxb;
            vb[i] = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(val + s_first + pb + i * 128 + 2 * l));
        }
        const uint4 ra = *reinterpret_cast<const uint4 *>(rows + s_first + p + 8 * l), rb = *reinterpret_cast<const uint4 *>(rows + s_first + pb + 8 * l);
#pragma unroll
        for (int i = 0; i < 4; i++) acc += va[i].x * xa[i].x + va[i].y * xa[i].y + vb[i].x * xb[i].x + vb[i].y * xb[i].y;
        acc += (double)((ra.x ^ ra.y ^ ra.z ^ ra.w ^ rb.x ^ rb.y ^ rb.z ^ rb.w) & 1u);
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (l == 0 && p0 < p1) y[(s_first >> 11) + chunk] = acc;
}

int main(int argc, char **argv)
{
    const long E = argc > 1 ? atol(argv[1]) : 330000000L;
    const long NX = argc > 2 ? atol(argv[2]) : 10000000L;
    const int ntiles = (int)((NX + TILE - 1) / TILE);
    hipStream_t s; CK(hipStreamCreate(&s));
    const long EP = (E + 8191) / 8192 * 8192;
    double *x, *xg, *val, *y; unsigned *col2; int *dst; u16 *rows;
    CK(hipMalloc(&x, 8 * NX)); CK(hipMalloc(&val, 8 * (EP + 1024))); CK(hipMalloc(&xg, 8 * (EP + 1024))); CK(hipMalloc(&y, 8 * (EP / 2048 + 8)));
    CK(hipMalloc(&col2, 2 * (EP + 1024))); CK(hipMalloc(&dst, (EP / 8 + 64) * 4)); CK(hipMalloc(&rows, 2 * (EP + 1024)));
    CK(hipMemset(x, 0, 8 * NX)); CK(hipMemset(val, 0, 8 * (EP + 1024))); CK(hipMemset(xg, 0, 8 * (EP + 1024)));
    {   // columns: pseudo-random 13-bit pairs; dst: parity bits
        std::vector<unsigned> h((size_t)1 << 22);
        unsigned r = 12345u;
        for (auto &v : h) { r = r * 1664525u + 1013904223u; v = (r >> 3) & 0x1fff1fffu; }
        for (long off = 0; off < (EP + 1024) / 2; off += (long)h.size()) {
            size_t cnt = (size_t)std::min<long>((long)h.size(), (EP + 1024) / 2 - off);
            CK(hipMemcpy(col2 + off, h.data(), cnt * 4, hipMemcpyHostToDevice));
        }
        CK(hipMemset(dst, 0, (EP / 8 + 64) * 4)); CK(hipMemset(rows, 0, 2 * (EP + 1024)));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("entries %ld, x %ld doubles (%d tiles); bytes per product: streams %.2f GB + xg round trip %.2f GB\n", E, NX, ntiles,
           12.5 * E / 1e9, 16.0 * E / 1e9);
    // slab sizes (entries); the last one is "no slabs"
    std::vector<long> sizes = {3000000, 6000000, 9000000, 12000000, 18000000, 24000000, 48000000, E};
    for (int nt = 1; nt >= 0; --nt)
        for (long S : sizes) {
            S = std::min(S, E);
            const long nslab = (E + S - 1) / S;
            // pass-1 work items per slab: every tile, in pieces of <= 65536 entries (as plan_build cuts them)
            const long per_tile = (S + ntiles - 1) / ntiles;                   // entries of a slab per tile
            const long pieces = std::max<long>(1, (per_tile + 65535) / 65536);
            const long per_item_pairs = ((per_tile + pieces - 1) / pieces + 1) / 2;
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0, s));
                for (long k = 0; k < nslab; k++) {
                    const long s0 = k * S, s1 = std::min(E, s0 + S);
                    const long g_n = (s1 - s0 + 7) / 8;
                    const long xg_first = nslab > 1 ? 0 : s0;       // slabs reuse the FRONT of xg; one slab = the whole of it
                    const int items = (int)(ntiles * pieces);
                    hipLaunchKernelGGL(k_p1, dim3(items), dim3(512), 0, s, x, NX, ntiles, col2, dst, xg, s0 / 2, per_item_pairs, s1 / 2, xg_first / 8, g_n,
                                       (long)1000003, nt);
                    const long chunks = (s1 - s0 + 2047) / 2048;
                    hipLaunchKernelGGL(k_p2, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, s, xg, val, rows, y, xg_first, s0, s1, nt);
                }
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            CK(hipGetLastError());
            const double xload = (double)nslab * pieces * ntiles * TILE * 8;
            printf("slab %10ld entries (xg %6.1f MB) x %4ld slabs, %s xg: %8.1f us per product; x slices loaded %.2f GB; (streams + xg + x) / time = %.2f TB/s; streams-only rate %.2f TB/s\n",
                   S, 8.0 * S / 1e6, nslab, nt ? "non-temporal" : "default-policy", best * 1e3, xload / 1e9,
                   (28.5 * E + xload) / (best * 1e-3) / 1e12, 12.5 * E / (best * 1e-3) / 1e12);
            fflush(stdout);
        }
    return 0;
}
