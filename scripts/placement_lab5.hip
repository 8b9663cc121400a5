// placement_lab5: the two states of a box, by HOW the vectors (and, in the last phases, the matrix arrays) were allocated.
// One process, one generated headline matrix (10M rows, 33 constant diagonals); the product y = A.x through the C ABI with
//   A  hipMalloc pairs, each vector an allocation of its own (round 3's lab: the first pair fast, the later ones slow)
//   B  pairs mapped through the virtual-memory API: ONE physical handle per vector (hipMemCreate at the recommended granularity)
//   C  the same, the physical memory created in 2 MB handles (one per granule) and mapped side by side
//   D  pairs cut from one 1 GiB arena (one hipMemCreate, one mapping)
//   E  crossings: x of one family with y of another
// Every product is ONE dispatch of the A.x kernel; the manifest lines ("pair <tag> dispatches [a, b) us <t>") tie the dispatches
// of a `rocprofv3 --pmc` run of this program to the pairs (scripts/make_placement_summary.py).
//   hipcc --offload-arch=gfx950 -O2 -I include scripts/placement_lab5.hip -o scripts/bin/placement_lab5 -Lliblcg_amd/lib -llcg_hip -Wl,-rpath,'$ORIGIN/../../liblcg_amd/lib'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "lcg_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

static lcg_hip_csr_t A;
static hipStream_t S;
static int64_t N = 10000000;
static int dispatch_no = 0;
static int REPS = 8;
static int PROBES = 0;

__global__ void k_fill(double *x, int64_t n, unsigned seed) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) { unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; x[i] = (h & 0xffffff) / 16777216.0; }
}
// probe 1: a pure write of y, the product's store shape (8 B per lane, 512 B per wavefront, blocks of 64 rows x 4 lanes)
__global__ __launch_bounds__(256) void k_write(double *y, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)64 + (threadIdx.x & 63);
    if ((threadIdx.x >> 6) == 0 && i < n) y[i] = (double)i;
}
// probe 2: the product without its gathers -- every block streams its 64 x 33 values (16 B per lane) and writes its 64 sums
// (PER = values per row, chosen by the host so that the last block ends inside the array: 32 for the 32.8-entries-per-row headline)
__global__ __launch_bounds__(256) void k_stream_write(const double *__restrict__ val, double *y, int64_t n, int PER) {
    __shared__ double sh[256];
    const double2 *v = reinterpret_cast<const double2 *>(val + (int64_t)blockIdx.x * 64 * PER);
    double a = 0.0;
    for (int q = threadIdx.x; q < 64 * PER / 2; q += 256) { double2 t = v[q]; a += t.x + t.y; }
    sh[threadIdx.x] = a; __syncthreads();
    int64_t i = blockIdx.x * (int64_t)64 + (threadIdx.x & 63);
    if ((threadIdx.x >> 6) == 0 && i < n) y[i] = sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192];
}
static const double *VAL = nullptr;
static int PER = 32;
static double ev_us(void (*launch)(double *), double *y, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(y); launch(y); CK(hipStreamSynchronize(S));
    CK(hipEventRecord(e0, S)); for (int i = 0; i < reps; ++i) launch(y); CK(hipEventRecord(e1, S)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}
static void probes(const char *tag, double *y) {
    double w = ev_us([](double *yy) { k_write<<<(unsigned)((N + 63) / 64), 256, 0, S>>>(yy, N); }, y, 8);
    double sw = ev_us([](double *yy) { k_stream_write<<<(unsigned)((N + 63) / 64), 256, 0, S>>>(VAL, yy, N, PER); }, y, 8);
    printf("probe %-21s pure write of y %.1f us, value stream + write of y %.1f us\n", tag, w, sw);
}
static void fill(double *x, unsigned seed) { k_fill<<<(unsigned)((N + 255) / 256), 256, 0, S>>>(x, N, seed); }

static int FLAVORS = 0;
static double time_pair1(const char *tag, const double *x, double *y);
static double time_pair(const char *tag, const double *x, double *y) {
    if (!FLAVORS) return time_pair1(tag, x, y);
    double first = 0.0;
    for (int f = 0; f < 4; ++f) {       // LAB library only: the y stores plain / non-temporal / sc1 / sc0 sc1, same vectors
        char v[8], t2[96]; snprintf(v, sizeof v, "%d", f); setenv("LCG_HIP_Y_STORE", v, 1);
        snprintf(t2, sizeof t2, "%s/ystore%d", tag, f);
        double us = time_pair1(t2, x, y);
        if (f == 0) first = us;
    }
    setenv("LCG_HIP_Y_STORE", "0", 1);
    return first;
}
static double time_pair1(const char *tag, const double *x, double *y) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int first = dispatch_no;
    for (int i = 0; i < 2; ++i) { lcg_hip_csr_ax(A, x, y, (int)N); ++dispatch_no; }
    CK(hipStreamSynchronize(S));
    CK(hipEventRecord(e0, S));
    for (int i = 0; i < REPS; ++i) { lcg_hip_csr_ax(A, x, y, (int)N); ++dispatch_no; }
    CK(hipEventRecord(e1, S)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / REPS;
    printf("pair %-30s dispatches [%d, %d) us %.1f  x %p y %p\n", tag, first, dispatch_no, us, (const void *)x, (void *)y);
    if (PROBES) probes(tag, y);
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return us;
}

struct Vmm { void *va = nullptr; size_t size = 0; std::vector<hipMemGenericAllocationHandle_t> h; };
static size_t gran_min = 0, gran_rec = 0;

static hipMemAllocationProp prop0() {
    hipMemAllocationProp p; memset(&p, 0, sizeof p);
    p.type = hipMemAllocationTypePinned; p.location.type = hipMemLocationTypeDevice; p.location.id = 0;
    return p;
}
// chunk == 0: one physical handle for the whole size; else one handle per `chunk` bytes
static bool vmm_alloc(Vmm &v, size_t bytes, size_t chunk, size_t va_align) {
    hipMemAllocationProp p = prop0();
    size_t g = chunk ? chunk : gran_rec;
    v.size = (bytes + g - 1) / g * g;
    if (hipMemAddressReserve(&v.va, v.size, va_align, nullptr, 0) != hipSuccess) return false;
    size_t piece = chunk ? chunk : v.size;
    for (size_t off = 0; off < v.size; off += piece) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, piece, &p, 0) != hipSuccess) return false;
        if (hipMemMap((char *)v.va + off, piece, 0, h, 0) != hipSuccess) return false;
        v.h.push_back(h);
    }
    hipMemAccessDesc d; memset(&d, 0, sizeof d);
    d.location.type = hipMemLocationTypeDevice; d.location.id = 0; d.flags = hipMemAccessFlagsProtReadWrite;
    return hipMemSetAccess(v.va, v.size, &d, 1) == hipSuccess;
}

int main(int argc, char **argv) {
    int npairs_a = argc > 1 ? atoi(argv[1]) : 8;
    int phases = argc > 2 ? atoi(argv[2]) : 31;       // bit mask A=1 B=2 C=4 D=8 E=16
    if (argc > 3) N = atoll(argv[3]);
    if (argc > 4) PROBES = atoi(argv[4]);
    if (argc > 5) FLAVORS = atoi(argv[5]);
    if (lcg_hip_init(0)) { fprintf(stderr, "init: %s\n", lcg_hip_last_error()); return 2; }
    S = (hipStream_t)lcg_hip_get_stream();
    if (lcg_hip_csr_generate_ex(&A, N, 16, LCG_HIP_GEN_DIAGONALS, 131072, 1, 1, 0.01, 0, N)) { fprintf(stderr, "generate: %s\n", lcg_hip_last_error()); return 2; }
    hipMemAllocationProp p = prop0();
    CK(hipMemGetAllocationGranularity(&gran_min, &p, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&gran_rec, &p, hipMemAllocationGranularityRecommended));
    const int *rp, *ci; const double *va;
    lcg_hip_csr_arrays(A, &rp, &ci, &va);
    VAL = va;
    PER = (int)(lcg_hip_csr_nnz(A) / ((N + 63) / 64 * 64));       // whole blocks of 64 x PER values stay inside val[nnz]
    if (PER < 1 || (int64_t)((N + 63) / 64) * 64 * PER > lcg_hip_csr_nnz(A)) { fprintf(stderr, "probe shape does not fit\n"); return 2; }
    printf("granularity min %zu recommended %zu; rowptr %p col %p val %p\n", gran_min, gran_rec, (const void *)rp, (const void *)ci, (const void *)va);
    size_t bytes = sizeof(double) * (size_t)N;
    std::vector<double *> ax, ay; std::vector<double> at;
    std::vector<void *> junk;
    if (phases & 1) {
        for (int i = 0; i < npairs_a; ++i) {
            void *j; CK(hipMalloc(&j, (size_t)((i * 53 + 7) * 1031) * 8)); junk.push_back(j);
            double *x, *y; CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); fill(x, 17 + i);
            char tag[64]; snprintf(tag, sizeof tag, "A%d:hipMalloc", i);
            ax.push_back(x); ay.push_back(y); at.push_back(time_pair(tag, x, y));
        }
    }
    std::vector<Vmm> keep;
    double *bx = nullptr, *by = nullptr;
    if (phases & 2) {
        for (int i = 0; i < 6; ++i) {
            Vmm vx, vy;
            if (!vmm_alloc(vx, bytes, 0, 0) || !vmm_alloc(vy, bytes, 0, 0)) { printf("phase B: virtual-memory API failed: %s\n", hipGetErrorString(hipGetLastError())); break; }
            fill((double *)vx.va, 91 + i);
            char tag[64]; snprintf(tag, sizeof tag, "B%d:vmm-one-handle", i);
            time_pair(tag, (double *)vx.va, (double *)vy.va);
            bx = (double *)vx.va; by = (double *)vy.va;
            keep.push_back(vx); keep.push_back(vy);
        }
    }
    if (phases & 4) {
        for (int i = 0; i < 3; ++i) {
            Vmm vx, vy;
            if (!vmm_alloc(vx, bytes, 2u << 20, 0) || !vmm_alloc(vy, bytes, 2u << 20, 0)) { printf("phase C: virtual-memory API failed: %s\n", hipGetErrorString(hipGetLastError())); break; }
            fill((double *)vx.va, 191 + i);
            char tag[64]; snprintf(tag, sizeof tag, "C%d:vmm-2MB-handles", i);
            time_pair(tag, (double *)vx.va, (double *)vy.va);
            keep.push_back(vx); keep.push_back(vy);
        }
    }
    if (phases & 32) {       // one handle per vector, the size rounded up to 2 MB / to 128 MB
        for (int i = 0; i < 4; ++i) {
            Vmm vx, vy;
            const size_t r = i < 2 ? (size_t)2 << 20 : (size_t)128 << 20;
            const size_t b2 = (bytes + r - 1) / r * r;
            if (!vmm_alloc(vx, b2, 0, 0) || !vmm_alloc(vy, b2, 0, 0)) { printf("phase F: virtual-memory API failed\n"); break; }
            fill((double *)vx.va, 391 + i);
            char tag[64]; snprintf(tag, sizeof tag, "F%d:vmm-one-handle-%zuMB", i, b2 >> 20);
            time_pair(tag, (double *)vx.va, (double *)vy.va);
            keep.push_back(vx); keep.push_back(vy);
        }
    }
    if (phases & 64) {       // hipMalloc of 128 MB / 1 GiB, the vector at its front
        for (int i = 0; i < 4; ++i) {
            double *x, *y; const size_t b2 = i < 2 ? (size_t)128 << 20 : (size_t)1 << 30;
            CK(hipMalloc(&x, b2)); CK(hipMalloc(&y, b2)); fill(x, 491 + i);
            char tag[64]; snprintf(tag, sizeof tag, "G%d:hipMalloc-%zuMB", i, b2 >> 20);
            time_pair(tag, x, y);
        }
    }
    double *dx = nullptr, *dy = nullptr;
    if (phases & 8) {
        Vmm arena;
        if (vmm_alloc(arena, (size_t)1 << 30, 0, (size_t)1 << 30)) {
            size_t step = (bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20);
            for (int i = 0; i < 6; ++i) {
                double *x = (double *)((char *)arena.va + (size_t)(2 * i) * step), *y = (double *)((char *)arena.va + (size_t)(2 * i + 1) * step);
                fill(x, 291 + i);
                char tag[64]; snprintf(tag, sizeof tag, "D%d:arena-1GiB-cut", i);
                time_pair(tag, x, y);
                dx = x; dy = y;
            }
            keep.push_back(arena);
        } else printf("phase D: virtual-memory API failed: %s\n", hipGetErrorString(hipGetLastError()));
    }
    if ((phases & 16) && !ax.empty()) {
        // the fastest and the slowest hipMalloc pair, crossed with each other and with the mapped vectors
        int f = 0, s = 0;
        for (size_t i = 0; i < at.size(); ++i) { if (at[i] < at[f]) f = (int)i; if (at[i] > at[s]) s = (int)i; }
        char tag[64];
        snprintf(tag, sizeof tag, "E:xA%d(fast)+yA%d(slow)", f, s); time_pair(tag, ax[f], ay[s]);
        snprintf(tag, sizeof tag, "E:xA%d(slow)+yA%d(fast)", s, f); time_pair(tag, ax[s], ay[f]);
        if (by) { snprintf(tag, sizeof tag, "E:xA%d(slow)+yB", s); time_pair(tag, ax[s], by); snprintf(tag, sizeof tag, "E:xB+yA%d(slow)", s); time_pair(tag, bx, ay[s]); }
        if (dy) { snprintf(tag, sizeof tag, "E:xA%d(slow)+yD", s); time_pair(tag, ax[s], dy); snprintf(tag, sizeof tag, "E:xD+yA%d(slow)", s); time_pair(tag, dx, ay[s]); }
        snprintf(tag, sizeof tag, "E:A%d again", f); time_pair(tag, ax[f], ay[f]);
        snprintf(tag, sizeof tag, "E:A%d again", s); time_pair(tag, ax[s], ay[s]);
    }
    printf("kernel: %s\n", lcg_hip_csr_last_kernel(A));
    return 0;
}
