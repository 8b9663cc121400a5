// gridbar_lab: what does a barrier over a whole resident grid cost on the MI355X, against the kernel boundary it would replace?
// The small systems of BASELINE configs[0] / [1] (case_10K_A: 10^4 rows, 1M-row Laplacian) run CG at two launches per iteration,
// 13 / 36 us: launch-bound.  A solve in ONE launch needs two barriers per iteration (the product's gathers wait for every g; the
// step waits for every partial sum).  Measured here, no library:
//   A  barrier only                     (atomic arrive at agent scope, the last block bumps a generation word the others poll)
//   B  barrier + exchange               (every thread writes a double write-through, after the barrier reads another block's
//                                        with an sc1 load; checked: the values of THIS round arrived)
//   C  the same with plain stores and an agent-scope fence on both sides (L2 write-back + invalidate: eight private L2s)
//   D  the kernel boundary: K dependent launches of a kernel that does the same exchange
// for grids of G blocks x T threads.  Every spin has a clock bound: a barrier that is not met ends the kernel with a flag.
//   hipcc --offload-arch=gfx950 -O2 scripts/gridbar_lab.hip -o scripts/bin/gridbar_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Bar { unsigned cnt; unsigned pad0[31]; unsigned gen; unsigned pad1[31]; int bad; };

__device__ __forceinline__ bool grid_bar(Bar *b, unsigned nblk, unsigned &phase)
{
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        phase++;
        ok = 1;
        const unsigned prev = __hip_atomic_fetch_add(&b->cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == phase * nblk) __hip_atomic_store(&b->gen, phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < phase) {
                if (wall_clock64() - t0 > 200000000LL) { ok = 0; b->bad = 1; break; }       // 2 s at 100 MHz
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
    return ok != 0;
}

template <int MODE>     // 0 barrier only, 1 write-through + sc1 loads, 2 plain + fences
__global__ void k_bars(Bar *b, double *buf, int K, unsigned nblk, int *errs)
{
    unsigned phase = 0;
    const long me = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long other = (long)((blockIdx.x + nblk / 2 + 1) % nblk) * blockDim.x + threadIdx.x;
    int wrong = 0;
    for (int k = 1; k <= K; k++) {
        if (MODE == 1) __hip_atomic_store(buf + me, (double)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) { buf[me] = (double)k; __threadfence(); }
        if (!grid_bar(b, nblk, phase)) return;
        if (MODE == 1) { const double v = __hip_atomic_load(buf + other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (v != (double)k) wrong++; }
        if (MODE == 2) { __threadfence(); const double v = buf[other]; if (v != (double)k) wrong++; }
        if (MODE != 0) { if (!grid_bar(b, nblk, phase)) return; }      // nobody overwrites what another block still reads
    }
    if (wrong) atomicAdd(errs, wrong);
}

__global__ void k_step(double *buf, int k, unsigned nblk, int *errs)
{
    const long me = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long other = (long)((blockIdx.x + nblk / 2 + 1) % nblk) * blockDim.x + threadIdx.x;
    if (k > 1 && buf[other + (long)((k - 1) & 1) * nblk * blockDim.x] != (double)(k - 1)) atomicAdd(errs, 1);
    buf[me + (long)(k & 1) * nblk * blockDim.x] = (double)k;
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 2000;
    hipStream_t s; CK(hipStreamCreate(&s));
    Bar *b; int *errs; double *buf;
    CK(hipMalloc(&b, sizeof(Bar))); CK(hipMalloc(&errs, 4)); CK(hipMalloc(&buf, (size_t)2 * 1024 * 1024 * 8 * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int shapes[][2] = {{10, 1024}, {40, 256}, {64, 256}, {256, 256}, {256, 1024}, {512, 256}, {512, 512}, {1024, 256}};
    printf("K = %d rounds; us per round (a round of B / C is TWO barriers and one exchange; of D one launch)\n", K);
    for (auto &sh : shapes) {
        const unsigned G = sh[0], T = sh[1];
        int nb = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_bars<1>, T, 0));
        hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
        if ((long)nb * p.multiProcessorCount < (long)G) { printf("G %u x T %u: not resident (%d per CU)\n", G, T, nb); continue; }
        double us[4] = {0, 0, 0, 0}; int bad[4] = {0, 0, 0, 0};
        for (int mode = 0; mode < 4; mode++) {
            CK(hipMemsetAsync(b, 0, sizeof(Bar), s)); CK(hipMemsetAsync(errs, 0, 4, s));
            CK(hipMemsetAsync(buf, 0, (size_t)2 * G * T * 8, s));
            for (int rep = 0; rep < 2; rep++) {
                if (rep) { CK(hipMemsetAsync(b, 0, sizeof(Bar), s)); CK(hipEventRecord(e0, s)); }
                int k = K; unsigned g = G; void *args[] = {&b, &buf, &k, &g, &errs};
                if (mode == 0) CK(hipLaunchCooperativeKernel((void *)k_bars<0>, dim3(G), dim3(T), args, 0, s));
                if (mode == 1) CK(hipLaunchCooperativeKernel((void *)k_bars<1>, dim3(G), dim3(T), args, 0, s));
                if (mode == 2) CK(hipLaunchCooperativeKernel((void *)k_bars<2>, dim3(G), dim3(T), args, 0, s));
                if (mode == 3) for (int q = 1; q <= K; q++) hipLaunchKernelGGL(k_step, dim3(G), dim3(T), 0, s, buf, q, G, errs);
            }
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            us[mode] = ms * 1e3 / K;
            int h[2] = {0, 0}; Bar hb;
            CK(hipMemcpy(h, errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost));
            bad[mode] = h[0] + 1000000 * hb.bad;
        }
        printf("G %4u x T %4u: A barrier %6.2f | B write-through exchange %6.2f | C fenced exchange %6.2f | D launches %6.2f   (wrong values / timeouts: %d %d %d %d)\n",
               G, T, us[0], us[1], us[2], us[3], bad[0], bad[1], bad[2], bad[3]);
        fflush(stdout);
    }
    return 0;
}
