// vmm_lab: what does a 1 GiB chunk cost through the virtual-memory API (hipMemCreate + hipMemMap + hipMemSetAccess) against hipMalloc,
// once the device's memory has been in use?  The placement walk (driver.hpp) looks for a vector's place by allocating 1 GiB chunks; out
// of memory a process has released before, ONE hipMalloc of 1 GiB costs 30 ms .. 0.5 s (the driver clears what it hands out), which
// ends the walk after a few chunks on a box that is not fresh.  If physical chunks came cheaper through hipMemCreate, the walk could
// reach the 96th chunk anywhere.
//   1. dirty: DIRTY chunks of 1 GiB allocated, written, freed;  2. N chunks by hipMalloc, ms per call;  3. N chunks by the VMM calls.
//   hipcc --offload-arch=gfx950 -O2 scripts/vmm_lab.hip -o scripts/bin/vmm_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_touch(double *p, long n) { long i = (long)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1.0; }

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int DIRTY = argc > 1 ? atoi(argv[1]) : 48, N = argc > 2 ? atoi(argv[2]) : 40;
    const size_t CH = (size_t)1 << 30;
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f of %.1f GiB\n", fr / 1073741824.0, tot / 1073741824.0);
    auto report = [](const char *what, const std::vector<double> &ms) {
        double sum = 0, mx = 0; int slow = 0;
        for (double m : ms) { sum += m; if (m > mx) mx = m; if (m > 5.0) slow++; }
        printf("%-34s %zu chunks: %8.2f ms in all, the slowest %7.2f ms, %d over 5 ms; first eight:", what, ms.size(), sum, mx, slow);
        for (size_t i = 0; i < ms.size() && i < 8; i++) printf(" %.2f", ms[i]);
        printf("\n"); fflush(stdout);
    };
    for (int round = 0; round < 2; round++) {
        {   // fresh (round 0) or after the dirtying below (round 1)
            std::vector<double *> p(N); std::vector<double> ms;
            for (int i = 0; i < N; i++) { const double t = now_ms(); CK(hipMalloc(&p[i], CH)); ms.push_back(now_ms() - t); }
            report(round ? "hipMalloc, after dirtying" : "hipMalloc, as found", ms);
            for (auto q : p) CK(hipFree(q));
        }
        {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            std::vector<hipMemGenericAllocationHandle_t> h(N); std::vector<void *> va(N); std::vector<double> ms, ms_create;
            for (int i = 0; i < N; i++) {
                const double t = now_ms();
                CK(hipMemCreate(&h[i], CH, &prop, 0));
                const double t1 = now_ms();
                CK(hipMemAddressReserve(&va[i], CH, gran, nullptr, 0));
                CK(hipMemMap(va[i], CH, 0, h[i], 0));
                CK(hipMemSetAccess(va[i], CH, &acc, 1));
                ms.push_back(now_ms() - t); ms_create.push_back(t1 - t);
            }
            report(round ? "VMM create+map, after dirtying" : "VMM create+map, as found", ms);
            report("  of which hipMemCreate", ms_create);
            hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, (double *)va[N - 1], 262144L); CK(hipDeviceSynchronize());
            double v = 0; CK(hipMemcpy(&v, va[N - 1], 8, hipMemcpyDeviceToHost));
            printf("  (granularity %zu, a kernel wrote %.1f into the last chunk)\n", gran, v);
            for (int i = 0; i < N; i++) { CK(hipMemUnmap(va[i], CH)); CK(hipMemRelease(h[i])); CK(hipMemAddressFree(va[i], CH)); }
        }
        if (round == 0) {
            std::vector<double *> p(DIRTY);
            for (int i = 0; i < DIRTY; i++) { CK(hipMalloc(&p[i], CH)); hipLaunchKernelGGL(k_touch, dim3((unsigned)(CH / 8 / 256)), dim3(256), 0, 0, p[i], (long)(CH / 8)); }
            CK(hipDeviceSynchronize());
            for (auto q : p) CK(hipFree(q));
            printf("dirtied: %d chunks allocated, written and freed\n", DIRTY);
        }
    }
    return 0;
}
