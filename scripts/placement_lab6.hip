// placement_lab6: is the slow class a property of the written buffer alone, or of the PAIR (stream that is read, buffer that is
// written)?  No library: NR read buffers (512 MB each), NW write buffers (80 MB each), all hipMalloc; the kernel is the product's memory
// shape without its gathers -- block b streams 64 x 33 doubles of R (16 B per lane) and writes 64 doubles of W, 31,000 blocks -- timed
// for every (R, W) pair, twice.  Then the same with the roles of a few buffers swapped (W buffers read, R buffers written).
//   hipcc --offload-arch=gfx950 -O2 scripts/placement_lab6.hip -o scripts/bin/placement_lab6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ __launch_bounds__(256) void k_stream_write(const double *__restrict__ r, double *__restrict__ w, long nblk)
{
    __shared__ double sh[256];
    const double2 *v = reinterpret_cast<const double2 *>(r + (long)blockIdx.x * 64 * 33);
    double a = 0.0;
    for (int q = threadIdx.x; q < 64 * 33 / 2; q += 256) { double2 t = v[q]; a += t.x + t.y; }
    sh[threadIdx.x] = a; __syncthreads();
    if (threadIdx.x < 64) w[(long)blockIdx.x * 64 + threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192];
}

// write n doubles / read them back (sum): is what was just written served by the Infinity Cache when it is read again?
__global__ __launch_bounds__(256) void k_wr(double *w, long n) { long i = (blockIdx.x * 256L + threadIdx.x) * 2; if (i + 1 < n) *reinterpret_cast<double2 *>(w + i) = make_double2(1.0, 2.0); }
__global__ __launch_bounds__(256) void k_rd(const double *w, long n, double *out)
{
    double a = 0.0;
    for (long i = (blockIdx.x * 256L + threadIdx.x) * 2; i + 1 < n; i += (long)gridDim.x * 512) { double2 t = *reinterpret_cast<const double2 *>(w + i); a += t.x + t.y; }
    if (a == 12345.678) out[0] = a;
}

int main(int argc, char **argv)
{
    const int NR = argc > 1 ? atoi(argv[1]) : 6, NW = argc > 2 ? atoi(argv[2]) : 16;
    const size_t RB = (size_t)512 << 20, WB = 80000000;
    const long nblk = (long)(RB / 8 / (64 * 33));                  // blocks whose 64 x 33 doubles lie inside an R buffer
    if ((size_t)nblk * 64 * 8 > WB) { fprintf(stderr, "shape\n"); return 2; }
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<double *> R(NR), W(NW);
    for (int i = 0; i < NR; i++) { CK(hipMalloc(&R[i], RB)); CK(hipMemsetAsync(R[i], 0, RB, s)); }
    for (int j = 0; j < NW; j++) { void *junk; CK(hipMalloc(&junk, (size_t)(j * 37 + 5) * 4099 * 8)); CK(hipMalloc(&W[j], WB)); CK(hipMemsetAsync(W[j], 0, WB, s)); }
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto t = [&](const double *r, double *w) {
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_stream_write, dim3((unsigned)nblk), dim3(256), 0, s, r, w, nblk);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 6; i++) hipLaunchKernelGGL(k_stream_write, dim3((unsigned)nblk), dim3(256), 0, s, r, w, nblk);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / 6;
    };
    printf("%ld blocks: %zu MB read, %ld MB written per launch\n", nblk, RB >> 20, nblk * 512 >> 20);
    for (int rep = 0; rep < 2; rep++) {
        printf("rows = read buffer, columns = written buffer (us)%s\n", rep ? ", again" : "");
        for (int i = 0; i < NR; i++) {
            printf("R%-2d", i);
            for (int j = 0; j < NW; j++) printf(" %5.1f", t(R[i], W[j]));
            printf("\n"); fflush(stdout);
        }
    }
    printf("roles swapped: rows = W buffers read (their first 80 MB: fewer blocks), columns = R buffers written\n");
    const long nblk2 = (long)(WB / 8 / (64 * 33));
    auto t2 = [&](const double *r, double *w) {
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k_stream_write, dim3((unsigned)nblk2), dim3(256), 0, s, r, w, nblk2);
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k_stream_write, dim3((unsigned)nblk2), dim3(256), 0, s, r, w, nblk2);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3 / 20;
    };
    for (int j = 0; j < (NW < 8 ? NW : 8); j++) {
        printf("W%-2d", j);
        for (int i = 0; i < NR; i++) printf(" %5.1f", t2(W[j], R[i]));
        printf("\n");
    }
    {   // the classes again (R0), then: 64 MB written, read back, read again -- per written buffer, plus buffers from hipExtMallocWithFlags
        std::vector<double *> X = W; std::vector<const char *> tag(NW, "hipMalloc");
        struct { unsigned flag; const char *name; } fl[] = {{hipDeviceMallocUncached, "uncached"}, {hipDeviceMallocFinegrained, "fine-grained"}, {hipDeviceMallocContiguous, "contiguous"}};
        for (auto &f : fl)
            for (int k = 0; k < 3; k++) {
                double *p = nullptr;
                if (hipExtMallocWithFlags((void **)&p, WB, f.flag) == hipSuccess && p) { CK(hipMemsetAsync(p, 0, WB, s)); X.push_back(p); tag.push_back(f.name); }
                else { (void)hipGetLastError(); printf("hipExtMallocWithFlags(%s) failed\n", f.name); }
            }
        CK(hipStreamSynchronize(s));
        double *out; CK(hipMalloc(&out, 64));
        const long n64 = (64L << 20) / 8;
        double *flush; CK(hipMalloc(&flush, (size_t)1 << 30));
        auto ev = [&](auto &&launch) { CK(hipEventRecord(e0, s)); launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3; };
        printf("per written buffer: stream + write (us) | write 64 MB | read it back | read again | read after a 1 GiB flush\n");
        for (size_t j = 0; j < X.size(); j++) {
            const double sw = t(R[0], X[j]);
            CK(hipMemsetAsync(flush, 0, (size_t)1 << 30, s));
            double a = 0, b = 0, c2 = 0, d = 0;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemsetAsync(flush, 0, (size_t)1 << 30, s));
                a = ev([&] { hipLaunchKernelGGL(k_wr, dim3((unsigned)(n64 / 512)), dim3(256), 0, s, X[j], n64); });
                b = ev([&] { hipLaunchKernelGGL(k_rd, dim3(2048), dim3(256), 0, s, X[j], n64, out); });
                c2 = ev([&] { hipLaunchKernelGGL(k_rd, dim3(2048), dim3(256), 0, s, X[j], n64, out); });
                CK(hipMemsetAsync(flush, 0, (size_t)1 << 30, s));
                d = ev([&] { hipLaunchKernelGGL(k_rd, dim3(2048), dim3(256), 0, s, X[j], n64, out); });
            }
            printf("  %-2zu %-12s %6.1f | %6.1f | %6.1f | %6.1f | %6.1f\n", j, tag[j], sw, a, b, c2, d);
        }
    }
    // a written buffer walked through one 1 GiB allocation in steps of 80 MB: where inside it does the class change?
    double *big; CK(hipMalloc(&big, (size_t)4 << 30)); CK(hipMemsetAsync(big, 0, (size_t)4 << 30, s)); CK(hipStreamSynchronize(s));
    printf("written buffer = offsets of 64 MB inside one 4 GiB allocation, read buffer R0:\n");
    for (size_t off = 0; off + WB <= ((size_t)4 << 30); off += (size_t)64 << 20) printf(" %5.1f", t(R[0], big + off / 8));
    printf("\n");
    return 0;
}
