// placement_lab7: the MAP of the groups.  Lab 6 showed that a product-shaped kernel (stream 512 MB of R, write 15 MB of W) is 6 % slower
// when R and W belong to the same group of allocations, and that a whole 4 GiB allocation is one group.  Here: NB allocations of SZ MiB
// each, taken one after the other (the driver hands out memory top-down, so consecutive allocations are neighbours in memory), every
// one classified against a growing list of representatives (a buffer that is fast against all of them founds a new group).  The
// sequence of groups along the allocations shows how large a group is and whether the pattern repeats.
//   hipcc --offload-arch=gfx950 -O2 scripts/placement_lab7.hip -o scripts/bin/placement_lab7
//   placement_lab7 [NB = 48] [SZ MiB = 1024] [fine = 0: after the map, walk ONE buffer pair in steps of `fine` MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
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

int main(int argc, char **argv)
{
    const int NB = argc > 1 ? atoi(argv[1]) : 48;
    const size_t SZ = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
    const int fine = argc > 3 ? atoi(argv[3]) : 0;
    const size_t RB = std::min<size_t>((size_t)512 << 20, SZ / 2);          // read: the first half (<= 512 MB); written: 16 MB at 3/4
    const long nblk = (long)(RB / 8 / (64 * 33));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<double *> B(NB);
    for (int i = 0; i < NB; i++) { CK(hipMalloc(&B[i], SZ)); CK(hipMemsetAsync(B[i], 0, SZ, s)); }
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
    auto wpart = [&](int j) { return B[j] + (SZ / 4 * 3) / 8; };
    printf("%d allocations of %zu MiB; read %zu MB, %ld blocks\n", NB, SZ >> 20, RB >> 20, nblk);
    for (int i = 0; i < NB; i++) printf("buf %2d  %p\n", i, (void *)B[i]);
    // self pairs first: the level of "same group" on this box, then everything against the representatives
    std::vector<int> rep; std::vector<int> group(NB, -1);
    double self = 0; for (int i = 0; i < 3; i++) self += t(B[i], wpart(i)) / 3;
    printf("same allocation read and written: %.1f us\n", self);
    for (int j = 0; j < NB; j++) {
        printf("buf %2d:", j);
        int g = -1; double best = 1e9;
        std::vector<double> us;
        for (int r : rep) { us.push_back(t(B[r], wpart(j))); best = std::min(best, us.back()); }
        for (size_t k = 0; k < rep.size(); k++) { printf(" %5.1f", us[k]); }
        // slow against representative k (>= 3.5 % above the fastest pairing of this buffer, or -- a single representative -- near `self`)
        for (size_t k = 0; k < rep.size(); k++)
            if ((rep.size() > 1 && us[k] > best * 1.035) || (rep.size() == 1 && us[k] > self * 0.98)) g = (int)k;
        if (g < 0) { rep.push_back(j); g = (int)rep.size() - 1; printf("  -> founds group %d", g); }
        else printf("  -> group %d", g);
        group[j] = g;
        printf("\n"); fflush(stdout);
    }
    printf("map: "); for (int j = 0; j < NB; j++) printf("%c", 'A' + group[j]); printf("\n");
    if (fine > 0 && rep.size() >= 2) {
        // walk the written position through buffer rep[0] (read: rep[0] itself -> expect slow everywhere) and through a buffer of another
        // group in steps of `fine` MiB: does the class ever change inside one allocation?
        for (int which = 0; which < 2; which++) {
            const int b = rep[which];
            printf("written position walks through buf %d (read buf %d), step %d MiB:", b, rep[0], fine);
            for (size_t off = RB; off + ((size_t)16 << 20) <= SZ; off += (size_t)fine << 20) printf(" %5.1f", t(B[rep[0]], B[b] + off / 8));
            printf("\n");
        }
    }
    return 0;
}
