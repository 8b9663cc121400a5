// fake_rccl.hip -- TEST INFRASTRUCTURE, never shipped and never loaded by the product unless LCG_HIP_RCCL_LIB points at it.
//
// A stand-in for the eleven librccl entry points liblcg_hip.so binds by dlsym (liblcg_amd/csrc/comm.hip: load_rccl), so that the
// north-star exchange -- ncclAllGather of the x slices, ncclAllReduce of the dots, ncclReduceScatter of op(A).x, grouped
// ncclSend/ncclRecv of the neighbour ranges -- runs with SEVERAL RANKS ON ONE GPU.  The real RCCL refuses two ranks on one device,
// and a development box of this pool has one GPU: without this file the first execution of those code paths with P > 1 would be
// the 8-GPU measurement itself.
//
// What it keeps of the real calls (the properties the library's code depends on):
//   * stream order: every call only ENQUEUES work on the caller's stream and returns; data is moved by copies and kernels on that
//     stream, the ranks meet in one-block kernels that sit in the stream between them (each rank writes its arrival word into every
//     peer's control block -- uncached device memory mapped through HIP IPC -- and polls its own; the first form, host functions
//     meeting in host shared memory, stalled once in a few hundred solves with nobody waiting: the runtime's callback thread);
//   * the same call sequence on every rank of a communicator: collectives carry a signature (kind, count, type, sequence number)
//     that the ranks compare when they meet -- a mismatch is reported and aborts the communicator (the real library would hang);
//   * in-place all-gather (sendbuff == recvbuff + rank * count), all-reduce in place;
//   * point-to-point calls synchronise only the pairs involved (an empty group is no operation at all);
//   * a peer that never arrives ends the wait after FAKE_RCCL_TIMEOUT_S (30) seconds (the device's wall clock): the communicator is
//     aborted on every rank, every later call returns ncclRemoteError (the real library leaves that to its watchdog / the launcher).
// What it does not model: links, rings, trees, channels, CU occupancy of the collective kernels, and bandwidth.  Nothing measured
// through it is a performance number.
//
// Transport: every rank owns one staging buffer in device memory (2 x FAKE_RCCL_STAGING_MB, 256: one area for the collectives,
// one for point-to-point messages), exported by HIP IPC through a
// POSIX shared-memory block named in the ncclUniqueId (used at init and destroy only).  A sender copies into its own staging buffer;
// receivers copy out of the senders' (mapped) buffers; a second meeting releases the buffer.  Sums are added in rank order
// (identical bits on every rank).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int MAXP = 16;
constexpr uint32_t MAGIC = 0x46524343u;     // "FRCC"

// control block of one rank: uncached device memory, written by the peers (one writer per word), polled by the owner
struct Ctl {
    unsigned long long arrive[2][MAXP];     // [phase][src]: number of the last collective src has reached (+1)
    unsigned long long sig[MAXP];           // [src]: signature of src's current collective
    unsigned long long sent[MAXP];          // [src]: messages src has staged for me
    unsigned long long consumed[MAXP];      // [dst]: messages of mine dst has copied out
    unsigned long long abort_flag;          // anybody: the communicator is dead
};

struct Shm {
    std::atomic<uint32_t> magic;
    std::atomic<int> attached, detached, abort_flag;
    std::atomic<int> handle_ready[MAXP];
    hipIpcMemHandle_t staging[MAXP], ctl[MAXP];
    uint64_t staging_bytes[MAXP];
};

struct Pending { bool send; const void *sbuf; void *rbuf; size_t bytes; int peer; };

}  // namespace

struct ncclComm {
    Shm *shm = nullptr;
    int P = 1, me = 0;
    char *staging = nullptr;
    size_t staging_bytes = 0;
    char *peer[MAXP] = {nullptr};
    Ctl *ctl = nullptr, *peer_ctl[MAXP] = {nullptr};
    int *status = nullptr, *status_dev = nullptr;      // host-mapped: 0 fine, 1 time-out, 2 call sequence mismatch, 3 a peer aborted; [1..3]: details
    std::vector<void *> opened;
    uint64_t seq = 0;                   // collectives enqueued so far (same on every rank)
    uint64_t nsent[MAXP] = {0}, nrecv[MAXP] = {0};
    std::atomic<int> async_err{0};
    double timeout_s = 30.0;
    bool verbose = false;
    uint64_t calls[6] = {0};            // all-gather, all-reduce, reduce-scatter, send, recv, groups
};

namespace {

thread_local int g_group_depth = 0;
thread_local std::vector<Pending> g_pending;
thread_local ncclComm *g_group_comm = nullptr;
thread_local hipStream_t g_group_stream = nullptr;

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: case ncclFloat8e4m3: case ncclFloat8e5m2: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

void comm_abort(ncclComm *c, const char *why)
{
    if (!c->async_err.exchange((int)ncclRemoteError))
        std::fprintf(stderr, "[fake_rccl rank %d/%d] communicator aborted: %s\n", c->me, c->P, why);
    c->shm->abort_flag.store(1);
}

// spin until pred() or abort / time-out; true = pred held
template <class F>
bool wait_until(ncclComm *c, F pred, const char *what)
{
    const double t0 = now_s();
    unsigned spins = 0;
    while (!pred()) {
        if (c->shm->abort_flag.load(std::memory_order_relaxed)) { comm_abort(c, "a peer aborted"); return false; }
        if ((++spins & 0x3ff) == 0) {
            if (now_s() - t0 > c->timeout_s) {
                char buf[200];
                std::snprintf(buf, sizeof buf, "%s: no answer within %.0f s (a peer died or fell out of step)", what, c->timeout_s);
                comm_abort(c, buf);
                return false;
            }
            usleep(50);
        } else {
            sched_yield();
        }
    }
    return true;
}

// ---- meetings on the device -------------------------------------------------------------------------
struct SyncArgs { Ctl *mine; Ctl *peer[MAXP]; int P, me; long long timeout_ticks; int *status; };

__device__ __forceinline__ void word_store(unsigned long long *w, unsigned long long v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned long long word_load(const unsigned long long *w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// poll one word of my own control block; false = gave up (time-out or abort), with the reason in status[]
__device__ bool spin_ge(const SyncArgs &a, const unsigned long long *w, unsigned long long want, int code_detail)
{
    const long long t0 = wall_clock64();
    for (;;) {
        if (word_load(w) >= want) return true;
        if (word_load(&a.mine->abort_flag) != 0) { if (atomicCAS(a.status, 0, 3) == 0) a.status[1] = code_detail; return false; }
        if (wall_clock64() - t0 > a.timeout_ticks) {
            if (atomicCAS(a.status, 0, 1) == 0) a.status[1] = code_detail;
            for (int q = 0; q < a.P; q++) word_store(&a.peer[q]->abort_flag, 1ull);
            return false;
        }
        __builtin_amdgcn_s_sleep(16);
    }
}

// all ranks meet: collective number seq, phase 0 (data staged) or 1 (data read); phase 0 also compares the call signatures
__global__ void k_meet(SyncArgs a, unsigned long long seq, int phase, unsigned long long sig)
{
    const int q = threadIdx.x;
    if (q >= a.P || *a.status != 0) return;
    if (phase == 0) word_store(&a.peer[q]->sig[a.me], sig);
    __atomic_thread_fence(__ATOMIC_RELEASE);        // (what this stream wrote before -- the staging buffer -- is visible before the arrival is)
    word_store(&a.peer[q]->arrive[phase][a.me], seq + 1);
    const bool ok = spin_ge(a, &a.mine->arrive[phase][q], seq + 1, (int)(seq & 0x3fffffff));
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (ok && phase == 0 && word_load(&a.mine->sig[q]) != sig) {
        if (atomicCAS(a.status, 0, 2) == 0) { a.status[1] = (int)(seq & 0x3fffffff); a.status[2] = q; }
        for (int r = 0; r < a.P; r++) word_store(&a.peer[r]->abort_flag, 1ull);
    }
}

struct PairArgs { int n; int peer[MAXP]; unsigned long long want[MAXP]; int kind; };     // kind: 0 wait consumed, 1 post sent, 2 wait sent, 3 post consumed

__global__ void k_pair(SyncArgs a, PairArgs w)
{
    const int i = threadIdx.x;
    if (i >= w.n || *a.status != 0) return;
    const int q = w.peer[i];
    switch (w.kind) {
    case 0: (void)spin_ge(a, &a.mine->consumed[q], w.want[i], -1 - q); break;
    case 1: __atomic_thread_fence(__ATOMIC_RELEASE); word_store(&a.peer[q]->sent[a.me], w.want[i]); break;
    case 2: (void)spin_ge(a, &a.mine->sent[q], w.want[i], -100 - q); __atomic_thread_fence(__ATOMIC_ACQUIRE); break;
    default: __atomic_thread_fence(__ATOMIC_RELEASE); word_store(&a.peer[q]->consumed[a.me], w.want[i]); break;
    }
}

SyncArgs sync_args(ncclComm *c)
{
    SyncArgs a;
    a.mine = c->ctl; a.P = c->P; a.me = c->me; a.status = c->status_dev;
    a.timeout_ticks = (long long)(c->timeout_s * 1e8);      // wall_clock64: 100 MHz
    for (int q = 0; q < MAXP; q++) a.peer[q] = q < c->P ? c->peer_ctl[q] : nullptr;
    return a;
}

// what a kernel of an earlier call left in the status word (asynchronous, like the real library's async error)
bool failed(ncclComm *c)
{
    if (c->async_err.load()) return true;
    const int st = c->status ? *(volatile int *)c->status : 0;
    if (st == 0 && !c->shm->abort_flag.load()) return false;
    if (!c->async_err.exchange((int)ncclRemoteError)) {
        const int d = c->status ? c->status[1] : 0;
        if (st == 1 && d >= 0) std::fprintf(stderr, "[fake_rccl rank %d/%d] communicator aborted: collective #%d: a peer did not arrive within %.0f s (it died or fell out of step)\n", c->me, c->P, d, c->timeout_s);
        else if (st == 1) std::fprintf(stderr, "[fake_rccl rank %d/%d] communicator aborted: %s peer %d: nothing within %.0f s\n", c->me, c->P, d > -100 ? "ncclSend, slot still unread by" : "ncclRecv, nothing sent by", d > -100 ? -1 - d : -100 - d, c->timeout_s);
        else if (st == 2) std::fprintf(stderr, "[fake_rccl rank %d/%d] communicator aborted: collective #%d: rank %d issued another call (count, type or kind differ)\n", c->me, c->P, d, c->status[2]);
        else std::fprintf(stderr, "[fake_rccl rank %d/%d] communicator aborted: a peer aborted\n", c->me, c->P);
    }
    c->shm->abort_flag.store(1);
    return true;
}

ncclResult_t enqueue_meet(ncclComm *c, hipStream_t s, uint64_t seq, int phase, uint64_t sig, const char *)
{
    hipLaunchKernelGGL(k_meet, dim3(1), dim3(64), 0, s, sync_args(c), (unsigned long long)seq, phase, (unsigned long long)sig);
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

uint64_t signature(int kind, size_t count, ncclDataType_t t, uint64_t seq)
{
    uint64_t h = 1469598103934665603ull;
    for (uint64_t v : {(uint64_t)kind, (uint64_t)count, (uint64_t)t, seq}) { h ^= v; h *= 1099511628211ull; }
    return h;
}

struct Peers { const char *p[MAXP]; };

// out[i] = sum over ranks q (in rank order) of in_q[base + i]
template <class T>
__global__ void k_sum(Peers peers, int P, size_t base, T *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T acc = reinterpret_cast<const T *>(peers.p[0])[base + i];
        for (int q = 1; q < P; q++) acc += reinterpret_cast<const T *>(peers.p[q])[base + i];
        out[i] = acc;
    }
}

ncclResult_t launch_sum(ncclComm *c, ncclDataType_t t, size_t base, void *out, size_t n, hipStream_t s)
{
    Peers pr;
    for (int q = 0; q < MAXP; q++) pr.p[q] = q < c->P ? c->peer[q] : nullptr;
    const unsigned g = (unsigned)std::min<size_t>(1024, (n + 255) / 256);
    if (n == 0) return ncclSuccess;
    switch (t) {
    case ncclFloat64: hipLaunchKernelGGL((k_sum<double>), dim3(g), dim3(256), 0, s, pr, c->P, base, static_cast<double *>(out), n); break;
    case ncclFloat32: hipLaunchKernelGGL((k_sum<float>), dim3(g), dim3(256), 0, s, pr, c->P, base, static_cast<float *>(out), n); break;
    case ncclInt64: hipLaunchKernelGGL((k_sum<long long>), dim3(g), dim3(256), 0, s, pr, c->P, base, static_cast<long long *>(out), n); break;
    case ncclUint64: hipLaunchKernelGGL((k_sum<unsigned long long>), dim3(g), dim3(256), 0, s, pr, c->P, base, static_cast<unsigned long long *>(out), n); break;
    case ncclInt32: hipLaunchKernelGGL((k_sum<int>), dim3(g), dim3(256), 0, s, pr, c->P, base, static_cast<int *>(out), n); break;
    default: return ncclInvalidArgument;
    }
    return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t check(ncclComm *c)
{
    if (!c || !c->shm) return ncclInvalidArgument;
    if (failed(c)) return ncclRemoteError;
    if (g_group_depth > 0) return ncclInvalidUsage;     // collectives inside a group: not modelled (the library makes none)
    return ncclSuccess;
}

#define HIPOK(x) do { if ((x) != hipSuccess) return ncclUnhandledCudaError; } while (0)
#define NCCLOK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return r_; } while (0)

// ---- point to point ----------------------------------------------------------------------------------
ncclResult_t run_group(ncclComm *c, hipStream_t s, std::vector<Pending> &ops)
{
    if (ops.empty()) return ncclSuccess;
    if (failed(c)) return ncclRemoteError;
    const size_t slot = c->staging_bytes / (size_t)c->P / 256 * 256;
    bool seen_s[MAXP] = {false}, seen_r[MAXP] = {false};
    for (const Pending &o : ops) {
        if (o.peer < 0 || o.peer >= c->P) return ncclInvalidArgument;
        bool *seen = o.send ? seen_s : seen_r;
        if (seen[o.peer]) { std::fprintf(stderr, "[fake_rccl] two %s calls for peer %d in one group: not modelled\n", o.send ? "send" : "recv", o.peer); return ncclInvalidUsage; }
        seen[o.peer] = true;
        if (o.bytes > slot) {
            std::fprintf(stderr, "[fake_rccl] message of %zu bytes exceeds the %zu-byte slot: raise FAKE_RCCL_STAGING_MB\n", o.bytes, slot);
            return ncclInvalidArgument;
        }
    }
    auto stage = [&](int kind, bool sends) -> ncclResult_t {
        PairArgs w;
        w.n = 0; w.kind = kind;
        for (const Pending &o : ops)
            if (o.send == sends) {
                w.peer[w.n] = o.peer;
                const uint64_t k = sends ? c->nsent[o.peer] : c->nrecv[o.peer];
                w.want[w.n] = (kind == 0) ? k : k + 1;
                w.n++;
            }
        if (w.n == 0) return ncclSuccess;
        hipLaunchKernelGGL(k_pair, dim3(1), dim3(64), 0, s, sync_args(c), w);
        return hipGetLastError() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
    };
    // sends: my slot for each destination must have been read (previous message), then copy in, then announce
    NCCLOK(stage(0, true));
    for (const Pending &o : ops)
        if (o.send && o.bytes) {
            if (o.peer == c->me) continue;      // self-send: matched with the self-recv below, copied directly
            HIPOK(hipMemcpyAsync(c->staging + c->staging_bytes + slot * (size_t)o.peer, o.sbuf, o.bytes, hipMemcpyDeviceToDevice, s));
        }
    NCCLOK(stage(1, true));
    // receives: wait for the senders' announcements, copy out, release their slots
    NCCLOK(stage(2, false));
    for (const Pending &o : ops)
        if (!o.send && o.bytes) {
            if (o.peer == c->me) {
                for (const Pending &q : ops)
                    if (q.send && q.peer == c->me) HIPOK(hipMemcpyAsync(o.rbuf, q.sbuf, o.bytes, hipMemcpyDeviceToDevice, s));
                continue;
            }
            HIPOK(hipMemcpyAsync(o.rbuf, c->peer[o.peer] + c->staging_bytes + slot * (size_t)c->me, o.bytes, hipMemcpyDeviceToDevice, s));
        }
    NCCLOK(stage(3, false));
    for (const Pending &o : ops) { if (o.send) c->nsent[o.peer]++; else c->nrecv[o.peer]++; }
    return ncclSuccess;
}

ncclResult_t p2p_call(bool send, const void *sbuf, void *rbuf, size_t count, ncclDataType_t t, int peer, ncclComm *c, hipStream_t s)
{
    if (!c || !c->shm) return ncclInvalidArgument;
    const size_t ts = type_size(t);
    if (!ts) return ncclInvalidArgument;
    c->calls[send ? 3 : 4]++;
    Pending o{send, sbuf, rbuf, count * ts, peer};
    if (g_group_depth > 0) {
        if (g_group_comm && g_group_comm != c) return ncclInvalidUsage;
        if (!g_pending.empty() && g_group_stream != s) return ncclInvalidUsage;
        g_group_comm = c; g_group_stream = s;
        g_pending.push_back(o);
        return ncclSuccess;
    }
    std::vector<Pending> one{o};
    return run_group(c, s, one);
}

}  // namespace

extern "C" {

// marker: lets a test assert which library the product bound
const char *fakeRcclInfo(void) { return "fake_rccl: several ranks per GPU over HIP IPC staging buffers and peer-written control words (tests only)"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    std::memset(id, 0, sizeof *id);
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    std::snprintf(id->internal, sizeof id->internal, "/fakerccl-%d-%lx%lx", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) { perror("[fake_rccl] shm_open"); return ncclSystemError; }
    if (ftruncate(fd, sizeof(Shm)) != 0) { perror("[fake_rccl] ftruncate"); close(fd); shm_unlink(id->internal); return ncclSystemError; }
    close(fd);      // a fresh segment is zero-filled: every counter starts at 0
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > MAXP || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    id.internal[sizeof id.internal - 1] = 0;
    if (std::strncmp(id.internal, "/fakerccl-", 10) != 0) { std::fprintf(stderr, "[fake_rccl] this unique id was not made by ncclGetUniqueId of this library\n"); return ncclInvalidArgument; }
    const int fd = shm_open(id.internal, O_RDWR, 0600);
    if (fd < 0) { perror("[fake_rccl] shm_open (init)"); return ncclSystemError; }
    void *m = mmap(nullptr, sizeof(Shm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { perror("[fake_rccl] mmap"); return ncclSystemError; }
    ncclComm *c = new ncclComm();
    c->shm = static_cast<Shm *>(m);
    c->P = nranks; c->me = rank;
    if (const char *e = std::getenv("FAKE_RCCL_TIMEOUT_S")) c->timeout_s = std::max(0.5, atof(e));
    c->verbose = std::getenv("FAKE_RCCL_VERBOSE") != nullptr;
    size_t mb = 256;
    if (const char *e = std::getenv("FAKE_RCCL_STAGING_MB")) mb = (size_t)std::max(1L, atol(e));
    c->staging_bytes = mb << 20;
    auto bail = [&](ncclResult_t r, const char *why) {
        std::fprintf(stderr, "[fake_rccl rank %d] init failed: %s\n", rank, why);
        c->shm->abort_flag.store(1);
        for (void *p : c->opened) (void)hipIpcCloseMemHandle(p);
        if (c->staging) (void)hipFree(c->staging);
        if (c->ctl) (void)hipFree(c->ctl);
        if (c->status) (void)hipHostFree(c->status);
        munmap(c->shm, sizeof(Shm));
        delete c;
        return r;
    };
    // two areas of staging_bytes each: [0, staging_bytes) for the collectives, [staging_bytes, 2 x) for point-to-point messages (a
    // slot per destination).  They must not share bytes: a rank may already stage its next all-reduce while a slower peer still
    // copies the neighbour range addressed to it out of the same buffer (seen as a wrong BiCGStab iterate once in four runs).
    if (hipMalloc(&c->staging, 2 * c->staging_bytes) != hipSuccess) return bail(ncclUnhandledCudaError, "staging buffer allocation");
    if (hipMemset(c->staging, 0, 2 * c->staging_bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return bail(ncclUnhandledCudaError, "staging buffer clear");
    if (hipIpcGetMemHandle(&c->shm->staging[rank], c->staging) != hipSuccess) return bail(ncclUnhandledCudaError, "hipIpcGetMemHandle");
    {   // the control block: written by peers while a local kernel polls it -- it must not live in this GPU's L2 as ordinary memory does
        void *p = nullptr;
        hipError_t e = hipExtMallocWithFlags(&p, sizeof(Ctl), hipDeviceMallocUncached);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, sizeof(Ctl), hipDeviceMallocFinegrained); }
        if (e != hipSuccess) return bail(ncclUnhandledCudaError, "control block allocation (uncached / fine-grained device memory)");
        c->ctl = static_cast<Ctl *>(p);
        if (hipMemset(c->ctl, 0, sizeof(Ctl)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return bail(ncclUnhandledCudaError, "control block clear");
        if (hipIpcGetMemHandle(&c->shm->ctl[rank], c->ctl) != hipSuccess) return bail(ncclUnhandledCudaError, "hipIpcGetMemHandle (control block)");
        if (hipHostMalloc((void **)&c->status, 4 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return bail(ncclUnhandledCudaError, "status word");
        for (int i = 0; i < 4; i++) c->status[i] = 0;
        if (hipHostGetDevicePointer((void **)&c->status_dev, c->status, 0) != hipSuccess) return bail(ncclUnhandledCudaError, "status word (device view)");
    }
    c->shm->staging_bytes[rank] = c->staging_bytes;
    c->shm->magic.store(MAGIC);
    c->shm->handle_ready[rank].store(1, std::memory_order_release);
    c->shm->attached.fetch_add(1);
    for (int q = 0; q < nranks; q++) {
        if (!wait_until(c, [&] { return c->shm->handle_ready[q].load(std::memory_order_acquire) == 1; }, "ncclCommInitRank (waiting for the peers)"))
            return bail(ncclRemoteError, "a peer never arrived");
        if (c->shm->staging_bytes[q] != c->staging_bytes) return bail(ncclInvalidArgument, "the ranks disagree on FAKE_RCCL_STAGING_MB");
        if (q == rank) { c->peer[q] = c->staging; c->peer_ctl[q] = c->ctl; continue; }
        void *p = nullptr;
        if (hipIpcOpenMemHandle(&p, c->shm->staging[q], hipIpcMemLazyEnablePeerAccess) != hipSuccess) return bail(ncclUnhandledCudaError, "hipIpcOpenMemHandle");
        c->opened.push_back(p);
        c->peer[q] = static_cast<char *>(p);
        if (hipIpcOpenMemHandle(&p, c->shm->ctl[q], hipIpcMemLazyEnablePeerAccess) != hipSuccess) return bail(ncclUnhandledCudaError, "hipIpcOpenMemHandle (control block)");
        c->opened.push_back(p);
        c->peer_ctl[q] = static_cast<Ctl *>(p);
    }
    // everybody has mapped the block: its name can go (the mappings keep it alive; nothing is left behind if the ranks are killed)
    if (!wait_until(c, [&] { return c->shm->attached.load() >= nranks; }, "ncclCommInitRank (attach)")) return bail(ncclRemoteError, "a peer never attached");
    if (rank == 0) shm_unlink(id.internal);
    if (c->verbose) std::fprintf(stderr, "[fake_rccl rank %d/%d] communicator up, staging %zu MiB\n", rank, nranks, mb);
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    (void)hipDeviceSynchronize();
    if (c->verbose || std::getenv("FAKE_RCCL_STATS"))
        std::fprintf(stderr, "[fake_rccl rank %d/%d] calls: all-gather %llu, all-reduce %llu, reduce-scatter %llu, send %llu, recv %llu, groups %llu\n",
                     c->me, c->P, (unsigned long long)c->calls[0], (unsigned long long)c->calls[1], (unsigned long long)c->calls[2],
                     (unsigned long long)c->calls[3], (unsigned long long)c->calls[4], (unsigned long long)c->calls[5]);
    // nobody unmaps a buffer a peer may still read: leave together (or after the time-out, when a peer is gone)
    (void)failed(c);        // (reports what a kernel left behind, if nobody asked since)
    c->shm->detached.fetch_add(1);
    if (!c->async_err.load()) (void)wait_until(c, [&] { return c->shm->detached.load() >= c->P; }, "ncclCommDestroy");
    for (void *p : c->opened) (void)hipIpcCloseMemHandle(p);
    if (c->staging) (void)hipFree(c->staging);
    if (c->ctl) (void)hipFree(c->ctl);
    if (c->status) (void)hipHostFree(c->status);
    munmap(c->shm, sizeof(Shm));
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(ncclComm_t c)
{
    if (!c) return ncclInvalidArgument;
    comm_abort(c, "ncclCommAbort");
    return ncclCommDestroy(c);
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t c, ncclResult_t *err)
{
    if (!c || !err) return ncclInvalidArgument;
    (void)failed(c);
    *err = (ncclResult_t)c->async_err.load();
    return ncclSuccess;
}

// test hook: per-kind call counts of this rank (all-gather, all-reduce, reduce-scatter, send, recv, groups)
void fakeRcclCounts(ncclComm_t c, unsigned long long *six) { for (int i = 0; i < 6; i++) six[i] = c ? c->calls[i] : 0; }

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
    case ncclSystemError: return "unhandled system error (fake_rccl)";
    case ncclInternalError: return "internal error (fake_rccl)";
    case ncclInvalidArgument: return "invalid argument (fake_rccl)";
    case ncclInvalidUsage: return "invalid usage (fake_rccl)";
    case ncclRemoteError: return "remote process exited or fell out of step (fake_rccl)";
    default: return "unknown result code (fake_rccl)";
    }
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    NCCLOK(check(c));
    const size_t ts = type_size(t);
    if (!ts || (!send && count) || (!recv && count)) return ncclInvalidArgument;
    c->calls[0]++;
    const size_t bytes = count * ts;
    const size_t chunk = c->staging_bytes;
    const char *src = static_cast<const char *>(send);
    char *dst = static_cast<char *>(recv);
    for (size_t off = 0; off < bytes || off == 0; off += chunk) {
        const size_t n = std::min(chunk, bytes - off);
        const uint64_t seq = c->seq++;
        const uint64_t sig = signature(1, count, t, seq);
        if (n) HIPOK(hipMemcpyAsync(c->staging, src + off, n, hipMemcpyDeviceToDevice, s));
        NCCLOK(enqueue_meet(c, s, seq, 0, sig, "ncclAllGather"));
        for (int q = 0; q < c->P && n; q++) HIPOK(hipMemcpyAsync(dst + (size_t)q * bytes + off, c->peer[q], n, hipMemcpyDeviceToDevice, s));
        NCCLOK(enqueue_meet(c, s, seq, 1, sig, "ncclAllGather (release)"));
        if (bytes == 0) break;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    NCCLOK(check(c));
    const size_t ts = type_size(t);
    if (!ts || (!send && count) || (!recv && count)) return ncclInvalidArgument;
    if (op != ncclSum) { std::fprintf(stderr, "[fake_rccl] only ncclSum is modelled\n"); return ncclInvalidArgument; }
    c->calls[1]++;
    const size_t per = c->staging_bytes / ts;
    const char *src = static_cast<const char *>(send);
    char *dst = static_cast<char *>(recv);
    for (size_t off = 0; off < count || off == 0; off += per) {
        const size_t n = std::min(per, count - off);
        const uint64_t seq = c->seq++;
        const uint64_t sig = signature(2, count, t, seq);
        if (n) HIPOK(hipMemcpyAsync(c->staging, src + off * ts, n * ts, hipMemcpyDeviceToDevice, s));
        NCCLOK(enqueue_meet(c, s, seq, 0, sig, "ncclAllReduce"));
        NCCLOK(launch_sum(c, t, 0, dst + off * ts, n, s));
        NCCLOK(enqueue_meet(c, s, seq, 1, sig, "ncclAllReduce (release)"));
        if (count == 0) break;
    }
    return ncclSuccess;
}

ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    NCCLOK(check(c));
    const size_t ts = type_size(t);
    if (!ts || (!send && recvcount) || (!recv && recvcount)) return ncclInvalidArgument;
    if (op != ncclSum) { std::fprintf(stderr, "[fake_rccl] only ncclSum is modelled\n"); return ncclInvalidArgument; }
    c->calls[2]++;
    const size_t per = c->staging_bytes / ts / (size_t)c->P;       // elements of every destination's block per round
    if (per == 0) return ncclInvalidArgument;
    const char *src = static_cast<const char *>(send);
    char *dst = static_cast<char *>(recv);
    for (size_t off = 0; off < recvcount || off == 0; off += per) {
        const size_t n = std::min(per, recvcount - off);
        const uint64_t seq = c->seq++;
        const uint64_t sig = signature(3, recvcount, t, seq);
        for (int d = 0; d < c->P && n; d++)     // staging layout of a round: [destination][per]
            HIPOK(hipMemcpyAsync(c->staging + (size_t)d * per * ts, src + ((size_t)d * recvcount + off) * ts, n * ts, hipMemcpyDeviceToDevice, s));
        NCCLOK(enqueue_meet(c, s, seq, 0, sig, "ncclReduceScatter"));
        NCCLOK(launch_sum(c, t, (size_t)c->me * per, dst + off * ts, n, s));
        NCCLOK(enqueue_meet(c, s, seq, 1, sig, "ncclReduceScatter (release)"));
        if (recvcount == 0) break;
    }
    return ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    return p2p_call(true, buf, nullptr, count, t, peer, c, s);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s)
{
    return p2p_call(false, nullptr, buf, count, t, peer, c, s);
}

ncclResult_t ncclGroupStart(void)
{
    g_group_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void)
{
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    ncclResult_t r = ncclSuccess;
    if (g_group_comm) {
        g_group_comm->calls[5]++;
        r = run_group(g_group_comm, g_group_stream, g_pending);
    }
    g_pending.clear();
    g_group_comm = nullptr; g_group_stream = nullptr;
    return r;
}

}  // extern "C"
