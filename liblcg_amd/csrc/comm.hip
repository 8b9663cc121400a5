// comm.hip -- row-sharded operation over RCCL (one process per GPU, xGMI underneath).
//
// Nothing like this exists in the reference (SURVEY.md sections 2 #22-23, 8e).  Layout:
// rank r owns rows [r*rpr, min(N,(r+1)*rpr)), rpr = ceil(N/P), of A and of every vector.
//   A.x   : the local x slices are all-gathered into xfull (P*rpr doubles; only the tail of
//           the last slice is padding, so GLOBAL column indices address xfull directly).
//           The shard is split once into entries with locally-owned columns and the rest:
//           y = A_loc.x_loc runs on the compute stream WHILE the second stream gathers and then
//           multiplies the (few) rows that hold remote columns; their sums are added to y when
//           that stream's event fires.
//   dots  : k_scal reduces the local partials into DevState::red, one ncclAllReduce (sum,
//           <= MAXR doubles) makes them global, the scalar recurrence continues on device.
// RCCL is bound at run time (dlopen) so that the library loads on machines without it and
// shares the copy already mapped by the host program (e.g. PyTorch's).
#include <dlfcn.h>

#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "devcommon.hpp"

namespace lcgh {


struct Comm {
    void *lib = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool force = false;     // LCG_HIP_FORCE_COMM: run the collectives even with one rank (tests)
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string path;       // the shared object ncclAllGather was bound from
};

static Comm g_comm;     // function table (valid once lib != nullptr)

static int comm_fail(const char *what, ncclResult_t r)
{
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s: %s", what, g_comm.GetErrorString ? g_comm.GetErrorString(r) : "rccl error");
    ctx().err = buf;
    return LCG_HIP_E_COMM;
}

static int load_rccl()
{
    if (g_comm.lib) return 0;
    const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", nullptr};
    void *h = nullptr;
    // LCG_HIP_RCCL_LIB = path: exactly that build of the collectives library, privately (RTLD_LOCAL: the host program's own
    // copy, e.g. PyTorch's, keeps its symbols) -- a site's tuned RCCL, or tests/fake_rccl (several ranks on one GPU)
    if (const char *path = std::getenv("LCG_HIP_RCCL_LIB")) {
        if (*path) {
            h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
            if (!h) { ctx().err = std::string("LCG_HIP_RCCL_LIB: cannot load ") + path + ": " + dlerror(); return LCG_HIP_E_COMM; }
        }
    }
    // otherwise prefer a copy that is already mapped (RTLD_NOLOAD), then a fresh load
    for (int pass = 0; pass < 2 && !h; pass++)
        for (int i = 0; names[i] && !h; i++)
            h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
    if (!h) { ctx().err = std::string("cannot load librccl: ") + dlerror(); return LCG_HIP_E_COMM; }
    g_comm.lib = h;
#define SYM(field, name)                                                         \
    g_comm.field = reinterpret_cast<decltype(g_comm.field)>(dlsym(h, name));    \
    if (!g_comm.field) { ctx().err = "librccl lacks " name; g_comm.lib = nullptr; return LCG_HIP_E_COMM; }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(AllGather, "ncclAllGather")
    SYM(ReduceScatter, "ncclReduceScatter")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    Dl_info info;
    g_comm.path = (dladdr(reinterpret_cast<void *>(g_comm.AllGather), &info) && info.dli_fname) ? info.dli_fname : "?";
    return 0;
}

struct Xg;
static bool xg_world(int *P, int *me);
bool comm_active()
{
    if (g_comm.comm != nullptr && (g_comm.nranks > 1 || g_comm.force)) return true;
    int P = 1, me = 0;
    return xg_world(&P, &me) && P > 1;      // mailboxes only (no RCCL): the one-GPU rehearsal of several ranks
}

// ---- direct all-reduce over peer-mapped mailboxes (lcg_hip.h: lcg_hip_p2p_*) -------------------
struct Xg {
    bool connected = false, enabled = false;
    int P = 0, me = 0;
    double *mine = nullptr;             // [2][XG_MAXP][XG_SLOT], uncached (or fine-grained) device memory
    const char *mem_kind = "";
    double **peers_dev = nullptr;       // device array [P]
    unsigned long long *seq = nullptr;
    int *fail = nullptr;
    std::vector<void *> opened;         // peer mappings to close
    long long timeout_ticks = 2000000000LL;     // 20 s of the 100 MHz wall clock
};
static Xg g_xg;
static unsigned long long g_xg_generation = 0;  // connections made so far: a Direct plan belongs to exactly one
constexpr size_t XG_BYTES = sizeof(double) * 2 * XG_MAXP * XG_SLOT;

static bool xg_world(int *P, int *me)
{
    if (!g_xg.enabled) return false;
    *P = g_xg.P; *me = g_xg.me;
    return true;
}
static int world_size() { return g_comm.comm ? g_comm.nranks : (g_xg.enabled ? g_xg.P : 1); }
static int world_rank() { return g_comm.comm ? g_comm.rank : (g_xg.enabled ? g_xg.me : 0); }

bool xg_box(XgBox *out)
{
    if (!g_xg.enabled) return false;
    out->mine = g_xg.mine; out->peers = g_xg.peers_dev; out->seq = g_xg.seq; out->fail = g_xg.fail;
    out->timeout_ticks = g_xg.timeout_ticks; out->P = g_xg.P; out->me = g_xg.me;
    return true;
}

// stand-alone form: `count` doubles in device memory, in place
__global__ __launch_bounds__(VB) void k_xg_allreduce(double *v, int count, XgBox xb, int *ok_out)
{
    __shared__ double sums[MAXR];
    if ((int)threadIdx.x < MAXR) sums[threadIdx.x] = (int)threadIdx.x < count ? v[threadIdx.x] : 0.0;
    __syncthreads();
    const bool ok = xg_allreduce<MAXR>(xb, sums, xg_begin(xb));
    if ((int)threadIdx.x < count && ok) v[threadIdx.x] = sums[threadIdx.x];
    if (threadIdx.x == 0 && ok_out) *ok_out = ok ? 1 : 0;
}

// all-gather of MAXR 64-bit words per rank through the same mailboxes: out[q][r] = word r of rank q
__global__ __launch_bounds__(VB) void k_xg_allgather(const double *in, double *out, XgBox xb, int *ok_out)
{
    __shared__ double mine[MAXR];
    __shared__ double got[MAXR][XG_MAXP];
    if ((int)threadIdx.x < MAXR) mine[threadIdx.x] = in[threadIdx.x];
    __syncthreads();
    const bool ok = xg_exchange<MAXR>(xb, mine, got, xg_begin(xb));
    if (ok)
        for (int i = threadIdx.x; i < MAXR * xb.P; i += VB) out[i] = got[i % MAXR][i / MAXR];
    if (threadIdx.x == 0) *ok_out = ok ? 1 : 0;
}

// host form (collective, synchronises): words = 8 x 64 bit in, P x 8 out.  Doubles only carry the
// bits (plain 64-bit loads and stores end to end).
static int xg_allgather_host(const unsigned long long *in8, unsigned long long *outP8)
{
    Ctx &c = ctx();
    XgBox xb;
    if (!xg_box(&xb)) { c.err = "direct exchange needs the mailboxes (lcg_hip_p2p_enable)"; return LCG_HIP_E_COMM; }
    double *d = nullptr; int *dok = nullptr;
    HIPCHK(hipMalloc(&d, sizeof(double) * MAXR * (size_t)(xb.P + 1)));
    HIPCHK(hipMalloc(&dok, sizeof(int)));
    int hok = 0;
    hipError_t e = hipMemcpyAsync(d, in8, sizeof(double) * MAXR, hipMemcpyHostToDevice, c.stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_xg_allgather, dim3(1), dim3(VB), 0, c.stream, d, d + MAXR, xb, dok);
        e = hipMemcpyAsync(outP8, d + MAXR, sizeof(double) * MAXR * (size_t)xb.P, hipMemcpyDeviceToHost, c.stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&hok, dok, sizeof(int), hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    hipFree(d); hipFree(dok);
    if (e != hipSuccess) return fail(e, "mailbox all-gather", __FILE__, __LINE__);
    if (!hok) { c.err = "mailbox all-gather: a peer did not answer in time"; return LCG_HIP_E_COMM; }
    return 0;
}

// any number of words per rank (rounds of 8): out[q*nwords + i]
static int xg_allgather_words(const unsigned long long *in, int nwords, std::vector<unsigned long long> &out)
{
    const int P = g_xg.P;
    out.assign((size_t)P * nwords, 0);
    std::vector<unsigned long long> rnd((size_t)P * MAXR);
    for (int w0 = 0; w0 < nwords; w0 += MAXR) {
        unsigned long long buf[MAXR] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < MAXR && w0 + i < nwords; i++) buf[i] = in[w0 + i];
        int rc = xg_allgather_host(buf, rnd.data());
        if (rc) return rc;
        for (int q = 0; q < P; q++)
            for (int i = 0; i < MAXR && w0 + i < nwords; i++) out[(size_t)q * nwords + w0 + i] = rnd[(size_t)q * MAXR + i];
    }
    return 0;
}

// true only if every rank passed true (collective)
static int xg_agree(bool mine, bool *all)
{
    unsigned long long w = mine ? 1 : 0;
    std::vector<unsigned long long> out;
    int rc = xg_allgather_words(&w, 1, out);
    if (rc) return rc;
    *all = true;
    for (unsigned long long v : out) *all = *all && v == 1;
    return 0;
}

int comm_allreduce(double *dev, int count, hipStream_t s)
{
    XgBox xb;
    if (xg_box(&xb)) {
        if (count > MAXR) { ctx().err = "direct all-reduce: more than 8 values"; return LCG_HIP_E_ARG; }
        hipLaunchKernelGGL(k_xg_allreduce, dim3(1), dim3(VB), 0, s, dev, count, xb, (int *)nullptr);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (!g_comm.comm) return 0;
    ncclResult_t r = g_comm.AllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, g_comm.comm, s);
    if (r != ncclSuccess) return comm_fail("ncclAllReduce", r);
    return 0;
}

// local row count summed over ranks (the N of the stop rule, lcg.cpp:208)
double global_rows(Ctx &c, int n)
{
    if (!comm_active()) return (double)n;
    double v = (double)n;
    if (hipMemcpyAsync(c.state->red, &v, sizeof v, hipMemcpyHostToDevice, c.stream) != hipSuccess) return (double)n;
    if (comm_allreduce(c.state->red, 1, c.stream)) return (double)n;
    if (hipMemcpyAsync(&v, c.state->red, sizeof v, hipMemcpyDeviceToHost, c.stream) != hipSuccess) return (double)n;
    hipStreamSynchronize(c.stream);
    return v;
}

// The same number without a collective when the callback is the built-in A.x of a distributed matrix: the handle knows
// the global size (the all-reduce and its two copies drain the stream: ~60 us per solve, a tenth of a 20-iteration sharded
// solve at the 8-way shard size).  A pure function of the handle, so every rank takes the same branch.
double global_rows_of(Ctx &c, int n, const void *afp, const void *inst)
{
    if (comm_active() && inst && (afp == (const void *)lcg_hip_csr_ax || afp == (const void *)clcg_hip_csr_ax)) {
        const lcg_hip_csr *A = static_cast<const lcg_hip_csr *>(inst);
        if (A->distributed && A->n_rows == n && A->n_global > 0) return (double)A->n_global;
    }
    return global_rows(c, n);
}

// ---- shard split ------------------------------------------------------------------------------
__global__ void k_split_count(int n, long lo, long hi, const int *rowptr, const int *col, int *cl, int *cr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int a = 0;
    for (int k = rowptr[i]; k < rowptr[i + 1]; k++) a += (col[k] >= lo && col[k] < hi);
    cl[i] = a; cr[i] = rowptr[i + 1] - rowptr[i] - a;
}
template <class V>
__global__ void k_split_fill(int n, long lo, long hi, const int *rowptr, const int *col, const V *val,
                             const int *rpl, int *coll, V *vall, const int *rpr, int *colr, V *valr)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int a = rpl[i], b = rpr[i];
    for (int k = rowptr[i]; k < rowptr[i + 1]; k++) {
        const int c = col[k];
        if (c >= lo && c < hi) { coll[a] = (int)(c - lo); vall[a++] = val[k]; }
        else { colr[b] = c; valr[b++] = val[k]; }
    }
}

// Rows of the remote part that hold anything (for a banded matrix: the two ends of the shard).
__global__ void k_row_flags(int n, const int *rowptr, int *flag)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = rowptr[i + 1] > rowptr[i];
}
__global__ void k_row_compact(int n, const int *rowptr, const int *pos, int *rows, int *crowptr, int n_kept)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && rowptr[i + 1] > rowptr[i]) { rows[pos[i]] = i; crowptr[pos[i]] = rowptr[i]; }
    if (i == 0) crowptr[n_kept] = rowptr[n];
}
template <class V>
__global__ __launch_bounds__(VB) void k_scatter_add(int nr, const int *__restrict__ rows, const V *__restrict__ part,
                                                    V *__restrict__ y, const int *done)
{
    if (done && *done) return;
    const int j = blockIdx.x * VB + threadIdx.x;
    if (j < nr) { const int i = rows[j]; y[i] = vadd(y[i], part[j]); }
}
// a block's share of sum_j u[row_j] * (what the block added to y[row_j]): the remote-column part of the y.u a sharded product
// carries (the local product leaves the rest: csr.hip, csr_part_ax_dot) -- one partial per block, fixed order
// `big` (may be null): the local product's per-block sums; this block also adds its slice [blockIdx * per, + per) of them -- the
// second stage of the local sums costs no launch of its own
__device__ __forceinline__ void block_dot_store(double c, double *out, const double *__restrict__ big = nullptr, int nbig = 0, int per = 0)
{
    __shared__ double sh[VB / 64];
    if (big) {
        const int lo = blockIdx.x * per, hi = min(nbig, lo + per);
        for (int j = lo + (int)threadIdx.x; j < hi; j += VB) c += big[j];
    }
    const double t = wave_sum(c);
    if ((threadIdx.x & 63) == WSUM_LANE) sh[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < VB / 64; k++) v += sh[k];
        out[blockIdx.x] = v;
    }
}
__global__ __launch_bounds__(VB) void k_scatter_add_dot(int nr, const int *__restrict__ rows, const double *__restrict__ part,
                                                        double *__restrict__ y, const double *__restrict__ u, double *__restrict__ dot_out,
                                                        const double *__restrict__ big, int nbig, int per, const int *done)
{
    if (done && *done) return;
    const int j = blockIdx.x * VB + threadIdx.x;
    double c = 0.0;
    if (j < nr) { const int i = rows[j]; const double p = part[j]; y[i] += p; c = u[i] * p; }
    block_dot_store(c, dot_out, big, nbig, per);
}

static int alloc_cols(CsrPart &P, int n, long nnz, bool cplx)
{
    P.n_rows = n; P.nnz = nnz; P.owned = true;
    HIPCHK(hipMalloc(&P.col, sizeof(int) * (size_t)(nnz > 0 ? nnz : 1) + 64));
    HIPCHK(hipMalloc(&P.val, sizeof(double) * (cplx ? 2 : 1) * (size_t)(nnz > 0 ? nnz : 1) + 64));
    P.padded = true;
    return 0;
}

// ---- neighbour (range) exchange plan ---------------------------------------------------------------
// Instead of gathering all of x, rank r receives from each peer q only the CONTIGUOUS range of q's
// slice that r's remote columns touch, straight into xfull at its global offset (so the remote
// part of the shard needs no column remapping).  For a banded matrix that is 2*W entries per rank
// instead of N*(P-1)/P; for scattered columns it degenerates to the all-gather volume.
struct HaloPlan {
    std::vector<long long> need;    // [2*P]: lo, hi (global columns) I need from peer q; lo >= hi = nothing
    std::vector<long long> give;    // [2*P]: lo, hi (global rows of MINE) peer q needs
    long long recv_total = 0, send_total = 0;
};

__global__ void k_need_ranges(long nnz, const int *col, long rpr, int nranks, long long *lohi)
{
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (long)gridDim.x * blockDim.x) {
        const long c = col[k];
        const int q = (int)(c / rpr);
        if (q < nranks) { atomicMin(&lohi[2 * q], (long long)c); atomicMax(&lohi[2 * q + 1], (long long)c + 1); }
    }
}

// lo/hi per owner rank of the remote part's columns (device scan, host result)
static int need_ranges(lcg_hip_csr *A, int nranks, std::vector<long long> &out)
{
    Ctx &c = ctx();
    out.assign(2 * (size_t)nranks, 0);
    for (int q = 0; q < nranks; q++) { out[2 * q] = (1LL << 62); out[2 * q + 1] = 0; }
    if (A->rem.nnz == 0) return 0;
    long long *d = nullptr;
    HIPCHK(hipMalloc(&d, sizeof(long long) * 2 * nranks));
    hipError_t e = hipMemcpyAsync(d, out.data(), sizeof(long long) * 2 * nranks, hipMemcpyHostToDevice, c.stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_need_ranges, dim3(1024), dim3(VB), 0, c.stream, (long)A->rem.nnz, A->rem.col, (long)A->rows_per_rank, nranks, d);
        e = hipMemcpyAsync(out.data(), d, sizeof(long long) * 2 * nranks, hipMemcpyDeviceToHost, c.stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    hipFree(d);
    if (e != hipSuccess) return fail(e, "halo ranges", __FILE__, __LINE__);
    return 0;
}

static int halo_setup(lcg_hip_csr *A)
{
    Ctx &c = ctx();
    const int P = g_comm.nranks, me = g_comm.rank;
    HaloPlan *h = new HaloPlan();
    int rc = need_ranges(A, P, h->need);
    if (rc) { delete h; return rc; }
    // everybody learns everybody's needs: table[q][p] = what q needs from p
    long long *dmine = nullptr, *dall = nullptr;
    std::vector<long long> table(2 * (size_t)P * P);
    hipError_t e = hipMalloc(&dall, sizeof(long long) * 2 * P * P);
    if (e != hipSuccess) { delete h; return fail(e, "halo table", __FILE__, __LINE__); }
    dmine = dall + 2 * (size_t)P * me;
    e = hipMemcpyAsync(dmine, h->need.data(), sizeof(long long) * 2 * P, hipMemcpyHostToDevice, c.stream);
    if (e == hipSuccess) {
        ncclResult_t r = g_comm.AllGather(dmine, dall, 2 * (size_t)P, ncclInt64, g_comm.comm, c.stream);
        if (r != ncclSuccess) { hipFree(dall); delete h; return comm_fail("halo ncclAllGather", r); }
        e = hipMemcpyAsync(table.data(), dall, sizeof(long long) * 2 * P * P, hipMemcpyDeviceToHost, c.stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    hipFree(dall);
    if (e != hipSuccess) { delete h; return fail(e, "halo table exchange", __FILE__, __LINE__); }
    h->give.assign(2 * (size_t)P, 0);
    for (int q = 0; q < P; q++) {
        h->give[2 * q] = table[2 * ((size_t)q * P + me)];
        h->give[2 * q + 1] = table[2 * ((size_t)q * P + me) + 1];
        if (q == me) { h->need[2 * q] = h->need[2 * q + 1] = 0; h->give[2 * q] = h->give[2 * q + 1] = 0; }
        if (h->need[2 * q + 1] > h->need[2 * q]) h->recv_total += h->need[2 * q + 1] - h->need[2 * q];
        if (h->give[2 * q + 1] > h->give[2 * q]) h->send_total += h->give[2 * q + 1] - h->give[2 * q];
    }
    A->halo = h;
    return 0;
}

static int halo_exchange(lcg_hip_csr *A, const double *x, hipStream_t s)
{
    const HaloPlan *h = static_cast<const HaloPlan *>(A->halo);
    const int P = g_comm.nranks;
    const size_t w = A->is_complex ? 2 : 1;
    ncclResult_t r = g_comm.GroupStart();
    if (r != ncclSuccess) return comm_fail("ncclGroupStart", r);
    for (int q = 0; q < P && r == ncclSuccess; q++) {
        const long long lo = h->give[2 * q], hi = h->give[2 * q + 1];
        if (hi > lo) r = g_comm.Send(x + w * (size_t)(lo - A->row0), w * (size_t)(hi - lo), ncclDouble, q, g_comm.comm, s);
    }
    for (int q = 0; q < P && r == ncclSuccess; q++) {
        const long long lo = h->need[2 * q], hi = h->need[2 * q + 1];
        if (hi > lo) r = g_comm.Recv(A->xfull + w * (size_t)lo, w * (size_t)(hi - lo), ncclDouble, q, g_comm.comm, s);
    }
    ncclResult_t r2 = g_comm.GroupEnd();
    if (r != ncclSuccess) return comm_fail("ncclSend/Recv", r);
    if (r2 != ncclSuccess) return comm_fail("ncclGroupEnd", r2);
    return 0;
}


// ---- direct neighbour exchange over peer mappings (dist mode 2) ------------------------------------
// Same plan as the neighbour exchange above (contiguous ranges per owner), but no collective call
// and no second stream: the OWNER writes its range of x into the neighbours' receive buffers
// (pushing blocks ride in front of the local product's grid: csr.hip) and raises its flag word
// there; the remote-column product of a neighbour waits for the flags of the same call.
//   recv   : landing zone, [2][P*rpr] doubles (x w) of UNCACHED device memory addressed by GLOBAL
//            column: it is written by peers over the fabric and read once, by this GPU, with coalesced
//            loads (k_recv) into the ordinary gather buffer xfull that the remote-column product reads
//            -- the same split RCCL makes (uncached transport buffer, local copy out).  The two halves
//            alternate per call: a neighbour can be one call ahead, never two, because its next push
//            needs my flag of this call first (the neighbourhood is made symmetric for that).
//   flags  : [XG_MAXP] 64-bit words by source rank, uncached memory (polled while peers write)
struct Direct {
    double *recv = nullptr;
    unsigned long long *flags = nullptr;
    unsigned int *ticket = nullptr;
    size_t half = 0;                    // doubles per half of recv
    unsigned long long calls = 0;       // A.x calls made with this matrix (same on every rank)
    bool uses_mailbox = true;           // plans point into the mailbox state (false for the self-loop rehearsal)
    unsigned long long generation = 0;  // mailbox connection the plans were made under (their fail word lives there)
    int nnb = 0;
    int nb_rank[XG_MAXSEG];
    double *nb_recv[XG_MAXSEG];         // neighbour's recv as mapped here
    unsigned long long *nb_flags[XG_MAXSEG];
    long long give_lo[XG_MAXSEG], give_hi[XG_MAXSEG];   // global rows of mine the neighbour reads (may be empty)
    std::vector<void *> opened;
    long long recv_total = 0;
    long long need_lo[XG_MAXSEG], need_hi[XG_MAXSEG];   // global columns I read from the neighbour (may be empty)
    PushPlan push;      // src/dst/seq filled per call
    PushPlan copy;      // landing zone -> xfull (same shape: segments in chunks of PUSH_CHUNK)
    WaitPlan wait;
};

__global__ __launch_bounds__(VB) void k_push(PushPlan pp) { push_block(pp, blockIdx.x); }

// Wait for the neighbours' flags of this call, then move what they wrote from the landing zone into
// the gather buffer (block b moves chunk b; 8-byte system-scope loads, coalesced).
__global__ __launch_bounds__(VB) void k_recv(WaitPlan wp, PushPlan cp, DevState *st)
{
    if (!wait_flags<true>(wp)) {
        if (st && blockIdx.x == 0 && threadIdx.x == 0) { st->done = 1; st->status = ST_COMM; }
        return;
    }
    if (cp.nseg == 0) return;
    const int b = blockIdx.x;
    int s = 0;
    while (s + 1 < cp.nseg && b >= cp.first_block[s + 1]) s++;
    const long off = (long)(b - cp.first_block[s]) * PUSH_CHUNK;
    const long cnt = min((long)PUSH_CHUNK, cp.count[s] - off);
    double *src = const_cast<double *>(cp.src[s]) + off;
    double *dst = cp.dst[s] + off;
    constexpr int PER = PUSH_CHUNK / VB;
    double v[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const long i = threadIdx.x + (long)q * VB;
        v[q] = __hip_atomic_load(src + (i < cnt ? i : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const long i = threadIdx.x + (long)q * VB;
        if (i < cnt) dst[i] = v[q];
    }
}

// Remote-column product of the rows that have remote columns (a few % of the shard):
// y[rows[j]] += sum_k val[k] * xfull[col[k]], T lanes per row.  The kernel is a chain of dependent
// loads over many short rows, not bandwidth: the target row and its current y are fetched beside the
// row bounds, and a lane keeps four entries in flight.
// TOY: add into y (single-stream form) or store the compact sums out[j] (two-stream form: the sums are
// computed beside the local product and added by k_scatter_add once both are done).
template <class V, int T, bool TOY, bool DOT = false>
__global__ __launch_bounds__(VB) void k_remote(int nr, const int *__restrict__ rowptr, const int *__restrict__ col,
                                               const V *__restrict__ val, const int *__restrict__ rows,
                                               const V *__restrict__ xfull, V *__restrict__ y, const int *done,
                                               const double *__restrict__ u = nullptr, double *__restrict__ dot_out = nullptr,
                                               const double *__restrict__ big = nullptr, int nbig = 0, int per = 0)
{
    static_assert(!DOT || (TOY && sizeof(V) == 8), "the remote part carries its share of y.u in the real form that adds into y");
    if (done && *done) return;
    const long gt = (long)blockIdx.x * VB + threadIdx.x;
    const long j = gt / T;
    const int lane = (int)(gt % T);
    V acc = vzero(V()), yold = vzero(V());
    int i = 0;
    if (j < nr) {
        const int b = rowptr[j], e = rowptr[j + 1];
        int k = b + lane;
        if (TOY) { i = rows[j]; if (lane == 0) yold = y[i]; }
        for (; k + 3 * T < e; k += 4 * T) {
            int c[4]; V a[4], xv[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { c[q] = col[k + q * T]; a[q] = val[k + q * T]; }
#pragma unroll
            for (int q = 0; q < 4; q++) xv[q] = xfull[c[q]];
#pragma unroll
            for (int q = 0; q < 4; q++) acc = mac(a[q], xv[q], acc);
        }
        if (k + T < e) {
            const int c0 = col[k], c1 = col[k + T];
            const V a0 = val[k], a1 = val[k + T];
            const V x0 = xfull[c0], x1 = xfull[c1];
            acc = mac(a0, x0, acc); acc = mac(a1, x1, acc);
            k += 2 * T;
        }
        if (k < e) { const int c0 = col[k]; acc = mac(val[k], xfull[c0], acc); }
    }
#pragma unroll
    for (int off = T / 2; off > 0; off >>= 1) acc = vadd(acc, shfl_down_v(acc, off, T));
    if (j < nr && lane == 0) { if (TOY) y[i] = vadd(yold, acc); else y[j] = acc; }
    if constexpr (DOT) block_dot_store((j < nr && lane == 0) ? u[i] * acc : 0.0, dot_out, big, nbig, per);
}

static void direct_free(lcg_hip_csr *A)
{
    Direct *D = static_cast<Direct *>(A->direct);
    if (!D) return;
    if (ctx().inited) (void)hipDeviceSynchronize();
    for (void *p : D->opened) (void)hipIpcCloseMemHandle(p);
    if (D->recv) (void)hipFree(D->recv);
    if (D->flags) (void)hipFree(D->flags);
    if (D->ticket) (void)hipFree(D->ticket);
    delete D;
    A->direct = nullptr;
}

// landing zone -> xfull: one segment per neighbour I take from (dst fixed, src depends on the parity)
static void direct_copy_plan(lcg_hip_csr *A, Direct *D)
{
    const size_t w = A->is_complex ? 2 : 1;
    PushPlan &cp = D->copy;
    cp = PushPlan();
    int nb = 0;
    for (int s = 0; s < D->nnb; s++) {
        const long long cnt = (D->need_hi[s] - D->need_lo[s]) * (long long)w;
        if (cnt <= 0) continue;
        cp.count[cp.nseg] = (long)cnt;
        cp.dst[cp.nseg] = A->xfull + w * (size_t)D->need_lo[s];
        cp.first_block[cp.nseg] = nb;
        nb += (int)((cnt + PUSH_CHUNK - 1) / PUSH_CHUNK);
        cp.nseg++;
    }
    cp.first_block[cp.nseg] = nb;
    cp.nblocks = nb > 0 ? nb : 1;
}

// Collective over the mailboxes.  Every rank takes the same decision at every step (xg_agree), so
// a rank that cannot go on makes all of them return LCG_HIP_E_COMM and the caller picks another mode.
static int direct_setup(lcg_hip_csr *A)
{
    Ctx &c = ctx();
    XgBox xb;
    if (!xg_box(&xb)) { c.err = "direct exchange needs the mailboxes (lcg_hip_p2p_connect + enable)"; return LCG_HIP_E_COMM; }
    const int P = xb.P, me = xb.me;
    const size_t w = A->is_complex ? 2 : 1;
    Direct *D = new Direct();
    A->direct = D;
    auto give_up = [&](int rc, const char *why) { if (why) c.err = why; direct_free(A); return rc; };

    // 1. what I need from each owner; everybody learns everybody's needs
    std::vector<long long> need;
    int rc = need_ranges(A, P, need);
    bool all = false;
    int rc2 = xg_agree(rc == 0, &all);
    if (rc2) return give_up(rc2, nullptr);
    if (!all) return give_up(rc ? rc : LCG_HIP_E_COMM, rc ? nullptr : "direct exchange: a peer could not scan its columns");
    std::vector<unsigned long long> table;
    rc = xg_allgather_words(reinterpret_cast<const unsigned long long *>(need.data()), 2 * P, table);
    if (rc) return give_up(rc, nullptr);
    auto T = [&](int q, int p, int k) { return (long long)table[(size_t)q * 2 * P + 2 * p + k]; };   // what q needs from p
    // 2. neighbours: anybody I give to or take from (symmetric by construction)
    for (int q = 0; q < P; q++) {
        if (q == me) continue;
        const bool take = need[2 * q + 1] > need[2 * q];
        const bool give = T(q, me, 1) > T(q, me, 0);
        if (!take && !give) continue;
        if (D->nnb < XG_MAXSEG) {
            const int s = D->nnb;
            D->nb_rank[s] = q;
            D->give_lo[s] = give ? T(q, me, 0) : 0; D->give_hi[s] = give ? T(q, me, 1) : 0;
            D->need_lo[s] = take ? need[2 * q] : 0; D->need_hi[s] = take ? need[2 * q + 1] : 0;
            if (take) D->recv_total += need[2 * q + 1] - need[2 * q];
        }
        D->nnb++;
    }
    const bool fits = D->nnb <= XG_MAXSEG;
    rc = xg_agree(fits, &all);
    if (rc) return give_up(rc, nullptr);
    if (!all) return give_up(LCG_HIP_E_COMM, "direct exchange: a rank has more than 8 neighbours (use mode 1 or 0)");
    // 3. buffers and their IPC handles
    D->half = (size_t)A->rows_per_rank * P * w;
    hipError_t e;
    {
        void *p = nullptr;
        e = hipExtMallocWithFlags(&p, sizeof(double) * 2 * D->half, hipDeviceMallocUncached);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, sizeof(double) * 2 * D->half, hipDeviceMallocFinegrained); }
        D->recv = static_cast<double *>(p);
    }
    if (e == hipSuccess) e = hipMemsetAsync(D->recv, 0, sizeof(double) * 2 * D->half, c.stream);
    if (e == hipSuccess) {
        void *p = nullptr;
        e = hipExtMallocWithFlags(&p, sizeof(unsigned long long) * XG_MAXP, hipDeviceMallocUncached);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, sizeof(unsigned long long) * XG_MAXP, hipDeviceMallocFinegrained); }
        D->flags = static_cast<unsigned long long *>(p);
    }
    if (e == hipSuccess) e = hipMemsetAsync(D->flags, 0, sizeof(unsigned long long) * XG_MAXP, c.stream);
    if (e == hipSuccess) e = hipMalloc(&D->ticket, sizeof(unsigned int));
    if (e == hipSuccess) e = hipMemsetAsync(D->ticket, 0, sizeof(unsigned int), c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    unsigned long long hw[16] = {0};
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (e == hipSuccess) { hipIpcMemHandle_t h; e = hipIpcGetMemHandle(&h, D->recv); std::memcpy(hw, &h, 64); }
    if (e == hipSuccess) { hipIpcMemHandle_t h; e = hipIpcGetMemHandle(&h, D->flags); std::memcpy(hw + 8, &h, 64); }
    if (e != hipSuccess) (void)fail(e, "direct exchange buffers", __FILE__, __LINE__);
    rc = xg_agree(e == hipSuccess, &all);
    if (rc) return give_up(rc, nullptr);
    if (!all) return give_up(LCG_HIP_E_COMM, e == hipSuccess ? "direct exchange: a peer could not allocate or export its buffers" : nullptr);
    std::vector<unsigned long long> handles;
    rc = xg_allgather_words(hw, 16, handles);
    if (rc) return give_up(rc, nullptr);
    // 4. map the neighbours' buffers
    e = hipSuccess;
    for (int s = 0; s < D->nnb && e == hipSuccess; s++) {
        const int q = D->nb_rank[s];
        hipIpcMemHandle_t h; void *p = nullptr;
        std::memcpy(&h, &handles[(size_t)q * 16], 64);
        e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) break;
        D->opened.push_back(p); D->nb_recv[s] = static_cast<double *>(p);
        std::memcpy(&h, &handles[(size_t)q * 16 + 8], 64);
        e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) break;
        D->opened.push_back(p); D->nb_flags[s] = static_cast<unsigned long long *>(p);
    }
    if (e != hipSuccess) (void)fail(e, "hipIpcOpenMemHandle (neighbour buffers)", __FILE__, __LINE__);
    rc = xg_agree(e == hipSuccess, &all);
    if (rc) return give_up(rc, nullptr);
    if (!all) return give_up(LCG_HIP_E_COMM, e == hipSuccess ? "direct exchange: a peer could not map a neighbour" : nullptr);
    // 5. the per-call plans (pointers that depend on the call's parity are filled in dist_spmv)
    PushPlan &pp = D->push;
    pp.ticket = D->ticket;
    int nb = 0;
    for (int s = 0; s < D->nnb; s++) {
        pp.flag[pp.nflag++] = D->nb_flags[s] + me;
        D->wait.flag[D->wait.n++] = D->flags + D->nb_rank[s];
        const long long cnt = (D->give_hi[s] - D->give_lo[s]) * (long long)w;
        if (cnt <= 0) continue;
        pp.count[pp.nseg] = (long)cnt;
        pp.first_block[pp.nseg] = nb;
        nb += (int)((cnt + PUSH_CHUNK - 1) / PUSH_CHUNK);
        pp.nseg++;
    }
    pp.first_block[pp.nseg] = nb;
    pp.nblocks = nb > 0 ? nb : (pp.nflag > 0 ? 1 : 0);     // flags go out even when no data does
    D->wait.timeout_ticks = xb.timeout_ticks;
    D->wait.fail = xb.fail;
    D->generation = g_xg_generation;
    direct_copy_plan(A, D);
    return 0;
}

// per call: the pushing plan with this call's number and parity
static void direct_plans(lcg_hip_csr *A, const double *x, PushPlan *pp, WaitPlan *wp, PushPlan *cp)
{
    Direct *D = static_cast<Direct *>(A->direct);
    const size_t w = A->is_complex ? 2 : 1;
    const unsigned long long k = ++D->calls;
    const size_t par = (size_t)(k & 1);
    *pp = D->push; *wp = D->wait; *cp = D->copy;
    pp->seq = k; wp->seq = k;
    int seg = 0, cseg = 0;
    for (int s = 0; s < D->nnb; s++) {
        if (D->give_hi[s] > D->give_lo[s]) {
            pp->src[seg] = x + w * (size_t)(D->give_lo[s] - A->row0);
            pp->dst[seg] = D->nb_recv[s] + par * D->half + w * (size_t)D->give_lo[s];
            seg++;
        }
        if (D->need_hi[s] > D->need_lo[s]) cp->src[cseg++] = D->recv + par * D->half + w * (size_t)D->need_lo[s];
    }
}

void dist_free(lcg_hip_csr *A)
{
    direct_free(A);
    if (A->halo) { delete static_cast<HaloPlan *>(A->halo); A->halo = nullptr; }
    if (!A->distributed) return;
    free_part(A->loc); free_part(A->rem);
    if (A->remc.rowptr) hipFree(A->remc.rowptr);
    if (A->remc.pk_base) hipFree(A->remc.pk_base);      // packed columns, had the compacted part qualified (csr.hip)
    if (A->remc.pk_ofs) hipFree(A->remc.pk_ofs);
    if (A->remc.pk_data) hipFree(A->remc.pk_data);
    A->remc = CsrPart();
    if (A->rem_rows) hipFree(A->rem_rows);
    if (A->rem_y) hipFree(A->rem_y);
    A->rem_rows = nullptr; A->rem_y = nullptr;
    if (A->xfull) hipFree(A->xfull);
    A->xfull = nullptr; A->distributed = false;
    // transposes made for the sharded product have the padded global height: not the unsharded matrix's
    for (int i = 1; i < 4; i++) free_part(A->op[i]);
    if (A->op_z) hipFree(A->op_z);
    A->op_z = nullptr;
}

// Split `main` (global columns) of a shard whose rows are [row0, row0+n_rows) into loc/rem.
// Usable without a communicator (nranks given explicitly) so the split and the two-part
// product can be tested on one GPU.
int dist_split(lcg_hip_csr *A, int64_t n_global, int nranks, int rank)
{
    Ctx &c = ctx();
    for (int i = 1; i < 4; i++) free_part(A->op[i]);     // transposes of the unsharded matrix, if any: another shape from here on
    const int n = A->n_rows;
    const int64_t rpr = (n_global + nranks - 1) / nranks;
    const int64_t row0 = (int64_t)rank * rpr;
    const int64_t expect = std::max<int64_t>(0, std::min<int64_t>(n_global, row0 + rpr) - row0);
    if (expect != n) { c.err = "shard row count does not match ceil(n_global/nranks) partition"; return LCG_HIP_E_ARG; }
    dist_free(A);
    A->n_global = n_global; A->row0 = row0; A->rows_per_rank = rpr;
    int *cl = nullptr, *cr = nullptr;
    HIPCHK(hipMalloc(&cl, sizeof(int) * (size_t)n));
    HIPCHK(hipMalloc(&cr, sizeof(int) * (size_t)n));
    const unsigned g = (unsigned)((n + VB - 1) / VB);
    hipLaunchKernelGGL(k_split_count, dim3(g), dim3(VB), 0, c.stream, n, (long)row0, (long)(row0 + n), A->main.rowptr, A->main.col, cl, cr);
    long nl = 0, nr = 0;
    HIPCHK(hipMalloc(&A->loc.rowptr, sizeof(int) * ((size_t)n + 1)));
    HIPCHK(hipMalloc(&A->rem.rowptr, sizeof(int) * ((size_t)n + 1)));
    int rc = device_exclusive_scan(n, cl, A->loc.rowptr, c.stream, &nl);
    if (!rc) rc = device_exclusive_scan(n, cr, A->rem.rowptr, c.stream, &nr);
    hipFree(cl); hipFree(cr);
    if (rc) return rc;
    rc = alloc_cols(A->loc, n, nl, A->is_complex); if (rc) return rc;
    rc = alloc_cols(A->rem, n, nr, A->is_complex); if (rc) return rc;
    A->loc.n_cols = n; A->rem.n_cols = rpr * nranks;
    if (A->is_complex)
        hipLaunchKernelGGL((k_split_fill<double2>), dim3(g), dim3(VB), 0, c.stream, n, (long)row0, (long)(row0 + n), A->main.rowptr,
                           A->main.col, reinterpret_cast<const double2 *>(A->main.val), A->loc.rowptr, A->loc.col,
                           reinterpret_cast<double2 *>(A->loc.val), A->rem.rowptr, A->rem.col, reinterpret_cast<double2 *>(A->rem.val));
    else
        hipLaunchKernelGGL((k_split_fill<double>), dim3(g), dim3(VB), 0, c.stream, n, (long)row0, (long)(row0 + n), A->main.rowptr,
                           A->main.col, A->main.val, A->loc.rowptr, A->loc.col, A->loc.val, A->rem.rowptr, A->rem.col, A->rem.val);
    HIPCHK(hipGetLastError());
    if (nr > 0) {
        // the remote part touches few rows (the two ends of a banded shard): drop the empty ones, so
        // that its product costs what its entries cost and can run beside the local product
        int *flag = nullptr, *pos = nullptr;
        long kept = 0;
        HIPCHK(hipMalloc(&flag, sizeof(int) * (size_t)n));
        HIPCHK(hipMalloc(&pos, sizeof(int) * ((size_t)n + 1)));
        hipLaunchKernelGGL(k_row_flags, dim3(g), dim3(VB), 0, c.stream, n, A->rem.rowptr, flag);
        rc = device_exclusive_scan(n, flag, pos, c.stream, &kept);
        if (!rc) {
            hipError_t e = hipMalloc(&A->remc.rowptr, sizeof(int) * ((size_t)kept + 1));
            if (e == hipSuccess) e = hipMalloc(&A->rem_rows, sizeof(int) * (size_t)kept);
            if (e == hipSuccess) e = hipMalloc(&A->rem_y, sizeof(double) * (A->is_complex ? 2 : 1) * 2 * (size_t)kept);     // x2: the direct mode alternates halves
            if (e != hipSuccess) rc = fail(e, "remote-row compaction", __FILE__, __LINE__);
        }
        if (!rc) {
            hipLaunchKernelGGL(k_row_compact, dim3(g), dim3(VB), 0, c.stream, n, A->rem.rowptr, pos, A->rem_rows,
                               A->remc.rowptr, (int)kept);
            A->remc.n_rows = (int)kept; A->remc.nnz = nr; A->remc.col = A->rem.col; A->remc.val = A->rem.val;
            A->remc.owned = false; A->remc.padded = true;
        }
        hipError_t e2 = hipStreamSynchronize(c.stream);
        hipFree(flag); hipFree(pos);
        if (rc) return rc;
        if (e2 != hipSuccess) return fail(e2, "remote-row compaction", __FILE__, __LINE__);
    }
    HIPCHK(hipMalloc(&A->xfull, sizeof(double) * (A->is_complex ? 2 : 1) * (size_t)(rpr * nranks)));
    HIPCHK(hipMemsetAsync(A->xfull, 0, sizeof(double) * (A->is_complex ? 2 : 1) * (size_t)(rpr * nranks), c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    A->distributed = true;
    return 0;
}

// y = this rank's rows of op(A).x for a transposed A (clbicg's A^H.d, clcg.cpp:187): every rank multiplies the transpose of
// ITS rows with ITS slice of x -- a vector of the matrix's (padded) height -- and ncclReduceScatter(sum) hands each rank the
// sum of everybody's contributions to its row block.  Strong-scaling cost: the whole height crosses the links once per product
// (8 N bytes in total, like the all-gather of the north-star product).
int dist_spmv_op(lcg_hip_csr *A, const CsrPart &T, const double *x, double *y)
{
    Ctx &c = ctx();
    const size_t w = A->is_complex ? 2 : 1;
    const int *done = ax_flag(c);
    if (!g_comm.comm && world_size() > 1) {
        c.err = "op(A).x on a sharded matrix sums the ranks' contributions with RCCL: no communicator (lcg_hip_comm_init)";
        return LCG_HIP_E_COMM;
    }
    const size_t rpr = (size_t)A->rows_per_rank, nt = (size_t)T.n_rows;
    if (!A->op_z) HIPCHK(hipMalloc(&A->op_z, sizeof(double) * w * (nt + rpr)));
    double *z = A->op_z, *blk = A->op_z + w * nt;
    const double mean = T.n_rows ? (double)T.nnz / T.n_rows : 0.0;
    int rc = spmv_launch(T, A->is_complex, 0, mean, x, z, false, c.stream, done);
    if (rc) return rc;
    const bool whole = (size_t)A->n_rows == rpr;         // the last rank's block may be shorter than the reduce-scatter's
    double *dst = whole ? y : blk;
    if (g_comm.comm) {
        ncclResult_t r = g_comm.ReduceScatter(z, dst, w * rpr, ncclDouble, ncclSum, g_comm.comm, c.stream);
        if (r != ncclSuccess) return comm_fail("ncclReduceScatter", r);
    } else {
        HIPCHK(hipMemcpyAsync(dst, z + w * (size_t)A->row0, sizeof(double) * w * (size_t)A->n_rows, hipMemcpyDeviceToDevice, c.stream));
    }
    if (!whole) HIPCHK(hipMemcpyAsync(y, blk, sizeof(double) * w * (size_t)A->n_rows, hipMemcpyDeviceToDevice, c.stream));
    return 0;
}

// The compact sums of the remote-column rows (out[j] = sum_k val[k] * xfull[col[k]] of remc row j) with the chain-of-short-rows kernel
// above -- the all-gather / neighbour-range exchanges put this product BEHIND the exchange, on the critical path: the row-block
// kernels the automatic choice would take for such a part (262K rows of ~6 entries on the 8-way shard of the headline system) need
// 41 us for it, k_remote 10.
static int remote_sums_launch(lcg_hip_csr *A, const double *xfull, double *out, hipStream_t s, const int *done)
{
    const int nr = A->remc.n_rows;
    if (nr <= 0) return 0;
    const double mean_r = (double)A->remc.nnz / nr;
    const int T = mean_r <= 8.0 ? 1 : (mean_r <= 24.0 ? 2 : 4);
    const unsigned g = (unsigned)(((long)nr * T + VB - 1) / VB);
#define RS_LAUNCH(TT)                                                                                                \
    do {                                                                                                             \
        if (A->is_complex)                                                                                           \
            hipLaunchKernelGGL((k_remote<double2, TT, false>), dim3(g), dim3(VB), 0, s, nr, A->remc.rowptr, A->remc.col, \
                               reinterpret_cast<const double2 *>(A->remc.val), A->rem_rows,                          \
                               reinterpret_cast<const double2 *>(xfull), reinterpret_cast<double2 *>(out), done);    \
        else                                                                                                         \
            hipLaunchKernelGGL((k_remote<double, TT, false>), dim3(g), dim3(VB), 0, s, nr, A->remc.rowptr, A->remc.col, \
                               A->remc.val, A->rem_rows, xfull, out, done);                                          \
    } while (0)
    switch (T) { case 1: RS_LAUNCH(1); break; case 2: RS_LAUNCH(2); break; default: RS_LAUNCH(4); }
#undef RS_LAUNCH
    HIPCHK(hipGetLastError());
    return 0;
}

// The x exchange of modes 0 and 1 on stream s: the neighbour ranges (grouped ncclSend / ncclRecv straight into the gather buffer at
// their global offsets), or this rank's slice copied into its place in the gather buffer and ncclAllGather in place.
static int exchange_x(lcg_hip_csr *A, const double *x, hipStream_t s)
{
    const size_t w = A->is_complex ? 2 : 1;
    if (g_comm.comm && A->dist_mode == 1 && A->halo) return halo_exchange(A, x, s);    // (the local product reads x itself)
    double *mine = A->xfull + w * (size_t)(A->row0);
    HIPCHK(hipMemcpyAsync(mine, x, sizeof(double) * w * (size_t)A->n_rows, hipMemcpyDeviceToDevice, s));
    if (g_comm.comm) {
        ncclResult_t r = g_comm.AllGather(mine, A->xfull, w * (size_t)A->rows_per_rank, ncclDouble, g_comm.comm, s);
        if (r != ncclSuccess) return comm_fail("ncclAllGather", r);
    }
    return 0;
}

// u != nullptr: the product also leaves y.u as partial sums in part[0 .. *slots) -- the local product's (folded to <= 512), then one per
// block of the remote-column kernel.  *fused says whether it did (when not, the plain product was made).
static int dist_spmv_impl(lcg_hip_csr *A, const double *x, double *y, const double *u, double *part, int *slots, bool *fused)
{
    Ctx &c = ctx();
    if (fused) *fused = false;
    bool dot = u != nullptr && !A->is_complex;
    int nslot = 0;
    const int *done = ax_flag(c);
    if (A->dist_mode < 0) {
        c.err = "this matrix's exchange could not be set up (lcg_hip_csr_distribute failed): distribute it again under another mode";
        return LCG_HIP_E_COMM;
    }
    if (A->dist_mode != 2 && !g_comm.comm && world_size() > 1) {
        // mailbox-only world: modes 0 and 1 have nothing to move x with -- a product over a stale gather buffer is not an answer
        c.err = "modes 0 and 1 move x with RCCL: no communicator (lcg_hip_comm_init)";
        return LCG_HIP_E_COMM;
    }
    if (A->dist_mode == 2 && !A->direct) {
        c.err = "direct exchange without a plan (lcg_hip_csr_distribute(A, n, 2) did not succeed)";
        return LCG_HIP_E_COMM;
    }
    if (A->dist_mode == 2 && A->direct) {
        // One stream, no collective, no event: [pushing blocks | local product | receiving blocks] | remote-column product.
        // (Measured and retired, LAB_NOTES / DESIGN 9: the receiving kernels on a second stream behind a join event -- equal, 114.2 vs
        //  114.6 us on the 8-way shard; the remote-column product gathering straight from the landing zone -- 4 us faster with one
        //  rank per GPU, but every block of that grid spins on the neighbours' flags and ranks that share a GPU starve each other;
        //  the scalar step in the last block of the remote-column kernel -- nothing, 101.2 vs 100.4 us per iteration.)
        if (static_cast<Direct *>(A->direct)->uses_mailbox &&
            (!g_xg.connected || static_cast<Direct *>(A->direct)->generation != g_xg_generation)) {
            c.err = "direct exchange: the mailboxes were disconnected (or connected anew) while this matrix still uses them (distribute it again)";
            return LCG_HIP_E_COMM;
        }
        PushPlan pp, cp; WaitPlan wp;
        direct_plans(A, x, &pp, &wp, &cp);
        const double mean_l = A->n_rows ? (double)A->loc.nnz / A->n_rows : 0.0;
        // test hook (tests/test_gpu_direct.py): this rank computes but never pushes -- what a dead link looks like to its neighbours
        static const bool withhold = std::getenv("LCG_HIP_TEST_WITHHOLD_PUSH") != nullptr;
        // The receiving blocks ride in the tail of the product's grid (devcommon.hpp: recv_block) instead of being a kernel of their
        // own behind it (k_recv; 84.5 -> 81.3 us per A.x on the self-loop shard): one launch and one kernel boundary less per product,
        // and the copies run beside the product's last blocks.  FORWARD PROGRESS: a receiving block spins until the neighbours'
        // pushing blocks have run, and pushing blocks wait for nobody -- they sit at the FRONT of their grid, receiving blocks at the
        // END of theirs.  Work-groups are dispatched in index order (observed on every gfx9 part; HIP does not promise it), so by
        // the time a receiving block holds a CU slot every pushing block of its own grid has been dispatched; and should dispatch
        // ever be reordered, a wait could only become a dead-lock if the receiving blocks of ONE grid filled every slot of the GPU
        // -- hence the bound below: at most RECV_TAIL_MAX receiving blocks (the chip holds >= 1280 work-groups of this size), the
        // receive as a kernel of its own beyond it (as under the withhold hook, whose product has no pushing grid).  Every wait
        // ends in a time-out in any case (LCG_HIP_P2P_TIMEOUT_MS), which stops all ranks' solves with LCG_HIP_E_COMM.
        constexpr int RECV_TAIL_MAX = 256;
        const bool recv_in_tail = !withhold && wp.n > 0 && cp.nblocks <= RECV_TAIL_MAX;
        if (recv_in_tail) {
            pp.nrecv = cp.nblocks; pp.rnseg = cp.nseg; pp.wp = wp; pp.rst = c.in_solve ? c.state : nullptr;
            for (int q = 0; q < cp.nseg; q++) { pp.rsrc[q] = cp.src[q]; pp.rdst[q] = cp.dst[q]; pp.rcount[q] = cp.count[q]; }
            for (int q = 0; q <= cp.nseg && q <= XG_MAXSEG; q++) pp.rfirst[q] = cp.first_block[q];
            if (pp.nrecv <= 0) pp.nrecv = 1;        // flag-only neighbours: one block awaits the flags
        }
        // the remote part's share of the dot rides in k_remote; its blocks must fit behind the local sums
        const long rem_blocks = A->remc.n_rows > 0 ? ((long)A->remc.n_rows * 4 + VB - 1) / VB : 0;
        dot = dot && !withhold && 512 + rem_blocks <= AXP_CAP;
        int rc = 0, f = 0;
        int nbig = 0;       // per-block sums of the local product that k_remote folds (0: the local product folded them itself)
        if (dot) {
            f = csr_part_ax_dot(A->loc, A->variant, mean_l, x, y, u, 0, part, &nslot, c.stream, done, &pp, A->remc.n_rows > 0 ? &nbig : nullptr);
            if (f < 0) return f;
        }
        dot = f == 1;
        if (!dot)
            rc = withhold ? spmv_launch(A->loc, A->is_complex, A->variant, mean_l, x, y, false, c.stream, done)
                          : spmv_launch_push(A->loc, A->is_complex, A->variant, mean_l, x, y, c.stream, done, pp);
        if (rc) return rc;
        Direct *D = static_cast<Direct *>(A->direct);
        if (debug_on() && D->calls == 1)
            std::fprintf(stderr, "[lcg_hip] direct: recv %lld doubles in %d blocks (%s), n_global %lld, rows %d\n", D->recv_total, cp.nblocks,
                         recv_in_tail ? "tail of the product's grid" : "k_recv", (long long)A->n_global, A->n_rows);
        if (wp.n > 0 && !recv_in_tail) {    // flags are awaited even after a stop: the neighbours' calls stay paired with mine
            hipLaunchKernelGGL(k_recv, dim3((unsigned)cp.nblocks), dim3(VB), 0, c.stream, wp, cp, c.in_solve ? c.state : nullptr);
            HIPCHK(hipGetLastError());
        }
        if (A->remc.n_rows > 0) {
            const int nr = A->remc.n_rows;
            const double mean_r = (double)A->remc.nnz / nr;
            const int T = mean_r <= 8.0 ? 1 : (mean_r <= 24.0 ? 2 : 4);
#define REMOTE_LAUNCH(TT)                                                                                            \
    do {                                                                                                             \
        const unsigned g = (unsigned)(((long)nr * TT + VB - 1) / VB);                                                \
        if (A->is_complex)                                                                                           \
            hipLaunchKernelGGL((k_remote<double2, TT, true>), dim3(g), dim3(VB), 0, c.stream, nr, A->remc.rowptr, A->remc.col, \
                               reinterpret_cast<const double2 *>(A->remc.val), A->rem_rows,                          \
                               reinterpret_cast<const double2 *>(A->xfull), reinterpret_cast<double2 *>(y), done);   \
        else                                                                                                         \
            hipLaunchKernelGGL((k_remote<double, TT, true>), dim3(g), dim3(VB), 0, c.stream, nr, A->remc.rowptr, A->remc.col, \
                               A->remc.val, A->rem_rows, A->xfull, y, done);                                         \
    } while (0)
#define REMOTE_DOT(TT)                                                                                               \
    do {                                                                                                             \
        const unsigned g = (unsigned)(((long)nr * TT + VB - 1) / VB);                                                \
        hipLaunchKernelGGL((k_remote<double, TT, true, true>), dim3(g), dim3(VB), 0, c.stream, nr, A->remc.rowptr, A->remc.col, \
                           A->remc.val, A->rem_rows, A->xfull, y, done, u, part + nslot,                             \
                           nbig ? A->loc.dot_part : nullptr, nbig, (int)((nbig + g - 1) / g));                       \
        nslot += (int)g;                                                                                             \
    } while (0)
#define REMOTE_CASE(TT) case TT: if (dot) REMOTE_DOT(TT); else REMOTE_LAUNCH(TT); break;
            switch (T) {
                REMOTE_CASE(1) REMOTE_CASE(2) REMOTE_CASE(4)
            }
#undef REMOTE_CASE
#undef REMOTE_DOT
#undef REMOTE_LAUNCH
            HIPCHK(hipGetLastError());
        }
        if (dot) { *slots = nslot; *fused = true; }
        return 0;
    }
    // gather on the second stream ...
    HIPCHK(hipEventRecord(c.ev_a, c.stream));
    HIPCHK(hipStreamWaitEvent(c.comm_stream, c.ev_a, 0));
    int rc = exchange_x(A, x, c.comm_stream);
    if (rc) return rc;
    // ... followed there by the product of the remote columns (few rows: their sums go to rem_y) ...
    if (A->remc.n_rows > 0) {
        rc = remote_sums_launch(A, A->xfull, A->rem_y, c.comm_stream, done);
        if (rc) return rc;
    }
    HIPCHK(hipEventRecord(c.ev_b, c.comm_stream));
    // ... while the locally-owned columns are multiplied here; then y[row] += rem_y
    const double mean_l = A->n_rows ? (double)A->loc.nnz / A->n_rows : 0.0;
    int nbig01 = 0;
    {
        const long rem_blocks = (A->remc.n_rows + VB - 1) / VB;
        dot = dot && 512 + rem_blocks <= AXP_CAP;
        int f = 0;
        if (dot) {
            f = csr_part_ax_dot(A->loc, A->variant, mean_l, x, y, u, 0, part, &nslot, c.stream, done, nullptr, A->remc.n_rows > 0 ? &nbig01 : nullptr);
            if (f < 0) return f;
        }
        dot = f == 1;
    }
    if (!dot) rc = spmv_launch(A->loc, A->is_complex, A->variant, mean_l, x, y, false, c.stream, done);
    if (rc) return rc;
    HIPCHK(hipStreamWaitEvent(c.stream, c.ev_b, 0));
    if (dot) {
        if (A->remc.n_rows > 0) {
            const unsigned g = (unsigned)((A->remc.n_rows + VB - 1) / VB);
            hipLaunchKernelGGL(k_scatter_add_dot, dim3(g), dim3(VB), 0, c.stream, A->remc.n_rows, A->rem_rows, A->rem_y, y, u, part + nslot,
                               nbig01 ? A->loc.dot_part : nullptr, nbig01, (int)((nbig01 + g - 1) / g), done);
            HIPCHK(hipGetLastError());
            nslot += (int)g;
        }
        *slots = nslot; *fused = true;
        return 0;
    }
    if (A->remc.n_rows > 0) {
        const unsigned g = (unsigned)((A->remc.n_rows + VB - 1) / VB);
        if (A->is_complex)
            hipLaunchKernelGGL((k_scatter_add<double2>), dim3(g), dim3(VB), 0, c.stream, A->remc.n_rows, A->rem_rows,
                               reinterpret_cast<const double2 *>(A->rem_y), reinterpret_cast<double2 *>(y), done);
        else
            hipLaunchKernelGGL((k_scatter_add<double>), dim3(g), dim3(VB), 0, c.stream, A->remc.n_rows, A->rem_rows,
                               A->rem_y, y, done);
        HIPCHK(hipGetLastError());
    }
    return rc;
}

int dist_spmv(lcg_hip_csr *A, const double *x, double *y) { return dist_spmv_impl(A, x, y, nullptr, nullptr, nullptr, nullptr); }

int dist_ax_dot(lcg_hip_csr *A, const double *x, double *y, const double *u, int yy, double *part, int *slots)
{
    static const bool off = [] { const char *e = lab_env("LCG_HIP_AX_DOT_SHARDED"); return e && atoi(e) == 0; }();     // A/B runs (LAB build)
    bool fused = false;
    const int rc = dist_spmv_impl(A, x, y, (yy || off) ? nullptr : u, part, slots, &fused);
    return rc ? (rc > 0 ? -rc : rc) : (fused ? 1 : 2);
}

} // namespace lcgh

using namespace lcgh;

extern "C" {

int lcg_hip_comm_unique_id(void *id128)
{
    int rc = load_rccl(); if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = g_comm.GetUniqueId(&id);
    if (r != ncclSuccess) return comm_fail("ncclGetUniqueId", r);
    static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
    std::memcpy(id128, &id, 128);
    return 0;
}

int lcg_hip_comm_init(int nranks, int rank, const void *id128)
{
    int rc = ensure_init(); if (rc) return rc;
    rc = load_rccl(); if (rc) return rc;
    if (g_comm.comm) return 0;
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    ncclResult_t r = g_comm.CommInitRank(&g_comm.comm, nranks, id, rank);
    if (r != ncclSuccess) { g_comm.comm = nullptr; return comm_fail("ncclCommInitRank", r); }
    g_comm.nranks = nranks; g_comm.rank = rank;
    g_comm.force = std::getenv("LCG_HIP_FORCE_COMM") != nullptr;
    // Do two ranks of this job sit on one device?  (The real RCCL refuses that; a stand-in bound through LCG_HIP_RCCL_LIB does not.)  The
    // ranks' device UUIDs meet in the communicator's first all-gather; a rank that shares its device makes no placement walks
    // (driver.hpp) -- what a walk holds, the rank next door cannot have.
    if (nranks > 1) {
        Ctx &c = ctx();
        hipUUID id;
        std::memset(&id, 0, sizeof id);
        static_assert(sizeof(hipUUID) == 16, "device UUID size");
        if (hipDeviceGetUuid(&id, c.device) != hipSuccess) { (void)hipGetLastError(); std::memset(&id, 0, sizeof id); }
        char *d = nullptr;
        HIPCHK(hipMalloc(&d, 16 * (size_t)nranks));
        std::vector<char> all(16 * (size_t)nranks);
        hipError_t e = hipMemcpyAsync(d + 16 * (size_t)rank, &id, 16, hipMemcpyHostToDevice, c.stream);
        if (e == hipSuccess) {
            r = g_comm.AllGather(d + 16 * (size_t)rank, d, 16, ncclChar, g_comm.comm, c.stream);
            if (r != ncclSuccess) { hipFree(d); return comm_fail("ncclAllGather (device identities)", r); }
            e = hipMemcpyAsync(all.data(), d, all.size(), hipMemcpyDeviceToHost, c.stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
        hipFree(d);
        if (e != hipSuccess) return fail(e, "device identities", __FILE__, __LINE__);
        for (int q = 0; q < nranks; q++)
            if (q != rank && std::memcmp(&all[16 * (size_t)q], &id, 16) == 0) c.ranks_share_device = true;
        if (c.ranks_share_device && debug_on()) std::fprintf(stderr, "[lcg_hip] rank %d shares its device with another rank of the communicator\n", rank);
    }
    return 0;
}

int lcg_hip_comm_destroy(void)
{
    if (g_comm.comm) { g_comm.CommDestroy(g_comm.comm); g_comm.comm = nullptr; }
    g_comm.nranks = 1; g_comm.rank = 0;
    return 0;
}

int lcg_hip_comm_rank(void) { return g_comm.rank; }
const char *lcg_hip_comm_library(void) { return g_comm.lib ? g_comm.path.c_str() : ""; }
int lcg_hip_comm_size(void) { return g_comm.nranks; }

int lcg_hip_csr_distribute(lcg_hip_csr_t A, int64_t n_global, int mode)
{
    if (!A || n_global <= 0) return LCG_HIP_E_ARG;
    if (mode < 0 || mode > 2) return LCG_HIP_E_ARG;
    if (mode != 2 && !g_comm.comm && world_size() > 1) {
        ctx().err = "modes 0 and 1 move x with RCCL: no communicator (lcg_hip_comm_init)";
        return LCG_HIP_E_COMM;
    }
    // the mode is recorded only once the split and the plan stand; until then (and after a failure) the handle
    // refuses products (dist_spmv) instead of running an exchange it does not have
    int rc = dist_split(A, n_global, world_size(), world_rank());
    if (rc) return rc;              // a rejected shard leaves the handle as it was (dist_split validates before it frees)
    A->dist_mode = -1;
    if (mode == 1 && g_comm.comm) rc = halo_setup(A);
    if (mode == 2) rc = direct_setup(A);
    if (rc) return rc;              // split stays; the caller distributes again under another mode
    A->dist_mode = mode;
    return 0;
}

// entries of x this rank receives per A.x (doubles; complex counts twice): plan volume
int64_t lcg_hip_csr_exchange_volume(lcg_hip_csr_t A)
{
    if (!A || !A->distributed) return 0;
    if (A->dist_mode == 1 && A->halo) return static_cast<const HaloPlan *>(A->halo)->recv_total;
    if (A->dist_mode == 2 && A->direct) return static_cast<const Direct *>(A->direct)->recv_total;
    return (int64_t)A->rows_per_rank * (world_size() - 1);
}

// test hook: the [lo,hi) column range needed from each of `nranks` owners (after split_for_test)
int lcg_hip_csr_need_ranges_for_test(lcg_hip_csr_t A, int nranks, int64_t *lohi)
{
    if (!A || !A->distributed || !lohi) return LCG_HIP_E_ARG;
    std::vector<long long> v;
    int rc = need_ranges(A, nranks, v);
    if (rc) return rc;
    for (int i = 0; i < 2 * nranks; i++) lohi[i] = v[i];
    return 0;
}

// test hook: split a shard as rank `rank` of `nranks` without any communicator; the caller
// fills the gather buffer itself through lcg_hip_csr_xfull()
int lcg_hip_csr_split_for_test(lcg_hip_csr_t A, int64_t n_global, int nranks, int rank)
{
    if (!A) return LCG_HIP_E_ARG;
    return dist_split(A, n_global, nranks, rank);
}

// Test/measurement hook (one GPU, no peers): the direct exchange with every neighbour replaced by
// this rank itself.  The pushing blocks write mirror-sized ranges of x into the rank's OWN receive
// buffer (at its own rows' offsets, which the remote-column product never reads) and raise the
// flags the product then waits for; both halves of the buffer are pre-filled from the gather
// buffer (lcg_hip_csr_xfull), so the product is the true one and the kernel chain is the real one.
int lcg_hip_csr_direct_selfloop_for_test(lcg_hip_csr_t A, int nranks, int rank)
{
    if (!A || !A->distributed) return LCG_HIP_E_ARG;
    Ctx &c = ctx();
    direct_free(A);
    std::vector<long long> need;
    int rc = need_ranges(A, nranks, need);
    if (rc) return rc;
    Direct *D = new Direct();
    A->direct = D;
    const size_t w = A->is_complex ? 2 : 1;
    D->half = (size_t)A->rows_per_rank * nranks * w;
    {
        void *pr = nullptr;
        HIPCHK(hipExtMallocWithFlags(&pr, sizeof(double) * 2 * D->half, hipDeviceMallocUncached));
        D->recv = static_cast<double *>(pr);
    }
    HIPCHK(hipMemcpyAsync(D->recv, A->xfull, sizeof(double) * D->half, hipMemcpyDeviceToDevice, c.stream));
    HIPCHK(hipMemcpyAsync(D->recv + D->half, A->xfull, sizeof(double) * D->half, hipMemcpyDeviceToDevice, c.stream));
    void *p = nullptr;
    HIPCHK(hipExtMallocWithFlags(&p, sizeof(unsigned long long) * XG_MAXP, hipDeviceMallocUncached));
    D->flags = static_cast<unsigned long long *>(p);
    HIPCHK(hipMemsetAsync(D->flags, 0, sizeof(unsigned long long) * XG_MAXP, c.stream));
    HIPCHK(hipMalloc(&D->ticket, sizeof(unsigned int)));
    HIPCHK(hipMemsetAsync(D->ticket, 0, sizeof(unsigned int), c.stream));
    static int fail_flag_host = 0;
    static int *fail_dev = nullptr;
    if (!fail_dev) { HIPCHK(hipMalloc(&fail_dev, sizeof(int))); HIPCHK(hipMemcpy(fail_dev, &fail_flag_host, sizeof(int), hipMemcpyHostToDevice)); }
    PushPlan &pp = D->push;
    pp.ticket = D->ticket;
    int nb = 0;
    for (int q = 0; q < nranks && D->nnb < XG_MAXSEG; q++) {
        if (q == rank || need[2 * q + 1] <= need[2 * q]) continue;
        const int sidx = D->nnb++;
        const long long cnt = std::min<long long>(need[2 * q + 1] - need[2 * q], A->n_rows);
        D->nb_rank[sidx] = q; D->nb_recv[sidx] = D->recv; D->nb_flags[sidx] = D->flags;
        D->give_lo[sidx] = q < rank ? A->row0 : A->row0 + A->n_rows - cnt;
        D->give_hi[sidx] = D->give_lo[sidx] + cnt;
        D->need_lo[sidx] = need[2 * q]; D->need_hi[sidx] = need[2 * q + 1];
        D->recv_total += cnt;
        pp.flag[pp.nflag++] = D->flags + q;             // "my word at the neighbour" = the word I wait on
        D->wait.flag[D->wait.n++] = D->flags + q;
        pp.count[pp.nseg] = (long)(cnt * (long long)w);
        pp.first_block[pp.nseg] = nb;
        nb += (int)((cnt * (long long)w + PUSH_CHUNK - 1) / PUSH_CHUNK);
        pp.nseg++;
    }
    pp.first_block[pp.nseg] = nb;
    pp.nblocks = nb > 0 ? nb : (pp.nflag > 0 ? 1 : 0);
    D->wait.timeout_ticks = 200000000LL;
    D->wait.fail = fail_dev;
    D->uses_mailbox = false;
    direct_copy_plan(A, D);
    HIPCHK(hipStreamSynchronize(c.stream));
    A->dist_mode = 2;
    return 0;
}
// Measurement hook (bench.py: comm_probe): ONE PART of a sharded product alone, on the streams the product itself uses, so that a
// multi-GPU line can say where an iteration's time goes.  part 1: the x exchange of modes 0 / 1 (second stream, fork and join
// included); 2: the local-column product (the kernel the loop runs, no dot carried); 4: the remote-column part (sums of the rows that
// hold remote columns out of whatever the gather buffer holds, and their addition to y).  Collective for part 1.
int lcg_hip_csr_ax_part_for_probe(lcg_hip_csr_t A, const double *x, double *y, int part)
{
    if (!A || !A->distributed || !x || !y) return LCG_HIP_E_ARG;
    Ctx &c = ctx();
    const int *done = nullptr;
    if (part == 1) {
        if (A->dist_mode != 0 && A->dist_mode != 1) { c.err = "the direct exchange rides in the product's own grid: it has no part to time alone"; return LCG_HIP_E_ARG; }
        HIPCHK(hipEventRecord(c.ev_a, c.stream));
        HIPCHK(hipStreamWaitEvent(c.comm_stream, c.ev_a, 0));
        int rc = exchange_x(A, x, c.comm_stream);
        if (rc) return rc;
        HIPCHK(hipEventRecord(c.ev_b, c.comm_stream));
        HIPCHK(hipStreamWaitEvent(c.stream, c.ev_b, 0));
        return 0;
    }
    if (part == 2) {
        const double mean_l = A->n_rows ? (double)A->loc.nnz / A->n_rows : 0.0;
        return spmv_launch(A->loc, A->is_complex, A->variant, mean_l, x, y, false, c.stream, done);
    }
    if (part == 4) {
        if (A->remc.n_rows <= 0) return 0;
        int rc = remote_sums_launch(A, A->xfull, A->rem_y, c.stream, done);
        if (rc) return rc;
        const unsigned g = (unsigned)((A->remc.n_rows + VB - 1) / VB);
        if (A->is_complex)
            hipLaunchKernelGGL((k_scatter_add<double2>), dim3(g), dim3(VB), 0, c.stream, A->remc.n_rows, A->rem_rows,
                               reinterpret_cast<const double2 *>(A->rem_y), reinterpret_cast<double2 *>(y), done);
        else
            hipLaunchKernelGGL((k_scatter_add<double>), dim3(g), dim3(VB), 0, c.stream, A->remc.n_rows, A->rem_rows, A->rem_y, y, done);
        HIPCHK(hipGetLastError());
        return 0;
    }
    return LCG_HIP_E_ARG;
}
double *lcg_hip_csr_xfull(lcg_hip_csr_t A) { return A ? A->xfull : nullptr; }
int64_t lcg_hip_csr_local_nnz(lcg_hip_csr_t A) { return A ? A->loc.nnz : 0; }

int lcg_hip_allreduce_sum(double *dev_values, int count)
{
    int rc = ensure_init(); if (rc) return rc;
    return comm_allreduce(dev_values, count, ctx().stream);
}

int lcg_hip_p2p_export(void *handle64)
{
    int rc = ensure_init(); if (rc) return rc;
    if (!handle64) return LCG_HIP_E_ARG;
    if (g_xg.connected) { ctx().err = "direct all-reduce already connected"; return LCG_HIP_E_ARG; }
    if (!g_xg.mine) {
        // the mailbox is written by peers over the fabric while a local kernel polls it: it must not
        // live in this GPU's L2 as ordinary (coarse-grained) memory does
        void *p = nullptr;
        hipError_t e = hipExtMallocWithFlags(&p, XG_BYTES, hipDeviceMallocUncached);
        g_xg.mem_kind = "uncached";
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipExtMallocWithFlags(&p, XG_BYTES, hipDeviceMallocFinegrained);
            g_xg.mem_kind = "fine-grained";
        }
        if (e != hipSuccess) return fail(e, "mailbox allocation (uncached / fine-grained device memory)", __FILE__, __LINE__);
        g_xg.mine = static_cast<double *>(p);
        HIPCHK(hipMemset(g_xg.mine, 0, XG_BYTES));
        HIPCHK(hipDeviceSynchronize());
    }
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) == LCG_HIP_P2P_HANDLE_BYTES, "IPC handle size");
    HIPCHK(hipIpcGetMemHandle(&h, g_xg.mine));
    std::memcpy(handle64, &h, sizeof h);
    return 0;
}

int lcg_hip_p2p_disconnect(void)
{
    if (ctx().inited) hipDeviceSynchronize();
    for (void *p : g_xg.opened) hipIpcCloseMemHandle(p);
    g_xg.opened.clear();
    if (g_xg.peers_dev) hipFree(g_xg.peers_dev);
    if (g_xg.seq) hipFree(g_xg.seq);
    if (g_xg.fail) hipFree(g_xg.fail);
    if (g_xg.mine) hipFree(g_xg.mine);
    g_xg = Xg();
    return 0;
}

int lcg_hip_p2p_connect(int nranks, int rank, const void *handles)
{
    int rc = ensure_init(); if (rc) return rc;
    if (nranks < 1 || nranks > XG_MAXP || rank < 0 || rank >= nranks || !handles) return LCG_HIP_E_ARG;
    if (!g_xg.mine) { ctx().err = "lcg_hip_p2p_connect before lcg_hip_p2p_export"; return LCG_HIP_E_ARG; }
    if (g_xg.connected) return 0;
    std::vector<double *> peers((size_t)nranks, nullptr);
    for (int q = 0; q < nranks; q++) {
        if (q == rank) { peers[q] = g_xg.mine; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const char *>(handles) + (size_t)q * sizeof h, sizeof h);
        void *p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (void *o : g_xg.opened) hipIpcCloseMemHandle(o);
            g_xg.opened.clear();
            return fail(e, "hipIpcOpenMemHandle (peer mailbox)", __FILE__, __LINE__);
        }
        g_xg.opened.push_back(p);
        peers[q] = static_cast<double *>(p);
    }
    HIPCHK(hipMalloc(&g_xg.peers_dev, sizeof(double *) * (size_t)nranks));
    HIPCHK(hipMalloc(&g_xg.seq, sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&g_xg.fail, sizeof(int)));
    HIPCHK(hipMemcpy(g_xg.peers_dev, peers.data(), sizeof(double *) * (size_t)nranks, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(g_xg.seq, 0, sizeof(unsigned long long)));
    HIPCHK(hipMemset(g_xg.fail, 0, sizeof(int)));
    HIPCHK(hipDeviceSynchronize());
    if (const char *e = std::getenv("LCG_HIP_P2P_TIMEOUT_MS")) g_xg.timeout_ticks = std::max(1LL, atoll(e)) * 100000LL;
    g_xg.P = nranks; g_xg.me = rank; g_xg.connected = true;
    g_xg_generation++;
    return 0;
}

// `rounds` all-reduces of values every rank can predict, with a 2 s timeout.  All ranks must call
// it together (it advances the shared sequence).  A failure leaves the path disabled for good.
int lcg_hip_p2p_selftest(int rounds)
{
    int rc = ensure_init(); if (rc) return rc;
    if (!g_xg.connected) { ctx().err = "direct all-reduce not connected"; return LCG_HIP_E_COMM; }
    Ctx &c = ctx();
    const int P = g_xg.P, me = g_xg.me;
    if (rounds < 1) rounds = 1;
    double *dv = nullptr; int *dok = nullptr;
    HIPCHK(hipMalloc(&dv, sizeof(double) * MAXR * (size_t)rounds));
    HIPCHK(hipMalloc(&dok, sizeof(int) * (size_t)rounds));
    std::vector<double> hv((size_t)MAXR * rounds);
    std::vector<int> hok((size_t)rounds, 0);
    for (int i = 0; i < rounds; i++)
        for (int r = 0; r < MAXR; r++) hv[(size_t)i * MAXR + r] = (me + 1.0) * (i + 1.0) + 0.125 * r;
    HIPCHK(hipMemcpyAsync(dv, hv.data(), sizeof(double) * hv.size(), hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipMemsetAsync(dok, 0, sizeof(int) * (size_t)rounds, c.stream));
    XgBox xb;
    xb.mine = g_xg.mine; xb.peers = g_xg.peers_dev; xb.seq = g_xg.seq; xb.fail = g_xg.fail;
    xb.timeout_ticks = std::min(g_xg.timeout_ticks, 200000000LL); xb.P = P; xb.me = me;
    for (int i = 0; i < rounds; i++)
        hipLaunchKernelGGL(k_xg_allreduce, dim3(1), dim3(VB), 0, c.stream, dv + (size_t)i * MAXR, MAXR, xb, dok + i);
    hipError_t e = hipMemcpyAsync(hv.data(), dv, sizeof(double) * hv.size(), hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(hok.data(), dok, sizeof(int) * (size_t)rounds, hipMemcpyDeviceToHost, c.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c.stream);
    hipFree(dv); hipFree(dok);
    if (e != hipSuccess) return fail(e, "direct all-reduce self-test", __FILE__, __LINE__);
    for (int i = 0; i < rounds; i++) {
        bool good = hok[i] == 1;
        for (int r = 0; r < MAXR && good; r++) {
            const double expect = 0.5 * P * (P + 1.0) * (i + 1.0) + 0.125 * r * P;   // exact in fp64
            good = hv[(size_t)i * MAXR + r] == expect;
        }
        if (!good) {
            char buf[160];
            std::snprintf(buf, sizeof buf, "direct all-reduce self-test failed in round %d (%s mailbox, %d ranks)", i, g_xg.mem_kind, P);
            c.err = buf;
            g_xg.enabled = false;
            return LCG_HIP_E_COMM;
        }
    }
    // the same question as at lcg_hip_comm_init, for jobs that only have the mailboxes: do two ranks sit on one device?
    if (P > 1) {
        hipUUID id;
        std::memset(&id, 0, sizeof id);
        if (hipDeviceGetUuid(&id, c.device) != hipSuccess) { (void)hipGetLastError(); std::memset(&id, 0, sizeof id); }
        unsigned long long w[2];
        std::memcpy(w, &id, 16);
        std::vector<unsigned long long> all;
        const bool was = g_xg.enabled;
        g_xg.enabled = true;        // (xg_allgather_words goes through xg_box)
        int rc2 = xg_allgather_words(w, 2, all);
        g_xg.enabled = was;
        if (rc2) return rc2;
        for (int q = 0; q < P; q++)
            if (q != me && all[2 * (size_t)q] == w[0] && all[2 * (size_t)q + 1] == w[1]) c.ranks_share_device = true;
    }
    return 0;
}

int lcg_hip_p2p_enable(int on)
{
    if (on && !g_xg.connected) { ctx().err = "direct all-reduce not connected"; return LCG_HIP_E_COMM; }
    g_xg.enabled = on != 0;
    return 0;
}

int lcg_hip_p2p_set_timeout_ms(int ms)
{
    if (ms < 1) return LCG_HIP_E_ARG;
    g_xg.timeout_ticks = (long long)ms * 100000LL;      // picked up by the next exchange; direct plans copy it when they are made
    return 0;
}

int lcg_hip_p2p_status(void)
{
    if (!g_xg.connected) return 0;
    // a timed-out exchange (mailbox or neighbour flags) leaves the failure flag up: the path is dead
    int f = 0;
    if (g_xg.fail && ctx().inited) {
        if (hipMemcpyAsync(&f, g_xg.fail, sizeof f, hipMemcpyDeviceToHost, ctx().stream) != hipSuccess ||
            hipStreamSynchronize(ctx().stream) != hipSuccess) { (void)hipGetLastError(); f = 1; }
    }
    if (f) return -1;
    return g_xg.enabled ? 2 : 1;
}

int lcg_hip_barrier(void)
{
    int rc = ensure_init(); if (rc) return rc;
    Ctx &c = ctx();
    if (g_comm.comm || g_xg.enabled) { rc = comm_allreduce(c.state->red + MAXR - 1, 1, c.stream); if (rc) return rc; }
    HIPCHK(hipStreamSynchronize(c.stream));
    return 0;
}

} // extern "C"
