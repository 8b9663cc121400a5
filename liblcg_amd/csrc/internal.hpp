// internal.hpp -- shared declarations of liblcg_hip (not installed).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "lcg_hip.h"

namespace lcgh {

// Run-time switches.  The shipped library reads the documented ones only ("Run-time switches": LCG_HIP_PACKED,
// _TILED, _BINNED, _RANGES, _AX_DOT, _NT_VECTORS, _PACKED_WINDOW, _P2P_TIMEOUT_MS, _FORCE_COMM, _DEBUG, _PLACE,
// _TEST_WITHHOLD_PUSH).  The knobs of the closed experiments (LAB_NOTES: thresholds, ring depths, batch sizes, A/B
// switches of single kernels) exist in a LAB BUILD only -- `make LAB=1` compiles with -DLCG_HIP_LAB -- and read as unset
// otherwise, so that every branch they select folds away in the shipped .so.
inline const char *lab_env(const char *name)
{
#ifdef LCG_HIP_LAB
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
// Cache policy of the row-block product's y stores (placement study, DESIGN 9 / profiles/r04_placement*): 0 plain, 1 non-temporal,
// 2 write-through (sc1), 3 write-through at system scope (sc0 sc1).  LAB builds read LCG_HIP_Y_STORE.
inline int y_store_policy()
{
    const char *e = lab_env("LCG_HIP_Y_STORE");      // (read at every launch: a lab program switches it between launches)
    return e ? atoi(e) : 0;
}
// LCG_HIP_DEBUG=1: the plan builders, the placement of the product's output and the direct exchange say on stderr what they chose and
// why; 2: a solve also synchronises after every launch and names it (fault isolation)
inline int debug_level()
{
    static const int lv = [] { const char *e = std::getenv("LCG_HIP_DEBUG"); return e ? atoi(e) : 0; }();
    return lv;
}
inline bool debug_on() { return debug_level() != 0; }


constexpr int VB = 256;     // threads per block of every vector / scalar kernel
constexpr int MAXG = 2048;  // most blocks a reducing kernel launches = stride of the partial-sum table
constexpr int MAXR = 8;     // most simultaneous reductions of one kernel

enum { ST_RUNNING = 0, ST_CONVERGED = 1, ST_NAN = 2, ST_ALREADY = 3, ST_COMM = 4 };

// Direct all-reduce over peer-mapped mailboxes (comm.hip, "direct all-reduce").  Handed by value to
// the scalar kernel; all pointers are device addresses valid on THIS rank.
constexpr int XG_SLOT = 16;     // doubles per (parity, source) slot: MAXR values + the sequence word, 128 B
constexpr int XG_MAXP = 64;     // most ranks (one lane of one wavefront per peer)
struct XgBox {
    double *mine = nullptr;             // my mailbox [2][P][XG_SLOT], uncached/fine-grained device memory
    double *const *peers = nullptr;     // [P]: every rank's mailbox as mapped into this process (peers[me] == mine)
    unsigned long long *seq = nullptr;  // all-reduces issued so far (advanced by the kernel itself)
    int *fail = nullptr;                // raised when a peer's contribution did not arrive in time
    long long timeout_ticks = 0;        // wall_clock64 ticks (100 MHz)
    int P = 0, me = 0;
};

// Direct neighbour exchange (comm.hip, dist mode 2): the rank that OWNS a piece of x writes it into
// the neighbours' receive buffers over the peer mappings and then raises a flag word there; the
// neighbour's remote-column product waits for the flags of the call it belongs to.
constexpr int XG_MAXSEG = 8;        // most neighbours of a rank under this mode
constexpr int PUSH_CHUNK = 4096;    // doubles one pushing block moves
struct DevState;
struct WaitPlan {
    int n = 0;
    unsigned long long seq = 0;
    long long timeout_ticks = 0;
    int *fail = nullptr;
    const unsigned long long *flag[XG_MAXSEG];  // the neighbours' flag words in MY flag array
};
struct PushPlan {
    int nseg = 0, nflag = 0, nblocks = 0;
    unsigned long long seq = 0;         // number of this A.x call (same on every rank)
    unsigned int *ticket = nullptr;     // blocks finished so far (the last one raises the flags)
    const double *src[XG_MAXSEG];
    double *dst[XG_MAXSEG];             // peer memory
    long count[XG_MAXSEG];              // doubles
    int first_block[XG_MAXSEG + 1];
    unsigned long long *flag[XG_MAXSEG];        // my flag word in each neighbour's flag array
    // RECEIVING blocks behind the product's blocks (one-stream direct exchange): block j of them waits for the neighbours' flags of
    // this call and copies chunk j of what they wrote from the landing zone into the gather buffer -- k_recv's work without its
    // launch, overlapped with the product's tail (devcommon.hpp: recv_block)
    int nrecv = 0, rnseg = 0;
    const double *rsrc[XG_MAXSEG];
    double *rdst[XG_MAXSEG];
    long rcount[XG_MAXSEG];
    int rfirst[XG_MAXSEG + 1];
    WaitPlan wp;
    DevState *rst = nullptr;            // where a time-out is recorded (may be null)
};


// Partial sums per running sum: slots 0 .. g[r]-1 of table row r hold the r-th sum's partials (one per block of the reducing
// pass).  One running sum (two with yy) may instead come from the epilogue of an A.x kernel that leaves one partial per
// workgroup -- up to AXP_CAP of them -- in a buffer of its own: ax_n entries at axp feed sum ax_row, those at axp + AXP_CAP feed
// sum ax_row + 1.  Everything is added in a fixed order: the same bits wherever and however often the reduction runs.
constexpr int AXP_CAP = 16384;
struct PartCount {
    int g[MAXR];
    const double *axp = nullptr;
    int ax_n = 0, ax_row = -1, ax_yy = 0;
    void all(int v) { for (int r = 0; r < MAXR; r++) g[r] = v; ax_row = -1; ax_n = 0; ax_yy = 0; }
};

// Sums an A.x kernel leaves beside y (one GPU, solver loops): workgroup w adds its rows' y_i * u_i into part[w] and, when yy is
// set, y_i * y_i into part[AXP_CAP + w] -- the dot that follows every A.x of the Krylov loops without its own pass over y.
struct DotPlan {
    const double *u = nullptr;
    double *part = nullptr;
    int yy = 0;
    int stride = AXP_CAP;   // distance of the y.y sums from the y.u sums in `part`
    int ystore = 0;         // how k_spmv_ldsp stores y (devcommon.hpp: store_y; set by the host from y_store_policy())
    int ux = 0;             // u is the product's own x (CG, PCG: d.Ad): a run block that holds its diagonal takes u from its gathers
    int dof = 1;            // k_spmv_ldsp, long rows: a packed column field stands for `dof` consecutive columns (csr.hip: k_pk_dof)
};

struct DevState;

// Mirror of the stop state in host-mapped pinned memory; written by the scalar kernels,
// polled by the host without touching the stream.
struct HostStatus {
    volatile double residual;
    volatile int t;
    volatile int done;
    volatile int status;
    volatile int it;        // written last
};

// Everything the iteration carries between kernels.  Lives in device memory; kernels read
// their coefficients from here so no scalar ever has to visit the host inside the loop.
struct DevState {
    double red[MAXR];   // reduced sums of the latest reduction (all-reduce buffer when sharded)
    double s[32];       // solver scalars (indices: enum in each solver)
    double residual;    // value the next loop head tests (lcg.cpp:208-209)
    double eps;
    double n_global;
    int abs_diff;
    int it;             // iteration bodies started
    int t;              // completed iterations (the reference's t)
    int done;           // set once: every later kernel becomes a no-op
    int status;         // ST_*
    int pub_mask;       // HostStatus is refreshed when (it & pub_mask) == 0, and at every stop
    int zero_guess;     // the initial guess is all zeros (solvers_real.hip: ax_setup): its product is not made, A.m counts as zeros
    HostStatus *host;
};

struct Comm;            // comm.hip

struct Ctx {
    bool inited = false;
    int device = 0;
    hipStream_t stream = nullptr;      // stream in use
    hipStream_t own_stream = nullptr;  // created by the library
    hipStream_t comm_stream = nullptr; // second stream for gather/compute overlap
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    double *partials = nullptr;        // [MAXR][MAXG]: the table the latest reducing pass wrote (one of partials_pair)
    DevState *state = nullptr;         // the state the next kernel is handed (one of state_pair: driver.hpp, vecf)
    double *partials_pair[2] = {nullptr, nullptr};
    double *ax_partials = nullptr;     // [2][AXP_CAP]: the sums an A.x kernel carried (csr.hip: k_spmv_lds1d), see PartCount
    DevState *state_pair[2] = {nullptr, nullptr};
    HostStatus *hstat = nullptr;       // pinned, mapped
    HostStatus *hstat_dev = nullptr;   // device alias of hstat
    double *scratch_host = nullptr;    // pinned, 64 doubles
    DevState *snap[2] = {nullptr, nullptr};     // pinned, mapped: per-batch copies of DevState (sharded loop, driver.hpp)
    DevState *snap_dev[2] = {nullptr, nullptr}; // their device aliases (written by the batch's last scalar step)
    hipEvent_t snap_ev[2] = {nullptr, nullptr};
    Comm *comm = nullptr;
    int last_iters = 0;
    double last_residual = 0.0;
    bool in_solve = false;             // a Driver is alive: A.x may honour DevState::done
    const int *ax_skip = nullptr;      // the flag the NEXT built-in products honour instead of DevState::done (solvers_real.hip: ax_setup)
    int ax_rc = 0;                     // first failure of a built-in callback (void by liblcg's typedef) during this solve
    int cg_schedule = 0;               // LCG_HIP_CG_*
    bool profile = false;
    int profile_every = 1;             // time every k-th A.x only (each timed call costs ~2 x 2 us of markers)
    long ax_seq = 0;
    std::vector<hipEvent_t> prof_ev;   // pairs
    int prof_used = 0;
    double last_ax_mean_us = 0.0;
    int last_ax_calls = 0;
    int prof_pending = 0;              // events of the finished solve not yet turned into last_ax_mean_us (done on demand)
    DevState *state_stage = nullptr;   // pinned: the initial state of a solve on its way to the device (no stream sync)
    // scratch vectors of past solves, kept for the next one (hipMalloc + hipFree cost ~0.2 ms per solve, as much as ten
    // iterations of a small system; lcg_hip_trim() gives them back)
    struct Scratch { double *p; size_t bytes; bool busy; void *arena; };   // arena: the allocation this vector is a slot of (driver.hpp: Placement), or null
    std::vector<Scratch> scratch;
    // where the product's output lies (driver.hpp: Placement): what y = A.x took into a given vector against a given matrix
    // (val = the matrix's value array; y == nullptr: "this matrix has had its walk, do not look again")
    struct PlaceMemo { const void *val; const double *y; float us; };
    std::vector<PlaceMemo> place_memo;
    void forget_places() { place_memo.clear(); }
    // one vector left the pool: its timings go, the per-matrix "has had its walk" markers (y == nullptr) stay
    void forget_vector(const double *y)
    {
        if (!y) return;
        for (size_t i = 0; i < place_memo.size();) if (place_memo[i].y == y) place_memo.erase(place_memo.begin() + (long)i); else i++;
    }
    // Bounds of the placement's walk (driver.hpp: Placement::run).  Production values; lcg_hip_placement_tune_for_test lowers them so
    // that the walk, the arenas and their eviction run under -m gpu at a few million rows.
    struct PlaceTune {
        size_t stream_min = (size_t)768 << 20;     // a product that streams less shows one kind of place only: nothing is timed
        size_t chunk = (size_t)1 << 30;            // the walk's step
        int max_chunks = 128;
        double wall_ms = 60.0;                     // looked at after EVERY allocation of the walk
        size_t hold_max = (size_t)64 << 30;        // what a walk may hold at once: this many bytes ...
        double hold_frac = 0.25;                   // ... and this fraction of the memory that was free when it started
        size_t keep_free = (size_t)8 << 30;        // never into the last bytes of free memory
        size_t shared_min = (size_t)4 << 30;       // others held more than this when the library was initialised: the device is shared, no walk
        bool predict = true;                       // another chunk only while time used + the dearest allocation so far <= wall_ms
        size_t released_max = (size_t)1 << 30;     // the library has given back more than this in the process's life: no walk (below)
        int force_find_at = -1;                    // test hook: the k-th TIMED chunk is taken as the faster place whatever the clock says
    } place_tune;
    // What the library has handed back to the device in this process's life (matrices destroyed, pool vectors and arenas trimmed, the
    // chunks of an earlier walk).  A 1 GiB hipMalloc costs 0.1-0.5 ms out of memory the process has never held and 30 ms .. 0.5 s ONE
    // CALL once the allocator recycles what was released (bench.py's variants, LCG_HIP_DEBUG=1: walks of 74 / 381 / 528 ms against a
    // bound of 60 that can only be looked at between calls) -- so the walk is made while the allocator is fresh: in practice once per
    // process, for its first large system.
    size_t released_bytes = 0;
    bool ranks_share_device = false;               // two ranks of the communicator / of the mailboxes sit on this device (comm.hip: found at connect time)
    size_t mem_total = 0, mem_free_at_init = 0;    // hipMemGetInfo at ensure_init: total - free = what others (and the host program) held
    // the latest walk: chunks allocated, wall time, most bytes held at once, 1 = a faster place was kept, why it ended
    // what the latest solve enqueued: vector passes, scalar steps (a step that sums over ranks is one step, whatever it launches),
    // reductions over ranks (RCCL or mailboxes), products (lcg_hip_last_launches)
    int cnt_vec = 0, cnt_scal = 0, cnt_allreduce = 0, cnt_ax = 0;
    int walks_made = 0;
    int walk_chunks = 0, walk_found = 0; double walk_ms = 0.0; size_t walk_held = 0; const char *walk_end = "";
    int place_mode = -1;               // lcg_hip_set_placement: -1 auto (large products on one GPU), 0 never, 1 whenever the callback is the built-in one
    int place_timed = 0;               // candidates timed by the latest solve (0: answered from the memo, or not tried)
    int place_moved = 0;               // roles the latest solve moved to another vector
    double place_us_first = 0.0, place_us_chosen = 0.0;   // the latest solve's first output: as allocated / as placed
    unsigned shadow_seed = 1;
    std::vector<double> shadow_vec;    // explicit rbar0 for the next complex solve
    std::string err;
};

Ctx &ctx();
int ensure_init();
// the flag a built-in product falls through on (null outside a solve)
inline const int *ax_flag(Ctx &c) { return c.ax_skip ? c.ax_skip : (c.in_solve ? &c.state->done : nullptr); }
int fail(hipError_t e, const char *what, const char *file, int line);

#define HIPCHK(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return ::lcgh::fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

// ---- CSR handle ------------------------------------------------------------------------
struct CsrPart {
    int n_rows = 0;
    int64_t nnz = 0;
    int *rowptr = nullptr;
    int *col = nullptr;
    double *val = nullptr;
    bool owned = false;
    bool padded = false;           // >= 64 readable bytes follow col[nnz] and val[nnz]
    mutable int slice_R = 0;       // rows per block the next field was computed for (0 = not yet)
    mutable int max_slice = 0;     // largest block slice, entries (csr.hip: k_max_slice)
    // packed column indices for the one-window kernel (csr.hip: "packed columns"), built on first use
    mutable int pk_mode = -1;      // -1 auto (large real matrices), 0 never, 1 whenever eligible
    mutable int pk_state = 0;      // 0 not tried, 1 ready, -1 not eligible
    mutable int *pk_base = nullptr, *pk_ofs = nullptr;     // per block of 64 rows: smallest column, first group
    mutable void *pk_data = nullptr;                        // 16 bytes per group of 6 entries
    mutable int pk_maxrow = 0;                              // longest row (chooses the gather batch of the kernel)
    mutable int pk_bits = 21;                               // width of a packed column: 18 (seven per group) or 21 (six)
    mutable int pk_R = 64;                                  // rows per block of the packed form (64; 32 / 16 for long rows)
    mutable int pk_dof = 1;                                 // long rows: every row's entries come in groups of pk_dof consecutive columns (one packed field per group)
    mutable int pk_runs = 0;                                // blocks stored as runs (row 0's columns only; csr.hip: k_pk_meta)
    mutable int pk_tpls = 0;                                // blocks stored as templates (<= 32 diagonals + a mask per row)
    mutable long pk_groups = 0;                             // 16-byte groups of the packed columns
    mutable double *dot_part = nullptr;                     // [2][dot_cap]: per-block (per-chunk) sums of a product that carries its dot (k_spmv_ldsp<DOT>, k_tile_spmv2<DOT>)
    mutable long dot_cap = 0;
    // two-pass "binned" product for scattered columns (csr_binned.hip), plan built on first use
    int64_t n_cols = 0;            // columns the part addresses (0 = unknown: never binned)
    mutable int bn_mode = -1;      // -1 auto (large real matrices whose row blocks span more of x than the L2 holds), 0 never, 1 whenever eligible
    mutable int bn_state = 0;      // 0 not tried, 1 plan ready, -1 not eligible / not chosen
    mutable void *bn_plan = nullptr;
    mutable const char *bn_why = "not tried";   // why the plan is (not) there
    mutable double mean_span = -1.0;    // mean column span of a 64-row block (-1 = not measured)
    mutable double line_ratio = -1.0;   // distinct 128-byte lines of x per entry of a 64-row block (-1 = not measured; csr_choice.hip: k_line_ratio)
    mutable double diag_like = -1.0;    // fraction of entries whose column is one more than the entry above them (-1 = not measured)
    // one-pass "tiled" product for row-random bands (csr_tiled.hip), plan built on first use
    mutable int tl_mode = -1;      // -1 auto, 0 never, 1 whenever eligible
    mutable int tl_state = 0;      // 0 not tried, 1 plan ready, -1 not eligible / not chosen
    mutable void *tl_plan = nullptr;
    mutable const char *tl_why = "not tried";
    // row ranges (csr_choice.hip, "row ranges"): a matrix whose row blocks fall into different column-pattern classes is multiplied range by
    // range, each range -- a view of this part's arrays -- choosing its own kernel family
    int64_t end_abs = -1;          // a view only: offset one past its last entry in the shared col / val (-1: this part owns offsets 0 .. nnz)
    mutable int rg_mode = -1;      // -1 auto (>= 4M entries), 0 never, 1 whenever two classes are found
    mutable int rg_state = 0;      // 0 not tried, 1 split, -1 one range
    mutable void *rg_plan = nullptr;
    int lr_mode = 0;               // a view only: 1 = rows far longer than an LDS window live here (csr_choice.hip: long_rows_launch)
    mutable void *lr_plan = nullptr;
    mutable const char *last_kernel = "";   // name of the kernel family the latest product used
    mutable double plan_ms = 0.0;           // host time spent choosing a kernel family and building its copy of the matrix (first product)
};

} // namespace lcgh

struct lcg_hip_csr {
    int n_rows = 0;         // local rows
    int n_cols = 0;         // columns addressed by `main` (global when sharded)
    bool is_complex = false;
    lcgh::CsrPart main;     // the whole shard (global columns)
    lcgh::CsrPart op[4];    // [1] conj(A), [2] A^T, [3] A^H as their own CSR, built on first use (csr_build.hip: op_part)
    double *invdiag = nullptr;  // reciprocal diagonal (1 or 2 doubles per row)
    int variant = 0;        // SpMV kernel choice (0 auto)
    double mean_row = 0.0;
    // --- sharded operation (comm.hip) ---
    bool distributed = false;
    int dist_mode = 0;
    int64_t n_global = 0;
    int64_t row0 = 0;           // first global row of this shard
    int64_t rows_per_rank = 0;
    lcgh::CsrPart loc;          // entries whose column is owned by this rank (LOCAL column index)
    lcgh::CsrPart rem;          // the others (column index into xfull / halo buffer)
    lcgh::CsrPart remc;         // rem without its empty rows: own rowptr, rem's col/val (comm.hip: dist_split)
    int *rem_rows = nullptr;    // local row of each remc row
    double *rem_y = nullptr;    // remc . xfull, scattered into y after the local product
    double *xfull = nullptr;    // gather buffer
    double *op_z = nullptr;     // (A_r)^T . x_r over the padded global height, and one rows-per-rank block behind it (dist_spmv_op)
    void *halo = nullptr;       // neighbour-exchange plan (comm.hip)
    void *direct = nullptr;     // direct (peer-mapped) exchange state (comm.hip, mode 2)
};

namespace lcgh {

// csr.hip
int spmv_launch(const CsrPart &P, bool is_complex, int variant, double mean_row, const double *x,
                double *y, bool accumulate, hipStream_t s, const int *done_flag);
// the same product with `pp.nblocks` pushing blocks in front of the grid (comm.hip, dist mode 2)
int spmv_launch_push(const CsrPart &P, bool is_complex, int variant, double mean_row, const double *x, double *y,
                     hipStream_t s, const int *done_flag, const PushPlan &pp);
int jacobi_launch(const lcg_hip_csr *A, const double *x, double *z, int n, hipStream_t s);       // csr_build.hip
// y = A.x with the sums y.u (and y.y) riding in the product: 1 = done, *slots partial sums wait in part[0 .. *slots) (and
// part[AXP_CAP ..)); 0 = this matrix / kernel family cannot (nothing was launched: the caller multiplies and reduces as before); < 0 failure
int csr_ax_dot(lcg_hip_csr *A, const double *x, double *y, const double *u, int yy, double *part, int *slots, hipStream_t s,
               const int *done_flag);
// the packed kernel of one part carrying y.u (and y.y), optionally with a shard's pushing blocks in front: 1 launched, 0 not this part
int csr_part_ax_dot(const CsrPart &P, int variant, double mean_row, const double *x, double *y, const double *u, int yy, double *part,
                    int *slots, hipStream_t s, const int *done_flag, const PushPlan *pp, int *nofold);
// csr_binned.hip
int binned_ready(const CsrPart &P, hipStream_t s);      // 1 plan ready, 0 not eligible, < 0 failure
int binned_launch(const CsrPart &P, const double *x, double *y, hipStream_t s, const int *done_flag);
void binned_free(CsrPart &P);
long binned_traffic_bytes(const CsrPart &P);
size_t binned_plan_bytes(const CsrPart &P);
// csr_tiled.hip
int tiled_ready(const CsrPart &P, hipStream_t s, double min_fill);      // 1 plan ready, 0 not eligible, < 0 failure
int tiled_launch(const CsrPart &P, const double *x, double *y, hipStream_t s, const int *done_flag, const PushPlan *push = nullptr,
                 const DotPlan *dot = nullptr);
int tiled_chunks(const CsrPart &P);         // chunks of 1024 rows = per-chunk sums a dot-carrying tiled product leaves
bool tiled_dot_ok(const CsrPart &P);        // the plan's shape has a dot-carrying instance
void tiled_free(CsrPart &P);
long tiled_traffic_bytes(const CsrPart &P);
long tiled_tile_copy_bytes(const CsrPart &P);
size_t tiled_plan_bytes(const CsrPart &P);
// csr_build.hip
int op_part(lcg_hip_csr *A, int layout, int conjugate, const CsrPart **out);
int alloc_part(CsrPart &P, int n_rows, long nnz, bool cplx);        // owned, padded arrays of a part
int device_exclusive_scan(int n, const int *counts, int *rowptr, hipStream_t s, long *total);
// csr.hip
void free_part(CsrPart &P);                                          // a part's arrays and every plan built beside them

// comm.hip
int comm_allreduce(double *dev, int count, hipStream_t s);
bool comm_active();
bool xg_box(XgBox *out);        // true when the direct all-reduce is connected and enabled
int dist_spmv(lcg_hip_csr *A, const double *x, double *y);
// the sharded product, carrying y.u where it can: the local product's partial sums (folded) followed by the remote-column kernel's, in
// part[0 .. *slots).  The product is ALWAYS made: 1 = with the sum, 2 = without it (the caller reduces in its own pass), < 0 failure
int dist_ax_dot(lcg_hip_csr *A, const double *x, double *y, const double *u, int yy, double *part, int *slots);
int dist_spmv_op(lcg_hip_csr *A, const CsrPart &T, const double *x, double *y);     // y = this rank's rows of A^T.x / A^H.x (T = (A_r)^T, csr_build.hip: op_part)

} // namespace lcgh
