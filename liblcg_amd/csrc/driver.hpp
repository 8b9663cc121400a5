// driver.hpp -- host side of a device-resident Krylov loop.
//
// The host only enqueues: every coefficient lives in DevState, the stop test runs in the
// scalar kernels, and once it fires the remaining enqueued kernels fall through.  Without a
// progress callback the host therefore never blocks on the stream; it watches the
// host-mapped HostStatus to (a) stop enqueuing soon after convergence and (b) stay at most
// `inflight` iterations ahead.  With a progress callback the reference's contract (residual
// and live m handed over before every iteration, lcg.cpp:211-217) forces one stream
// synchronisation per iteration.
#pragma once

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <cstring>
#include <thread>
#include <typeinfo>
#include <vector>

#include "devcommon.hpp"

namespace lcgh {

struct Driver {
    Ctx &c;
    long n;            // local vector length (reals: elements, complex: complex elements)
    bool cplx;
    int max_it, abs_diff;
    double eps;
    int enq = 0;       // iteration bodies enqueued
    // a callback of the caller's own (not the built-in product / Jacobi) is in the loop: such a callback cannot honour the stop
    // flag, so nothing is enqueued ahead of a verdict the reference would have returned on first (lcg.cpp:186-203)
    bool user_cb = false;

    Driver(Ctx &c_, long n_, bool cplx_, int max_it_, double eps_, int abs_diff_)
        : c(c_), n(n_), cplx(cplx_), max_it(max_it_), abs_diff(abs_diff_), eps(eps_)
    { c.in_solve = true; c.ax_rc = 0; c.cnt_vec = c.cnt_scal = c.cnt_allreduce = c.cnt_ax = 0; }
    ~Driver() { c.in_solve = false; }
    Driver(const Driver &) = delete;

    // ---- launches ------------------------------------------------------------------------
    template <class Op> int vec(Op op, uintptr_t align_or = 0) { return vec_n(op, n, align_or); }
    // a reducing pass whose sums land in table rows row0, row0 + 1, ... beside the rows an earlier pass of the same body left
    // there (same grid: the per-row counts stay as they are)
    template <class Op> int vec_rows(Op op, int row0, uintptr_t align_or)
    {
        const PartCount keep = pcnt;
        double *table = c.partials;
        c.partials = table + (size_t)row0 * MAXG;
        int rc = vec_n(op, n, align_or);
        c.partials = table;
        const int g = pcnt.g[0];
        pcnt = keep;
        for (int r = 0; r < Op::NR; r++) pcnt.g[row0 + r] = g;
        return rc;
    }
    // same pass over a vector of another length (e.g. the per-block partials of a fused A.x)
    template <class Op> int vec_n(Op op, long n, uintptr_t align_or)
    {
        // complex vectors are naturally 16-byte elements; reals use the 2-wide path when
        // every pointer is 16-byte aligned
        const bool v2 = cplx || ((align_or & 15) == 0);
        const long items = cplx ? n : (v2 ? (n + 1) / 2 : n);
        const int g = grid_for(items);
        c.cnt_vec++;
        op.st = c.state;        // the state of THIS launch (vecf below moves it between the buffers of a pair)
        if (cplx) {
            hipLaunchKernelGGL((k_vec<Op, false>), dim3(g), dim3(VB), 0, c.stream, op, n, c.partials);
        } else if (v2) {
            hipLaunchKernelGGL((k_vec<Op, true>), dim3(g), dim3(VB), 0, c.stream, op, n, c.partials);
        } else {
            hipLaunchKernelGGL((k_vec<Op, false>), dim3(g), dim3(VB), 0, c.stream, op, n, c.partials);
        }
        if (Op::NR > 0) pcnt.all(g);
        HIPCHK(hipGetLastError());
        return dbg(typeid(Op).name());
    }
    // Scalar step `fin` (on the sums of the latest reducing pass) fused into the pass `op` that consumes its result:
    // one launch instead of k_scal + k_vec (devcommon.hpp: k_vecf).  One GPU; when the rows are sharded the
    // sums have to meet the other ranks' first, so the step stays its own kernel(s) around the all-reduce.
    template <class Fin, class Op> int vecf(Fin fin, Op op, uintptr_t align_or = 0)
    {
        static const bool off = lab_env("LCG_HIP_NO_FUSED_SCALAR") != nullptr;      // A/B runs
        if (comm_active() || off) {
            int rc = scal(fin);
            return rc ? rc : vec(op, align_or);
        }
        c.cnt_vec++;        // (the scalar step rides in this pass: no launch of its own)
        const bool v2 = !cplx && (align_or & 15) == 0;      // complex passes take one 16-byte element per lane (vec_n)
        const int g = grid_for(v2 ? (n + 1) / 2 : n);
        DevState *cur = c.state, *next = c.state == c.state_pair[0] ? c.state_pair[1] : c.state_pair[0];
        double *pin = c.partials, *pout = c.partials == c.partials_pair[0] ? c.partials_pair[1] : c.partials_pair[0];
        if (v2) hipLaunchKernelGGL((k_vecf<Fin, Op, true>), dim3(g), dim3(VB), 0, c.stream, fin, op, n, pin, pcnt, pout, cur, next);
        else hipLaunchKernelGGL((k_vecf<Fin, Op, false>), dim3(g), dim3(VB), 0, c.stream, fin, op, n, pin, pcnt, pout, cur, next);
        HIPCHK(hipGetLastError());
        c.state = next;
        if (Op::NR > 0) { c.partials = pout; pcnt.all(g); }
        return dbg(typeid(Op).name());
    }
    // partial sums waiting in c.partials, per running sum: the latest reducing pass sets all rows to its grid; an A.x that
    // carried a dot in its epilogue (csr_ax_dot) sets the rows it wrote
    PartCount pcnt = [] { PartCount p; p.all(1); return p; }();
    // A body whose closing scalar step rides in the NEXT body's first pass (vecf) leaves the last one open: `tail`
    // closes it, once, after the last body and before the state is read.
    std::function<int()> tail;
    int run_tail() { if (!tail) return 0; auto t = std::move(tail); tail = nullptr; return t(); }
    // LCG_HIP_DEBUG=2: synchronise after every launch and name it on stderr (fault isolation)
    bool debug_sync = debug_level() >= 2;
    int dbg(const char *what)
    {
        if (!debug_sync) return 0;
        std::fprintf(stderr, "[lcg_hip] %s ...", what); std::fflush(stderr);
        HIPCHK(hipStreamSynchronize(c.stream));
        std::fprintf(stderr, " ok\n");
        return 0;
    }

    // scalar step after the most recent reducing vec()
    // The sharded loop's host decides from the state as of the LAST body of a batch (run_lockstep).  With sharded rows only the
    // scalar steps change DevState, so every step of that body leaves the state it produced in `snap_to` (host-mapped; the body's
    // last step wins, in stream order): no device-to-host copy and none of its two boundaries on the stream.
    DevState *snap_to = nullptr;
    int snap_taken = 0;
    template <class Fin> int scal(Fin fin)
    {
        const PartCount g = pcnt;
        XgBox xb;
        c.cnt_scal++;
        if (comm_active() && Fin::NR > 0) c.cnt_allreduce++;       // (RCCL all-reduce between two launches, or the mailboxes inside one)
        if (snap_to) snap_taken++;
        if (comm_active() && Fin::NR > 0 && xg_box(&xb)) {
            // reduce + exchange over the peer mailboxes + scalar step in one launch
            hipLaunchKernelGGL((k_scal<Fin>), dim3(1), dim3(VB), 0, c.stream, fin, c.partials, g, c.state, SC_XGMI, xb, snap_to);
        } else if (comm_active() && Fin::NR > 0) {
            hipLaunchKernelGGL((k_scal<Fin>), dim3(1), dim3(VB), 0, c.stream, fin, c.partials, g, c.state, SC_REDUCE, xb, (DevState *)nullptr);
            int rc = comm_allreduce(c.state->red, Fin::NR, c.stream);
            if (rc) return rc;
            hipLaunchKernelGGL((k_scal<Fin>), dim3(1), dim3(VB), 0, c.stream, fin, c.partials, g, c.state, SC_FIN, xb, snap_to);
        } else {
            hipLaunchKernelGGL((k_scal<Fin>), dim3(1), dim3(VB), 0, c.stream, fin, c.partials, g, c.state, SC_FUSED, xb, snap_to);
        }
        HIPCHK(hipGetLastError());
        return dbg(typeid(Fin).name());
    }

    // ---- state -------------------------------------------------------------------------------
    int init_state(double n_global)
    {
        DevState h;
        std::memset(&h, 0, sizeof h);
        h.eps = eps; h.abs_diff = abs_diff; h.n_global = n_global; h.host = c.hstat_dev;
        // the lock-step loop of a sharded run never looks at the mirror; the asynchronous loop paces
        // itself on it: every 4th body on small systems (3-5 us of a 20-35 us iteration at 1e4 rows),
        // every body where a body is long and only 6 are kept in flight
        h.pub_mask = comm_active() ? 0x3fffffff : ((cplx ? 2 * n : n) >= (1 << 20) ? 0 : 3);
        if (const char *e = lab_env("LCG_HIP_PUBLISH_EVERY")) h.pub_mask = std::max(1, atoi(e)) - 1;
        c.hstat->it = 0; c.hstat->done = 0; c.hstat->status = 0; c.hstat->t = 0; c.hstat->residual = 0.0;
        (void)lcg_hip_last_ax_mean_us();        // a previous solve's events, if nobody asked yet: they are about to be reused
        // from a pinned staging slot, in stream order in front of the solve's first kernel: no synchronisation here (the
        // previous solve ended with one, so the slot is free)
        *c.state_stage = h;
        HIPCHK(hipMemcpyAsync(c.state, c.state_stage, sizeof h, hipMemcpyHostToDevice, c.stream));
        // kernels that a sharded A.x puts on the second stream read the stop flag too, and the direct exchange starts them
        // without a fork event: they must not see the previous solve's flag
        HIPCHK(hipEventRecord(c.ev_a, c.stream));
        HIPCHK(hipStreamWaitEvent(c.comm_stream, c.ev_a, 0));
        return 0;
    }
    int read_state(DevState &h)
    {
        HIPCHK(hipMemcpyAsync(&h, c.state, sizeof h, hipMemcpyDeviceToHost, c.stream));
        HIPCHK(hipStreamSynchronize(c.stream));
        return 0;
    }

    // A.x callback with optional event timing.  liblcg's callback types return void (lcg.h:37-38), so the built-in
    // callbacks park their failure (a refused exchange, an RCCL error, a HIP error) in Ctx::ax_rc; it ends the
    // solve here with that LCG_HIP_E_* code instead of letting the loop run on over a stale product.
    template <class F> int timed_ax(F &&call)
    {
        c.cnt_ax++;
        if (c.profile && (c.ax_seq++ % c.profile_every) == 0 && c.prof_used + 2 <= (int)c.prof_ev.size()) {
            HIPCHK(hipEventRecord(c.prof_ev[c.prof_used], c.stream));
            call();
            HIPCHK(hipEventRecord(c.prof_ev[c.prof_used + 1], c.stream));
            c.prof_used += 2;
        } else {
            call();
        }
        return c.ax_rc;
    }
    // same for the preconditioner callback
    template <class F> int checked_mx(F &&call) { call(); return c.ax_rc; }

    // ---- the loop --------------------------------------------------------------------------
    // body(): enqueue one counted iteration.  pfp(residual, t): progress callback or nullptr
    // semantics via `has_pfp`.  Returns the liblcg status code.
    template <class Body, class Pfp>
    int run(Body &&body, bool has_pfp, Pfp &&pfp, int code_max_it, int code_nan)
    {
        DevState h;
        int rc = 0;
        if (has_pfp || user_cb) {
            rc = read_state(h);
            if (rc) return rc;
            if (h.status == ST_ALREADY) {                   // lcg.cpp:186-203
                if (has_pfp) pfp(h.residual, 0);
                finish(h);
                return LCG_ALREADY_OPTIMIZIED;
            }
        }
        // (without a progress callback and with the built-in callbacks nobody needs the setup's verdict yet: "already optimised" has
        //  set the stop flag, the bodies enqueued below fall through -- the built-in products honour the flag --, and the verdict is
        //  read with the final state: one stream drain less per solve.  A user's A.x / M callback would be CALLED for every body
        //  enqueued ahead, which the reference never does after this verdict: there the verdict is read first.)
        if (has_pfp) {
            for (;;) {                                      // lcg.cpp:206-230, one sync per iteration
                if (pfp(h.residual, h.t)) { finish(h); return LCG_STOP; }
                if (h.residual <= eps) { finish(h); return LCG_CONVERGENCE; }
                if (max_it > 0 && h.t + 1 > max_it) { finish(h); return code_max_it; }
                rc = body(); if (rc) return rc;
                enq++;
                rc = read_state(h); if (rc) return rc;
                if (h.status == ST_COMM) return comm_lost(h);
                if (h.status == ST_NAN) { finish(h); return code_nan; }
            }
        }
        if (comm_active()) return run_lockstep(body, code_max_it, code_nan);
        // asynchronous path
        const long work = cplx ? 2 * n : n;
        int inflight = work >= (1 << 20) ? 6 : 24;
        if (const char *e = lab_env("LCG_HIP_INFLIGHT")) inflight = std::max(1, atoi(e));
        // (Replaying the body from a hipGraph was measured and dropped: on the launch-bound 1e4-row
        // system an iteration is six DEPENDENT ~2 us kernels, 23 us eager vs 24.6 us replayed in
        // batches of four -- the chain on the device is the limit, not the host's launch rate.)
        for (;;) {
            if (max_it > 0 && enq >= max_it) break;
            rc = body(); if (rc) return rc;
            enq++;
            if (c.hstat->done) break;
            // stay at most `inflight` bodies ahead of the device
            int spins = 0;
            while (c.hstat->it < enq - inflight && !c.hstat->done) {
                if (++spins > 2000) { std::this_thread::sleep_for(std::chrono::microseconds(20)); }
                if (spins > 200000) {   // backstop: the mapped mirror is not advancing
                    rc = read_state(h); if (rc) return rc;
                    if (h.done || h.it >= enq - inflight) break;
                    spins = 0;
                }
            }
            if ((enq & 255) == 0) {     // authoritative check now and then
                rc = read_state(h); if (rc) return rc;
                if (h.done) break;
            }
        }
        rc = run_tail(); if (rc) return rc;
        rc = read_state(h); if (rc) return rc;
        if (h.status == ST_COMM) return comm_lost(h);
        finish(h);
        if (h.status == ST_ALREADY) return LCG_ALREADY_OPTIMIZIED;
        if (h.status == ST_NAN) return code_nan;
        if (h.done && h.status == ST_CONVERGED) return LCG_CONVERGENCE;
        return code_max_it;
    }

    // Sharded rows: every body holds collectives, so every rank must enqueue the SAME number of
    // bodies -- a rank that stopped on its own view of the mapped stop flag would leave its peers
    // waiting in an all-reduce.  Bodies therefore go out in batches; behind each batch the stream
    // copies DevState into a pinned slot, and the decision to go on after batch k+1 is taken from
    // the slot of batch k.  That slot holds the state as of one fixed body, computed from
    // all-reduced sums only, so it is bit-identical on all ranks and so is the decision; and the
    // stream always holds one whole batch while the host waits (no drain).
    template <class Body>
    int run_lockstep(Body &&body, int code_max_it, int code_nan)
    {
        int batch = 8;
        if (const char *e = lab_env("LCG_HIP_BATCH")) batch = std::max(1, atoi(e));
        int nb = 0, rc = 0;
        DevState h;
        for (;;) {
            int todo = batch;
            if (max_it > 0) todo = std::min(batch, max_it - enq);
            if (todo <= 0) break;
            const int slot = nb & 1;
            for (int i = 0; i < todo; i++) {
                if (i == todo - 1) { snap_to = c.snap_dev[slot]; snap_taken = 0; }
                rc = body();
                if (rc) { snap_to = nullptr; return rc; }
                enq++;
            }
            snap_to = nullptr;
            // (a body without a scalar step of its own -- none of the built-in loops -- is copied the old way)
            if (!snap_taken) HIPCHK(hipMemcpyAsync(c.snap[slot], c.state, sizeof(DevState), hipMemcpyDeviceToHost, c.stream));
            HIPCHK(hipEventRecord(c.snap_ev[slot], c.stream));
            nb++;
            if (nb >= 2) {
                const int prev = (nb - 2) & 1;
                HIPCHK(hipEventSynchronize(c.snap_ev[prev]));
                if (c.snap[prev]->done) break;
            }
        }
        // the verdict: the snapshot behind the last batch (bodies enqueued after a stop fall through: counts, residual and status
        // are those of the stop) -- the stream is drained by waiting for that batch's event, nothing is copied
        if (nb > 0) {
            HIPCHK(hipEventSynchronize(c.snap_ev[(nb - 1) & 1]));
            HIPCHK(hipStreamSynchronize(c.stream));
            std::memcpy(&h, c.snap[(nb - 1) & 1], sizeof h);
        } else {
            rc = read_state(h); if (rc) return rc;
        }
        if (h.status == ST_COMM) return comm_lost(h);
        finish(h);
        if (h.status == ST_ALREADY) return LCG_ALREADY_OPTIMIZIED;
        if (h.status == ST_NAN) return code_nan;
        if (h.done && h.status == ST_CONVERGED) return LCG_CONVERGENCE;
        return code_max_it;
    }

    // a peer's sums did not arrive in the direct all-reduce (devcommon.hpp: xg_allreduce)
    int comm_lost(const DevState &h)
    {
        finish(h);
        c.err = "direct all-reduce: a peer's contribution did not arrive in time";
        return LCG_HIP_E_COMM;
    }

    void finish(const DevState &h)
    {
        c.last_iters = h.t;
        c.last_residual = h.residual;
        c.last_ax_calls = c.prof_used / 2;
        c.last_ax_mean_us = 0.0;
        c.prof_pending = c.profile ? c.prof_used : 0;   // turned into a mean when somebody asks (lcg_hip_last_ax_mean_us)
        c.prof_used = 0;
    }
};

// copy-in / copy-out of caller vectors when mem == HOST
struct HostBridge {
    double *dm = nullptr, *dB = nullptr;
    double *hm = nullptr;
    size_t bytes = 0;
    bool active = false;
    int open(int mem, double *&m, const double *&B, size_t nbytes, hipStream_t s)
    {
        if (mem == LCG_HIP_MEM_DEVICE) return 0;
        active = true; bytes = nbytes; hm = m;
        HIPCHK(hipMalloc(&dm, nbytes));
        HIPCHK(hipMalloc(&dB, nbytes));
        HIPCHK(hipMemcpyAsync(dm, m, nbytes, hipMemcpyHostToDevice, s));
        HIPCHK(hipMemcpyAsync(dB, B, nbytes, hipMemcpyHostToDevice, s));
        m = dm; B = dB;
        return 0;
    }
    int close(hipStream_t s)
    {
        if (!active) return 0;
        hipError_t e = hipMemcpyAsync(hm, dm, bytes, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        hipFree(dm); hipFree(dB);
        active = false;
        if (e != hipSuccess) return fail(e, "copy back m", __FILE__, __LINE__);
        return 0;
    }
    ~HostBridge() { if (active) { hipFree(dm); hipFree(dB); } }
};

// device scratch vectors of one solve (freed on scope exit); caller-provided pointers win
struct Workspace {
    int get(double *&out, double *given, size_t bytes)
    {
        if (given) { out = given; return 0; }
        Ctx &c = ctx();
        int best = -1;
        for (size_t i = 0; i < c.scratch.size(); i++)
            if (!c.scratch[i].busy && c.scratch[i].bytes >= bytes && (best < 0 || c.scratch[i].bytes < c.scratch[best].bytes)) best = (int)i;
        if (best < 0) {
            // nothing fits: give one idle smaller vector back and allocate
            for (size_t i = 0; i < c.scratch.size(); i++)
                if (!c.scratch[i].busy && !c.scratch[i].arena) { c.forget_vector(c.scratch[i].p); (void)hipFree(c.scratch[i].p); c.scratch.erase(c.scratch.begin() + i); break; }
            double *p = nullptr;
            HIPCHK(hipMalloc(&p, bytes));
            c.scratch.push_back({p, bytes, false, nullptr});
            best = (int)c.scratch.size() - 1;
            // indices held by this workspace may have shifted by the erase above: they are re-resolved by pointer below
        }
        c.scratch[best].busy = true;
        held.push_back(c.scratch[best].p);
        out = c.scratch[best].p;
        return 0;
    }
    std::vector<double *> held;
    bool owns(const double *p) const { for (double *h : held) if (h == p) return true; return false; }
    // an idle vector of the pool joins this solve (Placement: it becomes a product's output)
    void adopt(double *p)
    {
        for (auto &s : ctx().scratch) if (s.p == p) s.busy = true;
        held.push_back(p);
    }
    ~Workspace()
    {
        Ctx &c = ctx();
        for (double *p : held)
            for (auto &s : c.scratch)
                if (s.p == p) s.busy = false;
    }
};

// ---- where the work vectors lie (lcg_hip.h: lcg_hip_set_placement; profiles/r04_placement.txt) -------------------
// The time of a large y = A.x depends on WHICH allocation y is: 520-530 us or 580-595 us for the headline system, constant for
// the lifetime of the pair (value array, y).  The device's memory falls into THREE groups of 96 GiB (the three ranks of its 12-high
// HBM stacks, by every sign: one group a single stretch, the other two interleaved in runs of 2-8 GiB -- scripts/placement_lab7.hip
// over 250 allocations of 1 GiB), and a vector that lies in the group of the matrix's value array is slow to write while that array
// streams (scripts/placement_lab6.hip: six read buffers x sixteen written ones fall into matching groups).  Nothing the process can
// see -- virtual address, size, allocation call -- tells the group; only the clock does.  Inside the CG loop
// (scripts/placement_roles.py, every combination of classes for g, d, A.d): A.d beside the values costs 9 % of the iteration rate
// whatever the others do, d (the product's x, and the direction pass's output) 2.6 %, g 1.3 %.  The solvers allocate their work
// vectors anyway: before the first iteration each is classed by the product's time into it, and the roles are dealt in that order
// of weight -- products' outputs first, then the vector the product reads, then the rest -- the vectors outside the value array's
// group going to the heaviest roles.  No arithmetic changes.
struct Placement {
    static bool wanted(Ctx &c, int n, const void *afp, const void *inst)
    {
        if (c.place_mode == 0 || afp != (const void *)lcg_hip_csr_ax || inst == nullptr) return false;
        const lcg_hip_csr *A = static_cast<const lcg_hip_csr *>(inst);
        if (A->is_complex || n != A->n_rows || n < 4096) return false;
        return c.place_mode > 0 || streams(c, A);
    }
    // the rows this process multiplies by itself: the whole matrix, or -- sharded -- the entries with locally owned columns
    static const CsrPart &part(const lcg_hip_csr *A) { return A->distributed ? A->loc : A->main; }
    // the effect needs a stream far larger than the 256 MB Infinity Cache: at the 8-way shard size of the 10M-row system (490 MB streamed,
    // 10 MB written) eighteen candidates and a whole walk showed ONE kind of place, 69.9-71.8 us (profiles/r04_placement.txt, section 10)
    static bool streams(Ctx &c, const lcg_hip_csr *A) { return (size_t)part(A).nnz * 12 >= c.place_tune.stream_min; }
    // the mean row length the solver's loop hands the product (a shard's local-column part has its own: comm.hip) -- the kernel
    // shape that is timed here must be the one the loop runs
    static double mean_of(const lcg_hip_csr *A) { return A->distributed ? (A->n_rows ? (double)A->loc.nnz / A->n_rows : 0.0) : A->mean_row; }
    static float *memo(Ctx &c, const void *val, const double *y)
    {
        for (auto &m : c.place_memo) if (m.val == val && m.y == y) return &m.us;
        return nullptr;
    }
    static void forget_y(Ctx &c, const double *y)
    {
        for (size_t i = 0; i < c.place_memo.size();) if (c.place_memo[i].y == y) c.place_memo.erase(c.place_memo.begin() + i); else i++;
    }
    // what y = A.x takes into `y`: one product to warm up, two timed (x: the right-hand side)
    static int time_output(Ctx &c, const lcg_hip_csr *A, const double *x, double *y, float *us)
    {
        const CsrPart &P = part(A);
        if (float *m = memo(c, P.val, y)) { *us = *m; return 0; }
        hipEvent_t e0, e1;
        if (hipEventCreate(&e0) != hipSuccess) { (void)hipGetLastError(); return LCG_HIP_E_RUNTIME; }
        if (hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(e0); return LCG_HIP_E_RUNTIME; }
        int rc = spmv_launch(P, false, A->variant, mean_of(A), x, y, false, c.stream, nullptr);
        if (!rc) rc = hipEventRecord(e0, c.stream) == hipSuccess ? 0 : LCG_HIP_E_RUNTIME;
        for (int i = 0; i < 2 && !rc; i++) rc = spmv_launch(P, false, A->variant, mean_of(A), x, y, false, c.stream, nullptr);
        if (!rc && (hipEventRecord(e1, c.stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = LCG_HIP_E_RUNTIME;
        float ms = 0.f;
        if (!rc && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = LCG_HIP_E_RUNTIME;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (rc) return rc;
        *us = ms * 500.f;
        c.place_memo.push_back({P.val, y, *us});
        c.place_timed++;
        return 0;
    }
    // roles: the solve's work vectors by weight -- the products' outputs (n_out of them) first, then the vector the most frequent
    // product reads, then the rest.  Roles whose vector the caller supplied stay where they are.  x: any n-vector that holds finite
    // numbers (the right-hand side).
    // Placement is an optimisation: a timing that fails (an event that cannot be made, a trial product that errors) means
    // "not tried" -- the roles stay as allocated (they are dealt only after the last timing) and the solve goes on.
    static int run(Ctx &c, int n, const void *afp, void *inst, const double *x, Workspace &ws, std::initializer_list<double **> roles, int n_out)
    {
        const int rc = run_impl(c, n, afp, inst, x, ws, roles, n_out);
        if (rc) {
            if (debug_on()) fprintf(stderr, "[lcg_hip] placement: a timing failed (rc %d: %s): not tried, the vectors stay as allocated\n", rc, c.err.c_str());
            (void)hipGetLastError();
            c.place_timed = c.place_moved = 0;
        }
        return 0;
    }
    static int run_impl(Ctx &c, int n, const void *afp, void *inst, const double *x, Workspace &ws, std::initializer_list<double **> roles, int n_out)
    {
        c.place_timed = c.place_moved = 0; c.place_us_first = c.place_us_chosen = 0.0;
        if (!wanted(c, n, afp, inst)) return 0;
        const lcg_hip_csr *A = static_cast<const lcg_hip_csr *>(inst);
        const size_t bytes = sizeof(double) * (size_t)n;
        const void *val = part(A).val;
        constexpr float SAME = 1.03f;       // a role changes hands for 3 %
        constexpr float CLASS = 1.06f;      // two KINDS of places lie 8-12 % apart (timings of one kind spread by up to 4 %): what starts a walk and ends it
        struct Cand { double **role; double *p; float us; };
        std::vector<Cand> cand;
        int k = 0, out_owned = 0;
        for (double **r : roles) { if (ws.owns(*r)) { cand.push_back({r, *r, 0.f}); if (k < n_out) out_owned++; } k++; }
        if (out_owned == 0) return 0;
        const size_t n_own = cand.size();
        // idle vectors of the pool: the slots of arenas first (each was the fast place of SOME matrix), six at most -- a process that has
        // multiplied many matrices must not time its whole history against the next one
        for (int pass = 0; pass < 2; pass++)
            for (auto &s : c.scratch)
                if (!s.busy && s.bytes >= bytes && (s.arena != nullptr) == (pass == 0) && cand.size() < n_own + 6) cand.push_back({nullptr, s.p, 0.f});
        // the plan of a matrix is built by its first product: not on the clock
        if (part(A).last_kernel[0] == 0) { int rc = spmv_launch(part(A), false, A->variant, mean_of(A), x, cand[0].p, false, c.stream, nullptr); if (rc) return rc; }
        {   // What a better place can give is what the written vector costs in the worse one: 60 us per 80 MB (0.75 us per MiB).  Where that is
            // less than 4 % of the product -- the two-pass binned product at 1.75 ms, bands a million columns wide -- the kinds cannot be
            // told apart by the clock and nothing is tried (39 vectors timed and 64 chunks walked for nothing, 0.3 s, before this line).
            int rc = time_output(c, A, x, cand[0].p, &cand[0].us); if (rc) return rc;
            c.place_us_first = c.place_us_chosen = cand[0].us;
            if (c.place_mode < 0 && 0.75 * ((double)bytes / 1048576.0) < 0.04 * cand[0].us) {      // (automatic mode: a forced placement always tries)
                if (debug_on()) fprintf(stderr, "[lcg_hip] placement: not tried (a product of %.0f us writes %.0f MiB: nothing a place could give shows on the clock)\n", cand[0].us, bytes / 1048576.0);
                return 0;
            }
        }
        for (auto &q : cand) { int rc = time_output(c, A, x, q.p, &q.us); if (rc) return rc; }
        float lo = cand[0].us, hi = cand[0].us;
        for (auto &q : cand) { lo = std::min(lo, q.us); hi = std::max(hi, q.us); }
        // Every role is dealt, the heaviest first.  (Outputs only -- the vector the product reads pays less beside the values than the output
        // does, and the clock of one product cannot tell the two other groups apart -- was tried after one process in fifteen came out 2.5 %
        // slower placed than as allocated: on a box where nothing allocated is fast it gave 1488-1504 it/s where all roles out of the
        // found arena give 1524-1553; and trial SOLVES of six iterations per assignment, to let the loop itself choose, differ by less
        // than their own order effect.  Both are in profiles/r04_placement.txt; neither is in the code.)
        const size_t n_deal = n_own;
        size_t n_slow = 0;
        for (size_t i = 0; i < n_deal; i++) if (cand[i].us > lo * CLASS) n_slow++;
        // Not enough vectors outside the value array's group (or all alike: then nobody knows which kind they are).  What is allocated
        // one after the other lies side by side, so the library walks: chunks of 1 GiB, one after the other and all held, the product
        // timed into the start of every fourth, until one is clearly faster than our slow kind (or, all alike, clearly slower: then
        // ours are the fast kind) -- within hard bounds (Ctx::PlaceTune: 128 chunks, 60 ms looked at after every allocation, 64 GiB and a quarter
        // of the free memory held at once, never into the last 8 GiB, not at all on a device somebody else uses).  Larger steps
        // do not get further: an allocation of 4 GiB or more costs 30 ms per GiB, and the allocator serves small requests from near-by
        // memory whatever is held elsewhere.  The fast chunk is KEPT and cut into vectors for this and later solves (one group
        // throughout: a 4 GiB allocation walked in steps of 64 MB never changes class); everything else is given back at once.  One
        // walk per matrix.
        const bool alike = hi < lo * CLASS;
        const Ctx::PlaceTune &T = c.place_tune;
        const size_t CH = T.chunk;
        bool walk = streams(c, A) && (alike || n_slow > 0 || T.force_find_at >= 0) && memo(c, val, nullptr) == nullptr && bytes <= CH / 4;
        if (walk && T.shared_min != ~(size_t)0 && (c.ranks_share_device || (c.mem_total > 0 && c.mem_total - c.mem_free_at_init > T.shared_min))) {
            // somebody else lives on this device (other ranks, another framework's pool): what the walk holds, they cannot have
            if (debug_on()) fprintf(stderr, "[lcg_hip] placement walk: not made, the device is shared (%s; %.1f GiB were in use when the library was initialised)\n",
                                    c.ranks_share_device ? "another rank of this job sits on it" : "memory in use by others",
                                    (double)(c.mem_total - c.mem_free_at_init) / 1073741824.0);
            walk = false;
            c.place_memo.push_back({val, nullptr, 0.f});        // (decided once per matrix)
        }
        if (walk && T.released_max != ~(size_t)0 && c.released_bytes > T.released_max) {
            if (debug_on()) fprintf(stderr, "[lcg_hip] placement walk: not made, the library has given back %.1f GiB in this process (allocations out of recycled memory "
                                    "cost 30-500 ms a call; the walk is for a fresh allocator: the first large system of a process)\n", (double)c.released_bytes / 1073741824.0);
            walk = false;
            c.place_memo.push_back({val, nullptr, 0.f});
        }
        if (walk) {
            const float ours = alike ? lo : hi;         // the kind to get away from
            std::vector<double *> chunks;
            double *found = nullptr; float found_us = 0.f; int rc = 0;
            // (one of the three groups is a single stretch of 96 GiB: a matrix whose stream lies in it -- a fresh box hands out that
            //  stretch first -- has its nearest better place up to 96 chunks away.  Timing a chunk costs 2 ms: every fourth is timed up to
            //  the 32nd, every eighth beyond.)
            // HARD BOUNDS, each looked at after every single allocation: T.max_chunks chunks; T.wall_ms on the clock (a chunk costs
            // 0.02-0.5 ms out of memory nobody has held since the node was provisioned, 30 ms once the driver has to clear it); what is held at once --
            // min(T.hold_max, T.hold_frac x the memory free at the start); never into the last T.keep_free bytes.  (The clock can only
            // be read BETWEEN calls: another chunk is allocated only while the time used plus the dearest allocation so far stays within
            // wall_ms, and the walk is not made at all once the allocator recycles what the library released -- Ctx::released_bytes.)
            const auto w0 = std::chrono::steady_clock::now();
            auto elapsed = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count(); };
            size_t fr0 = 0, tot = 0;
            c.walks_made++;
            c.walk_chunks = 0; c.walk_found = 0; c.walk_held = 0; c.walk_end = "chunk limit";
            if (hipMemGetInfo(&fr0, &tot) != hipSuccess) { (void)hipGetLastError(); fr0 = 0; }
            const size_t hold_cap = std::min(T.hold_max, (size_t)(T.hold_frac * (double)fr0));
            // THREE kinds of places show on the clock of a product that reads a stream AND gathers x (tiled product, 10M rows: 563 / 605 /
            // 647 us -- y beside neither, beside x, beside the stream), so the first chunk that beats `ours` is not yet the answer: ours
            // 647, chunk 609 ended the round-4 walk, and the loop then ran in its slow state (687 us per product against 578 with a 563-us
            // place: profiles/r05_tiled_states.txt, one process in six).  The walk therefore keeps the BEST chunk it has seen and goes on
            // for a few timed chunks behind every find; it ends when nothing better has shown in LOOK_ON timed chunks (or at a bound).
            constexpr int LOOK_ON = 3;
            int timed_chunks = 0, since_find = 0, slower_seen = 0;
            double alloc_sum = 0.0, alloc_max = 0.0;
            for (int q = 0; q < T.max_chunks; q++) {
                if ((chunks.size() + 1 + (found ? 1 : 0)) * CH > hold_cap) { c.walk_end = found ? "found" : "hold limit"; break; }
                size_t fr = 0;
                if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < T.keep_free + CH) { (void)hipGetLastError(); c.walk_end = found ? "found" : "free-memory floor"; break; }
                double *p = nullptr;
                const double a0 = elapsed();
                if (hipMalloc(&p, CH) != hipSuccess) { (void)hipGetLastError(); c.walk_end = found ? "found" : "allocation refused"; break; }
                const double a1 = elapsed() - a0;
                alloc_sum += a1; alloc_max = std::max(alloc_max, a1);
                chunks.push_back(p);
                c.walk_chunks++;
                c.walk_held = std::max(c.walk_held, (chunks.size() + (found ? 1 : 0)) * CH);
                // (the clock is read between calls: the walk goes on only while the time used PLUS its dearest allocation so far fits the
                //  bound -- out of memory the process has held before a chunk costs 30 ms instead of 0.1-0.5)
                if (elapsed() + (T.predict ? alloc_max : 0.0) > T.wall_ms) { c.walk_end = found ? "found" : "wall clock"; break; }
                // (a device whose memory has all been in use since it was booted clears every chunk it hands out, 30 ms per GiB: 128 chunks
                //  would take 4 s.  The walk pays where allocations are cheap -- freshly provisioned nodes -- and says so after ONE chunk elsewhere.)
                if (T.predict && q == 0 && a1 > 5.0) { c.walk_end = "no fresh memory"; break; }
                if (q % (q < 32 ? 4 : 8) != 0) continue;
                float us = 0.f;
                rc = time_output(c, A, x, p, &us);
                if (rc) { c.walk_end = "timing failed"; break; }
                const bool forced = T.force_find_at >= 0 && timed_chunks == T.force_find_at;
                timed_chunks++;
                if (forced) { found = p; found_us = 0.9f * std::min(us, lo); chunks.pop_back(); c.walk_end = "found"; break; }
                if (T.force_find_at < 0) {
                    // a find: clearly faster than the kind to get away from -- or, behind a find, clearly faster than that find
                    const float bar = found ? found_us : ours;
                    if (us * CLASS < bar) {
                        if (found) chunks.insert(chunks.begin(), found);        // (the earlier find goes back with the others)
                        found = p; found_us = us; chunks.pop_back(); since_find = 0;
                    } else if (found) {
                        if (++since_find >= LOOK_ON) { c.walk_end = "found"; break; }
                    } else if (alike && ours * CLASS < us) {
                        // slower than ours: ours are not the slowest kind; after LOOK_ON such chunks and none faster, ours are taken for the fast kind
                        if (++slower_seen >= LOOK_ON) { c.walk_end = "ours are the fast kind"; break; }
                    }
                }
                if (elapsed() > T.wall_ms) { c.walk_end = found ? "found" : "wall clock"; break; }
            }
            if (found && std::strcmp(c.walk_end, "chunk limit") == 0) c.walk_end = "found";
            c.walk_ms = elapsed();
            c.walk_found = found ? 1 : 0;
            if (debug_on()) fprintf(stderr, "[lcg_hip] placement walk: %d chunks of %.0f MiB (%zu given back, %.1f GiB held at most), %s (%.1f us against %.1f), "
                                    "%.1f ms of %.0f allowed (allocations %.1f ms, the slowest %.1f), ended by: %s\n", c.walk_chunks, (double)CH / 1048576.0, chunks.size(), (double)c.walk_held / 1073741824.0,
                                    found ? "a faster place found and kept" : "nothing faster", found_us, ours, c.walk_ms, T.wall_ms, alloc_sum, alloc_max, c.walk_end);
            for (double *p : chunks) { (void)hipFree(p); forget_y(c, p); c.released_bytes += CH; }
            if (rc) { if (found) { (void)hipFree(found); forget_y(c, found); c.released_bytes += CH; } return rc; }
            if (found) {
                // (two arenas at most: an older one that nobody uses goes first)
                for (;;) {
                    std::vector<void *> arenas;
                    for (auto &s : c.scratch) if (s.arena && std::find(arenas.begin(), arenas.end(), s.arena) == arenas.end()) arenas.push_back(s.arena);
                    if (arenas.size() < 2) break;
                    void *victim = nullptr;
                    for (void *a : arenas) {
                        bool busy = false;
                        for (auto &s : c.scratch) if (s.arena == a && s.busy) busy = true;
                        if (!busy) { victim = a; break; }
                    }
                    if (!victim) break;
                    for (size_t i = 0; i < c.scratch.size();)
                        if (c.scratch[i].arena == victim) { forget_y(c, c.scratch[i].p); c.scratch.erase(c.scratch.begin() + (long)i); } else i++;
                    for (size_t i = 0; i < cand.size();) {      // (its slots may stand among the candidates)
                        bool gone = true;
                        for (auto &s : c.scratch) if (s.p == cand[i].p) gone = false;
                        if (cand[i].role == nullptr && gone) cand.erase(cand.begin() + (long)i); else i++;
                    }
                    (void)hipFree(victim);
                    c.released_bytes += CH;
                }
                // the chunk as an arena of the pool: slots of the vector's size (2 MiB-aligned), as many as the solve has roles + 2
                const size_t slot = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
                const size_t want = std::min(CH / slot, n_own + 2);
                for (size_t i = 0; i < want; i++) {
                    double *p = reinterpret_cast<double *>(reinterpret_cast<char *>(found) + i * slot);
                    c.scratch.push_back({p, slot, false, found});
                    if (i > 0) c.place_memo.push_back({val, p, found_us});
                    cand.push_back({nullptr, p, found_us});
                }
                lo = std::min(lo, found_us);
            }
            c.place_memo.push_back({val, nullptr, 0.f});
        }
        // Deal the roles by weight: a role keeps its vector unless one that no heavier role holds is clearly faster.  Positions trade
        // vectors, so every role still has a vector of its own; a vector of the pool that lands in a role joins the solve (the one it
        // displaced stays with the solve, unused, and returns to the pool with the others).
        for (size_t o = 0; o < n_deal; o++) {
            size_t best = o;
            for (size_t j = o + 1; j < cand.size(); j++) if (cand[j].us < cand[best].us) best = j;
            if (best != o && cand[best].us * SAME < cand[o].us) { std::swap(cand[o].p, cand[best].p); std::swap(cand[o].us, cand[best].us); }
        }
        for (size_t i = 0; i < n_own; i++) {
            if (*cand[i].role == cand[i].p) continue;
            *cand[i].role = cand[i].p;
            if (!ws.owns(cand[i].p)) ws.adopt(cand[i].p);
            c.place_moved++;
        }
        c.place_us_chosen = cand[0].us;
        if (debug_on()) {
            fprintf(stderr, "[lcg_hip] placement: %d timed, %d roles moved; first output %.1f -> %.1f us; roles by weight:", c.place_timed, c.place_moved,
                    c.place_us_first, c.place_us_chosen);
            for (size_t i = 0; i < cand.size(); i++) fprintf(stderr, "%s %.1f", i == n_own ? " | idle:" : "", cand[i].us);
            fprintf(stderr, "\n");
        }
        return 0;
    }
};

} // namespace lcgh
