// csr_choice.hip -- which kernel family a matrix (or a stretch of its rows) gets: the statistics the automatic choice measures (column span,
// diagonal-likeness, cache lines of x per entry), the rules for the binned and the tiled product (held against every family
// forced by scripts/choice_regret.py), row ranges, and the chunked product of rows far longer than an LDS window.  The kernels that
// multiply: csr.hip, csr_tiled.hip, csr_binned.hip.
#include <algorithm>
#include <cstring>

#include "csr_plan.hpp"

namespace lcgh {

// ---- scattered columns: which matrices take the two-pass binned product (csr_binned.hip) -----------------
// mean column span (largest - smallest column) of the blocks of 64 rows
__global__ __launch_bounds__(64) void k_span_sum(int n, const int *__restrict__ rowptr, const int *__restrict__ col, unsigned long long *sum)
{
    const long row0 = (long)blockIdx.x * PK_R;
    const int r1 = (int)min((long)n, row0 + PK_R);
    const int s = rowptr[row0], e = rowptr[r1];
    int lo = 0x7fffffff, hi = 0;
    for (int k = s + threadIdx.x; k < e; k += 64) { const int c = col[k]; lo = min(lo, c); hi = max(hi, c); }
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); }
    if (threadIdx.x == 0 && e > s) atomicAdd(sum, (unsigned long long)(hi - lo));
}

static double span_threshold()
{   // LCG_HIP_BINNED_SPAN: mean block span (columns) from which the automatic choice takes the binned product
    static const double v = [] { const char *e = lab_env("LCG_HIP_BINNED_SPAN"); return e ? atof(e) : (double)(1 << 20); }();
    return v;
}

static bool diag_like_measured(const CsrPart &P, hipStream_t s);     // (below)
static bool line_ratio_measured(const CsrPart &P, hipStream_t s);
constexpr double TILED_OVER_BINNED = 1.35;      // in units of span_threshold(): up to this mean block span a band is the tiled product's
static double line_ratio_threshold();

// true when P's products go through the binned format (plan built here on first use)
// P.mean_span (measured once): false when the measurement failed
static bool mean_span_measured(const CsrPart &P, hipStream_t s)
{
    if (P.mean_span >= 0.0) return true;
    unsigned long long *d = nullptr, h = 0;
    const int nb = (P.n_rows + PK_R - 1) / PK_R;
    bool ok = hipMalloc(&d, sizeof h) == hipSuccess && hipMemsetAsync(d, 0, sizeof h, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_span_sum, dim3(nb), dim3(64), 0, s, P.n_rows, P.rowptr, P.col, d);
        ok = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    if (d) hipFree(d);
    if (!ok) { (void)hipGetLastError(); return false; }
    P.mean_span = (double)h / nb;
    return true;
}

bool binned_chosen(const CsrPart &P, hipStream_t s)
{
    if (P.bn_state != 0) return P.bn_state > 0;
    static const int env = [] { const char *e = std::getenv("LCG_HIP_BINNED"); return e ? atoi(e) : -1; }();
    const int mode = env >= 0 ? env : P.bn_mode;
    if (mode == 0 || P.n_cols <= 0 || P.nnz <= 0) { P.bn_state = -1; P.bn_why = mode == 0 ? "switched off" : "empty matrix or unknown column count"; return false; }
    if (mode < 0 && (P.nnz < (1 << 22) || P.n_cols < 1000000)) {
        // automatic: only where x cannot sit in a cache (>= 1M columns) and the matrix is worth a second copy
        P.bn_state = -1; P.bn_why = "automatic mode: fewer than 4M entries or 1M columns"; return false;
    }
    if (mode < 0 && P.n_rows < (1 << 19)) { P.bn_state = -1; P.bn_why = "automatic mode: fewer than 512K rows"; return false; }
    PlanTimer timer(P, s);
    if (mode < 0) {
        if (!mean_span_measured(P, s)) { P.bn_state = -1; return false; }
        // (columns anywhere in a matrix of 1M columns span less than the threshold and are scattered all the same: 243 us binned, 347 tiled,
        //  372 packed at 1M rows)
        if (P.mean_span < span_threshold() && P.mean_span < 0.75 * (double)P.n_cols) {
            P.bn_state = -1; P.bn_why = "automatic mode: the row blocks' mean column span is below the threshold"; return false;
        }
        // between one and two million columns of span the tiled product is still the faster one where its plan accepts the matrix
        // (round 4, N = 1e7, 33 per row, columns drawn per row within +-W: W = 524288 / 786432 tiled 1008 / 1019 us, binned 1670;
        //  W = 1048576 tiled 1768, binned 1683 -- profiles/r04_choice_regret.txt)
        // (a BAND: the span a small part of the width.  Columns anywhere in a matrix of 1-1.5M columns have the same span and are the
        //  binned product's: 226 against 328 us at 1M rows)
        // (round 5: the tiled product's time grows with the span -- 1030 us at a span of 0.79M columns, 1790 at 1.57M, N = 1e7 -- and meets the
        //  binned product's 1640-1700 at about 1.4M: the upper end of this clause went from 2 thresholds to 1.35)
        if (P.mean_span < TILED_OVER_BINNED * span_threshold() && 4.0 * P.mean_span <= (double)P.n_cols && tiled_chosen(P, s)) {
            P.bn_state = -1; P.bn_why = "automatic mode: the tiled product takes it (mean column span below 1.35 thresholds)"; return false;
        }
        // wide, but along diagonals (a stencil on a grid with a million points per plane): every diagonal is a contiguous stream of
        // x for the row-block kernels, whatever the distance between the diagonals
        if (diag_like_measured(P, s) && P.diag_like > 0.5) {
            P.bn_state = -1; P.bn_why = "automatic mode: the columns run along diagonals (the row-block kernels gather contiguously)"; return false;
        }
        if (line_ratio_measured(P, s) && P.line_ratio < line_ratio_threshold()) {
            P.bn_state = -1; P.bn_why = "automatic mode: neighbouring rows share their cache lines of x (block-structured)"; return false;
        }
    }
    const int rc = binned_ready(P, s);      // sets bn_state
    if (rc <= 0) { P.bn_state = -1; return false; }
    return true;
}

// ---- row-random bands: which matrices take the one-pass tiled product (csr_tiled.hip) ------------------------------
// fraction of entries whose column is exactly one more than the entry in the same slot of the row above: ~1 for
// diagonals / stencils (a wavefront's gather is then one contiguous run and the row-block kernels are at their best),
// ~0 when every row draws its own columns
__global__ __launch_bounds__(VB) void k_diag_like(int n, const int *__restrict__ rowptr, const int *__restrict__ col, unsigned long long *sum)
{
    const int i = blockIdx.x * VB + threadIdx.x;
    unsigned cnt = 0;
    if (i + 1 < n) {
        const int a = rowptr[i], b = rowptr[i + 1], c = rowptr[i + 2];
        const int m = min(b - a, c - b);
        for (int s = 0; s < m; s++) cnt += col[a + s] + 1 == col[b + s];
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(sum, (unsigned long long)cnt);
}

// P.diag_like (measured once): false when the measurement failed
static bool diag_like_measured(const CsrPart &P, hipStream_t s)
{
    if (P.diag_like >= 0.0) return true;
    unsigned long long *d = nullptr, h = 0;
    bool ok = hipMalloc(&d, sizeof h) == hipSuccess && hipMemsetAsync(d, 0, sizeof h, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_diag_like, dim3((P.n_rows + VB - 1) / VB), dim3(VB), 0, s, P.n_rows, P.rowptr, P.col, d);
        ok = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    if (d) hipFree(d);
    if (!ok) { (void)hipGetLastError(); return false; }
    P.diag_like = P.nnz > 0 ? (double)h / (double)P.nnz : 0.0;
    return true;
}

// How many DIFFERENT 128-byte lines of x the entries of a 64-row block touch, per entry (1 = every gather a line of its own; a
// stencil with several unknowns per grid point ~0.03: its rows share their lines).  This -- not whether the columns advance by one
// per row -- is what decides whether the row-block kernels crawl: they fetch x by the line.  One wavefront per block marks the
// lines in a 16384-bit table in LDS (hashed) and corrects the count for collisions (linear counting).
__global__ __launch_bounds__(64) void k_line_ratio(int n, const int *__restrict__ rowptr, const int *__restrict__ col, double *sums)
{
    constexpr int M = 16384;
    __shared__ unsigned bits[M / 32];
    const long row0 = (long)blockIdx.x * PK_R;
    const int r1 = (int)min((long)n, row0 + PK_R);
    const int s = rowptr[row0], e = rowptr[r1];
    for (int i = threadIdx.x; i < M / 32; i += 64) bits[i] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    for (int k = s + threadIdx.x; k < e; k += 64) {
        const unsigned h = ((unsigned)(col[k] >> 4) * 2654435761u) >> 18;       // 14 bits
        atomicOr(&bits[h >> 5], 1u << (h & 31));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int c = 0;
    for (int i = threadIdx.x; i < M / 32; i += 64) c += __popc(bits[i]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if (threadIdx.x == 0 && e > s) {
        const double f = (double)c / M;
        const double est = f < 0.999 ? -(double)M * log(1.0 - f) : (double)(e - s);
        atomicAdd(sums, fmin(est, (double)(e - s)));
        atomicAdd(sums + 1, (double)(e - s));
    }
}

// P.line_ratio (measured once): false when the measurement failed
static bool line_ratio_measured(const CsrPart &P, hipStream_t s)
{
    if (P.line_ratio >= 0.0) return true;
    double *d = nullptr, h[2] = {0.0, 0.0};
    bool ok = hipMalloc(&d, sizeof h) == hipSuccess && hipMemsetAsync(d, 0, sizeof h, s) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_line_ratio, dim3((P.n_rows + PK_R - 1) / PK_R), dim3(64), 0, s, P.n_rows, P.rowptr, P.col, d);
        ok = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }
    if (d) hipFree(d);
    if (!ok) { (void)hipGetLastError(); return false; }
    P.line_ratio = h[1] > 0.0 ? h[0] / h[1] : 0.0;
    return true;
}
static double line_ratio_threshold()
{   // LCG_HIP_LINE_RATIO: least share of gathers that have a line of their own for the tiled / binned products to be considered
    static const double v = [] { const char *e = lab_env("LCG_HIP_LINE_RATIO"); return e ? atof(e) : 0.5; }();
    return v;
}

static double tiled_line_ratio_threshold()
{   // the same for the tiled product alone.  Measured at N = 1e7, 33 per row, columns drawn per row inside a band (round 3, one box):
    // line ratio 0.063 / 0.125 / 0.244 / 0.434 (band 2048 / 4096 / 8192 / 16384): tiled 917 / 886 / 689 / 604 us, packed row blocks
    // 728 / 808 / 900 / 1062 us; block-structured stencils sit below 0.05
    static const double v = [] { const char *e = lab_env("LCG_HIP_LINE_RATIO_TILED"); return e ? atof(e) : 0.18; }();
    return v;
}

static double tiled_fill_threshold()
{   // least mean number of entries per (workgroup of 8192 rows, tile of 2048 columns) pair.  Round 3's kernel needed 700; the rewritten one
    // (k_tile_spmv2) beats the binned product down to ~350 (W = 786432 at N = 1e7: 352 per pair, 1019 against 1669 us) and loses from
    // ~260 (W = 1048576: 1768 against 1683)
    static const double fill = [] { const char *e = lab_env("LCG_HIP_TILED_FILL"); return e ? atof(e) : 300.0; }();
    return fill;
}

bool tiled_chosen(const CsrPart &P, hipStream_t s)
{
    if (P.tl_state != 0) return P.tl_state > 0;
    static const int env = [] { const char *e = std::getenv("LCG_HIP_TILED"); return e ? atoi(e) : -1; }();
    const int mode = env >= 0 ? env : P.tl_mode;
    if (mode == 0 || P.n_cols <= 0 || P.nnz <= 0) { P.tl_state = -1; P.tl_why = mode == 0 ? "switched off" : "empty matrix or unknown column count"; return false; }
    if (mode < 0 && (P.nnz < (1 << 22) || P.n_cols < (1 << 18))) { P.tl_state = -1; P.tl_why = "automatic mode: fewer than 4M entries or 256K columns"; return false; }
    PlanTimer timer(P, s);
    double min_fill = 0.0;
    if (mode < 0) {
        if (!diag_like_measured(P, s)) { P.tl_state = -1; return false; }
        if (P.diag_like > 0.5) { P.tl_state = -1; P.tl_why = "automatic mode: the columns run along diagonals (the row-block kernels gather contiguously)"; return false; }
        const bool lr = line_ratio_measured(P, s);
        if (debug_on()) std::fprintf(stderr, "[lcg_hip] tiled choice: diag_like %.3f, line_ratio %.3f\n", P.diag_like, P.line_ratio);
        // (a workgroup of the tiled product owns 8192 rows: below ~1.5M rows it cannot fill the 256 CUs and its time stops falling with the
        //  size -- band of 8192 columns, line ratio 0.24: 144 us at 1M rows against 120 packed, but 150 against 198 at 2M and 278 against 377
        //  at 4M (profiles/r05_choice_regret.txt; round 4 had drawn this line at 4M rows: 33 % and 38 % lost there).  The band of 16384,
        //  0.43, is the tiled product's from 1M rows on: 116 against 140.)
        const double lr_least = P.n_rows < 3 * (1 << 19) ? std::max(0.3, tiled_line_ratio_threshold()) : tiled_line_ratio_threshold();
        if (lr && P.line_ratio < lr_least) {
            P.tl_state = -1; P.tl_why = "automatic mode: neighbouring rows share their cache lines of x (block-structured: the row-block kernels fetch few lines per entry)";
            return false;
        }
        // least mean number of entries per (workgroup of 8192 rows, tile of 2048 columns) pair.  Measured at N = 1e7, 33 per row (round 3):
        // W = 524288, 1052 per pair: tiled 1.01-1.17 ms, binned 1.70; W = 1048576, 527 per pair: tiled 1.96 ms, binned 1.72
        // (a workgroup per 8192 rows: a stretch of 200,000 rows would run on 25 of the 256 CUs)
        if (P.n_rows < (1 << 19)) { P.tl_state = -1; P.tl_why = "automatic mode: fewer than 512K rows"; return false; }
        // A band as wide as the matrix is no band.  Where x is small (fewer than 1.5M columns: 12 MB, at home in the caches) the row-block
        // kernels' time stops growing with the width of the band while the tiled product pays for every tile it touches: 1M rows, a mean
        // block span of 0.39 of the width (the generator's "+-524288"): 231 us tiled against 170 packed; 0.2: 172 against 169; 0.1: 138
        // against 172.  From 2M columns on the tiled product keeps its lead at every width (235 against 369 at the same band).
        if (P.n_cols < 3 * (1 << 19) && mean_span_measured(P, s) && P.mean_span > 0.3 * (double)P.n_cols) {
            P.tl_state = -1; P.tl_why = "automatic mode: the band is as wide as the matrix and x is small (the row-block kernels gather from the caches)";
            return false;
        }
        min_fill = tiled_fill_threshold();
    }
    const int rc = tiled_ready(P, s, min_fill);
    if (rc <= 0) { P.tl_state = -1; return false; }
    return true;
}

// ---- row ranges: one kernel family per stretch of rows ----------------------------------------------------------------------
// The choices above (run blocks / packed columns, tiled, binned) are made for a whole part from whole-part statistics.  A matrix
// that is a stencil in most of its rows and scattered in the rest would take ONE of them for all rows: the packed form is refused
// as soon as one 64-row block spans 2^21 columns, and the binned product costs the structured rows 2-3x.  So the rows are cut
// into chunks of RG_CHUNK, every chunk is classed by the two statistics the whole-part choices use -- the share of entries whose
// column is one more than the entry above them (k_diag_like) and the mean column span of a 64-row block (k_span_sum) --, equal
// neighbours are merged, stretches too small to pay for a launch of their own join a neighbour, and each remaining stretch
// becomes a VIEW of the part (its own row-pointer origin, the same col / val, offsets absolute) that chooses its kernel family
// like any part.  Rows are never reordered and every row is still summed by the kernel it would get in a matrix of its own
// class: the product of a range is bit-identical to the product of that range as a stand-alone matrix.
constexpr int RG_CHUNK = 2048;
constexpr int RG_MAX = 8;           // most ranges (more classes changes than that: no split)
constexpr int RG_ST = 4;            // statistics per 64-row block (k_range_stats)
constexpr int LR_LONG = 1024;       // a row of more entries than this is multiplied by chunks (long_rows_launch): a dense "arrow" row
constexpr int LR_MAXRUNS = 3;       // most stretches of such rows a matrix may have and still be split for them


// per 64-row block: [0] column span, [1] diagonal-like entries, [2] entries, [3] longest row
__global__ __launch_bounds__(64) void k_range_stats(int n, const int *__restrict__ rowptr, const int *__restrict__ col, unsigned int *stats)
{
    const long row0 = (long)blockIdx.x * PK_R;
    const int r1 = (int)min((long)n, row0 + PK_R);
    const int s = rowptr[row0], e = rowptr[r1];
    int lo = 0x7fffffff, hi = 0;
    for (int k = s + threadIdx.x; k < e; k += 64) { const int c = col[k]; lo = min(lo, c); hi = max(hi, c); }
    unsigned cnt = 0;
    const long i = row0 + threadIdx.x;
    int len = i < r1 ? rowptr[i + 1] - rowptr[i] : 0;
    if (i < r1 && i + 1 < n) {
        const int a = rowptr[i], b = rowptr[i + 1], c = rowptr[i + 2];
        const int m = min(b - a, c - b);
        for (int q = 0; q < m; q++) cnt += col[a + q] + 1 == col[b + q];
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_down(lo, off, 64)); hi = max(hi, __shfl_down(hi, off, 64)); cnt += __shfl_down(cnt, off, 64);
        len = max(len, __shfl_down(len, off, 64));
    }
    if (threadIdx.x == 0) {
        unsigned int *st = stats + RG_ST * (long)blockIdx.x;
        st[0] = e > s ? (unsigned)(hi - lo) : 0u; st[1] = cnt; st[2] = (unsigned)(e - s); st[3] = (unsigned)len;
    }
}


bool ranges_chosen(const CsrPart &P, hipStream_t s)
{
    if (P.rg_state != 0) return P.rg_state > 0;
    P.rg_state = -1;
    static const int env = [] { const char *e = std::getenv("LCG_HIP_RANGES"); return e ? atoi(e) : -1; }();
    const int mode = env >= 0 ? env : P.rg_mode;
    const int n = P.n_rows;
    if (mode == 0 || P.end_abs >= 0 || n < 2 * RG_CHUNK || P.n_cols <= 0 || P.nnz <= 0) return false;
    bool only_long = false;             // split for long rows only (small systems)
    if (mode < 0 && P.nnz < (1 << 22)) {
        only_long = true;
        // small systems are not worth several launches -- unless a 64-row block holds many LDS windows of entries (a dense row): then
        // one workgroup would walk it window by window while the rest of the product takes a few microseconds
        if (P.nnz < (1 << 17)) return false;
        int *d = nullptr, h = 0;
        bool ok = hipMalloc(&d, sizeof(int)) == hipSuccess && hipMemsetAsync(d, 0, sizeof(int), s) == hipSuccess;
        if (ok) {
            launch_max_slice(n, PK_R, P.rowptr, d, s);
            ok = hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        }
        if (d) hipFree(d);
        if (!ok) { (void)hipGetLastError(); return false; }
        if (h <= 8 * LdsCfg<double>::CH) return false;
    }
    PlanTimer timer(P, s);
    const int nb = (n + PK_R - 1) / PK_R;
    constexpr int BPC = RG_CHUNK / PK_R;         // blocks per chunk
    const int nc = (nb + BPC - 1) / BPC;
    std::vector<unsigned int> hb(RG_ST * (size_t)nb);
    {
        unsigned int *d = nullptr;
        bool ok = hipMalloc(&d, sizeof(unsigned int) * hb.size()) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(k_range_stats, dim3(nb), dim3(64), 0, s, n, P.rowptr, P.col, d);
            ok = hipMemcpyAsync(hb.data(), d, sizeof(unsigned int) * hb.size(), hipMemcpyDeviceToHost, s) == hipSuccess &&
                 hipStreamSynchronize(s) == hipSuccess;
        }
        if (d) hipFree(d);
        if (!ok) { (void)hipGetLastError(); return false; }
    }
    // class of a stretch from its sums: 0 structured (diagonals / stencils), 1 columns drawn per row inside a band, 2 scattered; -1 empty
    // (a stretch whose columns are drawn per row but too widely for the tiled product -- its workgroups of 8192 rows would find
    //  fewer entries per 2048-column tile than the tiled plan asks for -- is multiplied like scattered columns: the row-block
    //  kernels, its only other choice, pay a cache line per gather there)
    auto classify = [only_long](double span_sum, double dl, double ent, double blocks) {
        if (ent <= 0.0 || blocks <= 0.0) return -1;
        if (only_long || dl / ent > 0.5) return 0;
        const double span = span_sum / blocks;
        if (span >= TILED_OVER_BINNED * span_threshold()) return 2;
        const double fill = 128.0 * (ent / blocks) / ((span + 8192.0) / 2048.0);
        return fill < tiled_fill_threshold() && span >= (double)(1 << 19) ? 2 : 1;
    };
    // class 3: a 64-row block that holds a row of more than LR_LONG entries (a dense row of an "arrow" matrix: constraints, mean values).
    // One such row makes its block's slice larger than any LDS window, the whole part falls to the window-by-window kernel and ONE
    // workgroup walks the row alone (a 10M-entry row: milliseconds).  Its stretch becomes a range of its own whose rows are multiplied in
    // chunks by many workgroups (long_rows_launch); up to LR_MAXRUNS such stretches, else the class is not used.
    bool use_long = true;
    auto block_class = [&](int b) {
        if (use_long && hb[RG_ST * (size_t)b + 3] > (unsigned)LR_LONG) return 3;
        return classify((double)hb[RG_ST * (size_t)b], (double)hb[RG_ST * (size_t)b + 1], (double)hb[RG_ST * (size_t)b + 2], 1.0);
    };
    std::vector<int> cls(nc);
    std::vector<double> cent(nc);
    auto class_chunks = [&]() {
        for (int c = 0; c < nc; c++) {
            double sp = 0.0, dl = 0.0, ent = 0.0, blocks = 0.0;
            bool lng = false;
            for (int b = c * BPC; b < std::min(nb, (c + 1) * BPC); b++)
                if (hb[RG_ST * (size_t)b + 2]) {
                    if (use_long && hb[RG_ST * (size_t)b + 3] > (unsigned)LR_LONG) { lng = true; continue; }       // (its entries would drown the chunk's statistics)
                    sp += hb[RG_ST * (size_t)b]; dl += hb[RG_ST * (size_t)b + 1]; ent += hb[RG_ST * (size_t)b + 2]; blocks += 1.0;
                }
            cls[c] = lng ? 3 : classify(sp, dl, ent, blocks); cent[c] = ent;
        }
        for (int c = 0; c < nc; c++) if (cls[c] < 0) cls[c] = c > 0 ? cls[c - 1] : 0;      // empty chunks follow their predecessor
    };
    class_chunks();
    {
        int longruns = 0;
        for (int c = 0; c < nc; c++) longruns += cls[c] == 3 && (c == 0 || cls[c - 1] != 3);
        if (longruns > LR_MAXRUNS) { use_long = false; class_chunks(); }
    }
    struct Run { int c0, c1, k; double ent; };
    std::vector<Run> runs;
    for (int c = 0; c < nc; c++) {
        if (runs.empty() || runs.back().k != cls[c]) runs.push_back({c, c + 1, cls[c], 0.0});
        runs.back().c1 = c + 1; runs.back().ent += cent[c];
    }
    // A stretch of class 1 too short for the tiled product (it fills the chip from ~1.5M rows) beside a scattered stretch is multiplied WITH
    // it: the binned product takes any columns, and row-block kernels over columns that wide pay a cache line per gather (4M rows, columns
    // drawn within 1.5M of the row: the two 524K-row ends -- their spans are clipped by the matrix's edges -- as ranges of their own 802 us,
    // the whole matrix binned 632: profiles/r05_choice_regret.txt).
    if (mode < 0) {
        auto rows_of = [&](const Run &r) { return (long)(std::min(nb, r.c1 * BPC) - r.c0 * BPC) * PK_R; };
        bool changed = true;
        while (changed) {
            changed = false;
            for (size_t i = 0; i < runs.size(); i++)
                if (runs[i].k == 1 && rows_of(runs[i]) < 3L * (1L << 19) &&
                    ((i > 0 && runs[i - 1].k == 2) || (i + 1 < runs.size() && runs[i + 1].k == 2))) { runs[i].k = 2; changed = true; }
            for (size_t i = 0; i + 1 < runs.size();)
                if (runs[i].k == runs[i + 1].k) { runs[i].c1 = runs[i + 1].c1; runs[i].ent += runs[i + 1].ent; runs.erase(runs.begin() + (long)i + 1); changed = true; }
                else i++;
        }
    }
    // a stretch with fewer entries than a launch of its own is worth joins its larger neighbour (smallest first)
    const double min_ent = (mode > 0 || only_long) ? 1.0 : (double)(1 << 18);      // (only_long: the few stretches there are stand for the dense rows)
    for (;;) {
        if (runs.size() < 2) break;
        // (a stretch of long rows never joins a neighbour: it would take the neighbour's one-window kernels away)
        auto weight = [&](size_t i) { return runs[i].k == 3 ? 1e300 : runs[i].ent; };
        size_t w = 0;
        for (size_t i = 1; i < runs.size(); i++) if (weight(i) < weight(w)) w = i;
        if (weight(w) >= min_ent && runs.size() <= (size_t)RG_MAX) break;
        if (runs[w].k == 3) return false;                   // (more stretches than ranges even so: no split)
        size_t to = w == 0 ? 1 : (w + 1 == runs.size() ? w - 1 : (runs[w - 1].ent >= runs[w + 1].ent ? w - 1 : w + 1));
        if (runs[to].k == 3) {                              // (not into a stretch of long rows either, where there is another neighbour)
            const size_t other = to == w + 1 ? (w > 0 ? w - 1 : to) : (w + 1 < runs.size() ? w + 1 : to);
            if (runs[other].k != 3) to = other;
        }
        runs[to].c0 = std::min(runs[to].c0, runs[w].c0); runs[to].c1 = std::max(runs[to].c1, runs[w].c1); runs[to].ent += runs[w].ent;
        runs.erase(runs.begin() + (long)w);
        for (size_t i = 0; i + 1 < runs.size();)      // neighbours of one class become one stretch
            if (runs[i].k == runs[i + 1].k) { runs[i].c1 = runs[i + 1].c1; runs[i].ent += runs[i + 1].ent; runs.erase(runs.begin() + (long)i + 1); }
            else i++;
    }
    if (runs.size() < 2) return false;
    if (mode < 0) {
        // A split pays where some stretch takes ANOTHER family than the row blocks -- the tiled product (class 1: >= 1.5M rows and 4M
        // entries), the binned product (class 2: >= 1M rows and 4M entries, below), long rows -- or where a few wide blocks (a span of
        // 2^21 columns or more) cost the whole matrix its packed columns while a structured stretch alone would get them.  Where every
        // stretch ends in row-block kernels anyway and the whole matrix can be packed, one launch over all rows is faster than one per
        // stretch: 1M rows, 800K of constant diagonals + 200K scattered, 103.5 us packed whole against 115.9 us in two ranges
        // (profiles/r04_choice_regret.txt, second table: 12 % regret); 4M rows, 3.2M + 0.8M: 434 us whole against 525 us (21 %).
        bool other = false;
        for (const Run &r : runs) {
            const long rows = (long)(std::min(nb, r.c1 * BPC) - r.c0 * BPC) * PK_R;
            if (r.k == 3) other = true;
            if (r.k == 2 && r.ent >= (double)(1 << 22) && rows >= (1L << 20)) other = true;
            if (r.k == 1 && r.ent >= (double)(1 << 22) && rows >= 3L * (1L << 19)) other = true;    // (the tiled product fills the chip from ~1.5M rows)
        }
        if (!other) {
            bool wide = false;
            for (int b = 0; b < nb && !wide; b++) wide = hb[RG_ST * (size_t)b] >= (1u << 21);
            if (!wide) return false;
        }
    }
    // The cuts, to the 64-row block: inside the two chunks that meet at a cut the boundary goes where the fewest blocks end up on
    // the side of the other class (one wide block inside a structured range would cost that whole range its packed columns).
    std::vector<int> cutb(runs.size() + 1);          // in blocks
    cutb[0] = 0; cutb[runs.size()] = nb;
    for (size_t i = 1; i < runs.size(); i++) {
        const int ka = runs[i - 1].k;
        if (runs[i].k == 3 || ka == 3) {
            // around long rows the cut hugs the blocks that hold them: in front of the first such block of the stretch's first chunk,
            // behind the last one of its last chunk (the other blocks of those chunks belong to the neighbours)
            int at;
            if (runs[i].k == 3) {
                at = runs[i].c0 * BPC;
                while (at < std::min(nb, (runs[i].c0 + 1) * BPC) && block_class(at) != 3) at++;
            } else {
                at = std::min(nb, runs[i].c0 * BPC);
                while (at > (runs[i].c0 - 1) * BPC && block_class(at - 1) != 3) at--;
            }
            cutb[i] = std::max(at, cutb[i - 1]);
            continue;
        }
        const int b0 = std::max(cutb[i - 1], (runs[i].c0 - 1) * BPC), b1 = std::min(nb, (runs[i].c0 + 1) * BPC);
        int wrong = 0;
        for (int b = b0; b < b1; b++) { const int k = block_class(b); wrong += k == ka; }     // cut at b0: every block of class a on the wrong side
        int best = wrong, at = b0;
        for (int b = b0; b < b1; b++) {
            const int k = block_class(b);
            wrong += (k >= 0 && k != ka) - (k == ka);                                     // cut behind block b
            if (wrong < best) { best = wrong; at = b + 1; }
        }
        cutb[i] = at;
    }
    // row pointers at the cuts
    std::vector<int> cut(runs.size() + 1);
    for (size_t i = 0; i <= runs.size(); i++) {
        const long r = std::min<long>(n, (long)cutb[i] * PK_R);
        if (hipMemcpyAsync(&cut[i], P.rowptr + r, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return false; }
    RangePlan *R = new RangePlan();
    for (size_t i = 0; i < runs.size(); i++) {
        const int r0 = (int)std::min<long>(n, (long)cutb[i] * PK_R), r1 = (int)std::min<long>(n, (long)cutb[i + 1] * PK_R);
        if (r1 <= r0) continue;
        CsrPart V;
        V.n_rows = r1 - r0; V.nnz = cut[i + 1] - cut[i]; V.rowptr = P.rowptr + r0; V.col = P.col; V.val = P.val;
        V.owned = false; V.padded = r1 < n ? true : P.padded;       // (behind an inner range lies the next range)
        V.end_abs = cut[i + 1]; V.n_cols = P.n_cols;
        V.pk_mode = P.pk_mode; V.bn_mode = P.bn_mode; V.tl_mode = P.tl_mode; V.rg_mode = 0; V.rg_state = -1;
        // a stretch classed as scattered takes the binned product where it is eligible at all (its own mean span may sit just
        // under the whole-matrix threshold: the class was decided chunk by chunk)
        // (not for a short stretch: 200,000 scattered rows of a 1M-row matrix went from 86 to 176 us per product that way -- the binned
        //  passes launch a workgroup per 8192 columns and a wavefront per 2048 rows)
        if (runs[i].k == 2 && P.bn_mode < 0 && V.nnz >= (1 << 22) && V.n_rows >= (1 << 20)) V.bn_mode = 1;
        // nor the tiled product: a workgroup per 8192 rows leaves most of the chip idle below ~1.5M rows (a 524K-row stretch at the end of a
        // 10M-row band took 64 workgroups' time: 2294 us for the whole product against 1790 as ONE tiled product)
        if (P.tl_mode < 0 && V.n_rows < 3 * (1 << 19)) V.tl_mode = 0;
        if (runs[i].k == 3) V.lr_mode = 1;
        R->parts.push_back(V); R->r0.push_back(r0); R->seen.push_back(nullptr);
    }
    if (R->parts.size() < 2) { delete R; return false; }
    P.rg_plan = R; P.rg_state = 1;
    if (debug_on()) {
        std::fprintf(stderr, "[lcg_hip] row ranges of %d rows:", n);
        for (size_t i = 0; i < R->parts.size(); i++) std::fprintf(stderr, " [%d, %d) %ld entries;", R->r0[i], R->r0[i] + R->parts[i].n_rows, (long)R->parts[i].nnz);
        std::fprintf(stderr, "\n");
    }
    return true;
}

// ---- rows far longer than an LDS window (class 3 of the row ranges) -----------------------------------------------------------
// Every row of the stretch is cut into chunks of LR_CHUNK entries; one workgroup sums a chunk (strided over the lanes, four entries in
// flight per lane, a fixed tree at the end), one thread per row then adds the row's chunk sums in chunk order: the same bits from call to
// call.  The stretch is small (a few 64-row blocks around the long rows), so the lists are built on the host from its row pointers.
constexpr int LR_CHUNK = 8192;
struct LongRowPlan {
    int nitems = 0;
    int *k0 = nullptr, *k1 = nullptr;   // [nitems] absolute entry offsets of the chunks
    int *first = nullptr;               // [n_rows + 1] first chunk of every row
    double *partial = nullptr;          // [nitems]
};

__global__ __launch_bounds__(VB) void k_lr_partial(const int *__restrict__ k0, const int *__restrict__ k1, const int *__restrict__ col,
                                                   const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ partial,
                                                   const int *done)
{
    __shared__ double sh[VB / 64];
    if (done && *done) return;
    const int a = k0[blockIdx.x], b = k1[blockIdx.x];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int k = a + (int)threadIdx.x;
    for (; k + 3 * VB < b; k += 4 * VB) {
        int c[4]; double v[4], xv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { c[q] = col[k + q * VB]; v[q] = val[k + q * VB]; }
#pragma unroll
        for (int q = 0; q < 4; q++) xv[q] = x[c[q]];
#pragma unroll
        for (int q = 0; q < 4; q++) acc[q] = fma(v[q], xv[q], acc[q]);
    }
    for (int q = 0; k < b; k += VB, q++) acc[q] = fma(val[k], x[col[k]], acc[q]);
    double t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    t = wave_sum(t);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == WSUM_LANE) sh[w] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
#pragma unroll
        for (int q = 1; q < VB / 64; q++) r += sh[q];
        partial[blockIdx.x] = r;
    }
}

__global__ __launch_bounds__(VB) void k_lr_rows(int n, const int *__restrict__ first, const double *__restrict__ partial, double *__restrict__ y,
                                                const int *done)
{
    if (done && *done) return;
    const int r = blockIdx.x * VB + threadIdx.x;
    if (r >= n) return;
    double t = 0.0;
    for (int i = first[r]; i < first[r + 1]; i++) t += partial[i];
    y[r] = t;
}

void long_rows_free(const CsrPart &P)
{
    LongRowPlan *L = static_cast<LongRowPlan *>(P.lr_plan);
    if (!L) return;
    for (void *p : {(void *)L->k0, (void *)L->k1, (void *)L->first, (void *)L->partial}) if (p) (void)hipFree(p);
    delete L;
    P.lr_plan = nullptr;
}

int long_rows_launch(const CsrPart &P, const double *x, double *y, hipStream_t s, const int *done)
{
    const int n = P.n_rows;
    LongRowPlan *L = static_cast<LongRowPlan *>(P.lr_plan);
    if (!L) {
        PlanTimer timer(P, s);
        std::vector<int> rp((size_t)n + 1);
        HIPCHK(hipMemcpyAsync(rp.data(), P.rowptr, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        std::vector<int> k0, k1, first((size_t)n + 1);
        for (int r = 0; r < n; r++) {
            first[r] = (int)k0.size();
            for (long a = rp[r]; a < rp[r + 1]; a += LR_CHUNK) { k0.push_back((int)a); k1.push_back((int)std::min<long>(rp[r + 1], a + LR_CHUNK)); }
        }
        first[n] = (int)k0.size();
        L = new LongRowPlan();
        L->nitems = (int)k0.size();
        const size_t ni = std::max<size_t>(1, k0.size());
        bool ok = hipMalloc(&L->k0, sizeof(int) * ni) == hipSuccess && hipMalloc(&L->k1, sizeof(int) * ni) == hipSuccess &&
                  hipMalloc(&L->first, sizeof(int) * ((size_t)n + 1)) == hipSuccess && hipMalloc(&L->partial, sizeof(double) * ni) == hipSuccess;
        if (ok && L->nitems)
            ok = hipMemcpyAsync(L->k0, k0.data(), sizeof(int) * k0.size(), hipMemcpyHostToDevice, s) == hipSuccess &&
                 hipMemcpyAsync(L->k1, k1.data(), sizeof(int) * k1.size(), hipMemcpyHostToDevice, s) == hipSuccess;
        if (ok) ok = hipMemcpyAsync(L->first, first.data(), sizeof(int) * first.size(), hipMemcpyHostToDevice, s) == hipSuccess &&
                     hipStreamSynchronize(s) == hipSuccess;          // (the host vectors go out of scope)
        P.lr_plan = L;
        if (!ok) { long_rows_free(P); return fail(hipGetLastError(), "long-row plan", __FILE__, __LINE__); }
    }
    if (L->nitems) hipLaunchKernelGGL(k_lr_partial, dim3(L->nitems), dim3(VB), 0, s, L->k0, L->k1, P.col, P.val, x, L->partial, done);
    hipLaunchKernelGGL(k_lr_rows, dim3((n + VB - 1) / VB), dim3(VB), 0, s, n, L->first, L->partial, y, done);
    HIPCHK(hipGetLastError());
    P.last_kernel = "k_lr_partial + k_lr_rows (long rows multiplied in chunks of 8192 entries)";
    return 0;
}

void ranges_free(const CsrPart &P)
{
    RangePlan *R = static_cast<RangePlan *>(P.rg_plan);
    // the range-by-range product's description lives in the plan (R->desc): nobody may be left pointing into it
    // (lcg_hip_csr_last_kernel / _last_traffic_model between this call and the next product)
    if (R) { for (CsrPart &V : R->parts) free_part(V); delete R; P.last_kernel = ""; }
    P.rg_plan = nullptr; P.rg_state = 0;
    ctx().forget_places();       // another kernel family may stream another copy of the matrix (driver.hpp: Placement)
}

} // namespace lcgh
