/*
 * lcg_hip.h -- C ABI of liblcg_hip.so: an MI355X (gfx950) implementation of the
 * conjugate-gradient iteration hot path of liblcg, behind liblcg's own solver
 * entry points and callback types.
 *
 * Every entry point names the reference interface (path:line under
 * /root/reference/src/lib) it stands in for.  Plain C: opaque handles, POD
 * structs, raw pointers and sizes; no C++/torch types cross this boundary.
 * The C++ header include/lcg_dropin.hpp layers the reference's exact C++
 * signatures (default arguments, std::complex) over these symbols.
 *
 * Memory convention.  The iteration is device resident.  `mem` says where the
 * caller's m (in/out) and B (in) live:
 *   LCG_HIP_MEM_HOST   host pointers; copied in, solved on the GPU, m copied back
 *                      (the behaviour of lcg_solver_cuda, lcg_cuda.cu:103-111,210)
 *   LCG_HIP_MEM_DEVICE device pointers; nothing is copied.
 * Callbacks ALWAYS receive device pointers (as the reference's CUDA callbacks
 * do, lcg_cuda.h:45-46) and must enqueue their work on lcg_hip_get_stream()
 * without synchronising.  The progress callback receives the device m.
 *
 * All functions return 0 / a liblcg status code (util.h:69-90) unless noted;
 * LCG_HIP_E_* (<= -2000) report runtime failures (HIP/RCCL errors, missing GPU).
 * There is no CPU fallback: without a usable GPU every compute entry fails.
 */
#ifndef LCG_HIP_H
#define LCG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ types */

/* util.h:95-148 (bit-compatible, 64 bytes) */
typedef struct lcg_para {
    int    max_iterations;   /* 0 = until convergence */
    double epsilon;          /* in (0,1) */
    int    abs_diff;         /* 0: |g|^2/max(|m|^2,1)   1: |g|/N   (lcg.cpp:208-209) */
    double restart_epsilon;  /* unused by the solvers provided here */
    double step;
    double sigma;
    double beta;
    int    maxi_m;
} lcg_para;

/* util.h:247-273 (24 bytes) */
typedef struct clcg_para {
    int    max_iterations;
    double epsilon;
    int    abs_diff;         /* complex residual is |<r,r>|^2/max(|<m,m>|^2,1): clcg.cpp:295-296 */
} clcg_para;

/* util.h:32-64 (the reference's own type name; the entry points below take it as int) */
typedef enum lcg_solver_enum { LCG_CG = 0, LCG_PCG = 1, LCG_CGS = 2, LCG_BICGSTAB = 3, LCG_BICGSTAB2 = 4, LCG_PG = 5, LCG_SPG = 6 } lcg_solver_enum;
/* util.h:187-221 */
typedef enum clcg_solver_enum { CLCG_BICG = 0, CLCG_BICG_SYM = 1, CLCG_CGS = 2, CLCG_BICGSTAB = 3, CLCG_TFQMR = 4,
                                CLCG_PCG = 5, CLCG_PBICG = 6 } clcg_solver_enum;
/* util.h:69-90 */
enum {
    LCG_SUCCESS = 0, LCG_CONVERGENCE = 0, LCG_STOP = 1, LCG_ALREADY_OPTIMIZIED = 2,
    LCG_UNKNOWN_ERROR = -1024, LCG_INVILAD_VARIABLE_SIZE = -1023, LCG_INVILAD_MAX_ITERATIONS = -1022,
    LCG_INVILAD_EPSILON = -1021, LCG_INVILAD_RESTART_EPSILON = -1020,
    LCG_REACHED_MAX_ITERATIONS = -1019, LCG_NULL_PRECONDITION_MATRIX = -1018, LCG_NAN_VALUE = -1017,
    LCG_INVALID_POINTER = -1016, LCG_INVALID_LAMBDA = -1015, LCG_INVALID_SIGMA = -1014,
    LCG_INVALID_BETA = -1013, LCG_INVALID_MAXIM = -1012, LCG_SIZE_NOT_MATCH = -1011
};
/* util.h:226-242.  NB the complex loops return LCG_REACHED_MAX_ITERATIONS (-1019) and
 * LCG_ALREADY_OPTIMIZIED from the REAL enum (clcg.cpp:126,164); kept as is. */
enum {
    CLCG_SUCCESS = 0, CLCG_CONVERGENCE = 0, CLCG_STOP = 1, CLCG_ALREADY_OPTIMIZIED = 2,
    CLCG_UNKNOWN_ERROR = -1024, CLCG_INVILAD_VARIABLE_SIZE = -1023, CLCG_INVILAD_MAX_ITERATIONS = -1022,
    CLCG_INVILAD_EPSILON = -1021, CLCG_REACHED_MAX_ITERATIONS = -1020, CLCG_NAN_VALUE = -1019,
    CLCG_INVALID_POINTER = -1018, CLCG_SIZE_NOT_MATCH = -1017, CLCG_UNKNOWN_SOLVER = -1016
};
enum {
    LCG_HIP_E_RUNTIME = -2000,   /* a HIP call failed (lcg_hip_last_error() has the text) */
    LCG_HIP_E_NO_DEVICE = -2001, /* no usable gfx950 device */
    LCG_HIP_E_COMM = -2002,      /* RCCL missing or a collective failed */
    LCG_HIP_E_ARG = -2003        /* bad handle / argument to a non-solver entry */
};
enum { LCG_HIP_MEM_HOST = 0, LCG_HIP_MEM_DEVICE = 1 };

/* lcg.h:37-38.  x and prod_Ax are DEVICE pointers. Also the type of M^-1.x (lcg.h:90). */
typedef void (*lcg_axfunc_ptr)(void *instance, const double *x, double *prod_Ax, const int n_size);
/* lcg.h:53-54.  m is the DEVICE solution vector; non-zero return stops with LCG_STOP. */
typedef int (*lcg_progress_ptr)(void *instance, const double *m, const double converge,
                                const lcg_para *param, const int n_size, const int k);
/* clcg.h:40-41.  Complex vectors are interleaved (re,im) doubles == std::complex<double>.
 * layout: 0 MatNormal / 1 MatTranspose; conjugate: 0 NonConjugate / 1 Conjugate
 * (algebra.h:31-50).  The solvers here only ever pass (0,0), as clcg.cpp:463,474,620,630,707,759,771. */
typedef void (*clcg_hip_axfunc_ptr)(void *instance, const double *x, double *prod_Ax,
                                    const int n_size, int layout, int conjugate);
/* clcg.h:56-57 */
typedef int (*clcg_hip_progress_ptr)(void *instance, const double *m, const double converge,
                                     const clcg_para *param, const int n_size, const int k);

/* ------------------------------------------------------- runtime / stream */
int  lcg_hip_init(int device);                 /* selects the device; idempotent. */
int  lcg_hip_set_stream(void *hip_stream);     /* NULL = the library's own stream */
void *lcg_hip_get_stream(void);                /* stream callbacks must launch on */
int  lcg_hip_synchronize(void);   /* also where a timed-out direct exchange (multi-GPU) surfaces: LCG_HIP_E_COMM */
/* Blocking copy on the library stream.  kind: 1 host->device, 2 device->host, 3 device->device
 * (the cudaMemcpy calls of the reference's GPU samples, e.g. sample8.cu:160-166). */
int  lcg_hip_memcpy(void *dst, const void *src, uint64_t bytes, int kind);
const char *lcg_hip_last_error(void);
lcg_para  lcg_hip_default_parameters(void);    /* util.h:153,163 */
clcg_para clcg_hip_default_parameters(void);   /* util.h:278,287 */
/* Facts about the most recent solve on this thread (liblcg reports them only through Pfp). */
int    lcg_hip_last_iterations(void);
double lcg_hip_last_residual(void);
/* Mean device time of the A.x callback over the last solve, in microseconds, from HIP
 * events recorded on the solver stream around each call; 0 unless profiling was enabled.
 * on = 1 times every call, on = k > 1 every k-th call (two stream markers per timed call). */
int    lcg_hip_set_profiling(int on);
double lcg_hip_last_ax_mean_us(void);
int    lcg_hip_last_ax_calls(void);
/* (With the built-in product the solve's SET-UP product y = A.m0 -- lcg.cpp:168, 314, 476, 648 -- carries no event pair and is not
 * counted: whether it is made at all is decided on the device (an all-zero guess needs none), so the figures above are about the
 * products of the iterations.  One consequence for matrices with Inf / NaN ENTRIES: the reference's A.0 already poisons g with
 * 0 x Inf = NaN before the loop, here the first NaN appears one product later -- LCG_NAN_VALUE is returned either way.) */
/* What the latest solve enqueued: vector passes, scalar steps (a step that sums over ranks counts once), reductions over ranks, A.x
 * callbacks (the set-up product included).  Any pointer may be NULL.  bench.py divides by the iterations: launches per iteration. */
int    lcg_hip_last_launches(int *vector_passes, int *scalar_steps, int *rank_reductions, int *products);
/* Always 0.  (Round 3 counted here the scalar steps that ran in the last block of a sharded product -- an opt-in experiment that
 * measured no gain, DESIGN 9, and was retired in round 4; the entry stays so that the library's exports do not change.) */
int    lcg_hip_last_finisher_steps(void);
/* The solvers keep their temporaries (the reference allocates and frees them per call, lcg.cpp:158-166,266-271) for the
 * next solve; this gives the idle ones back to the device. */
int    lcg_hip_trim(void);
/* How plain CG (LCG_CG, lcg.cpp:143-274) -- and PCG with the built-in Jacobi (lpcg, lcg.cpp:293-434: the same rearrangement
 * with u = M^-1 r, w = A u and u.r, w.u in the one reduction) -- is scheduled.  LCG_HIP_CG_CLASSIC: the reference's own
 * recurrence, two reductions per iteration (d.Ad, then m.m/g.g).  LCG_HIP_CG_ONE_REDUCTION: the
 * Chronopoulos-Gear rearrangement of the same recurrence -- w = A.g is applied to the gradient,
 * A.d follows from Ad = beta Ad - w, and g.g, g.w, m.m share ONE reduction (one RCCL all-reduce
 * per iteration instead of two); same iterates in exact arithmetic, same stop rule and counts.
 * LCG_HIP_CG_AUTO (default): one-reduction when the rows are sharded and, on one GPU, for systems of fewer than 2^20 rows
 * (a launch then costs more than the one word per row the rearrangement moves in addition: two launches instead of three) unless A.x or M is a callback of the
 * caller's own -- such a callback sees the reference's sequence of calls (the rearrangement makes one product more before the first
 * stop test); classic otherwise.  tests/test_gpu_solvers.py::test_cg_schedules_on_an_ill_conditioned_system holds both schedules
 * to the oracle's classic loop on a system of condition number 3.6e6 (iteration count, true residual, monitored = true). */
enum { LCG_HIP_CG_AUTO = 0, LCG_HIP_CG_CLASSIC = 1, LCG_HIP_CG_ONE_REDUCTION = 2 };
int    lcg_hip_set_cg_schedule(int schedule);
/* Where the work vectors lie.  The reference allocates its temporaries wherever malloc puts them (lcg.cpp:158-166); on an MI355X
 * the vector a large product WRITES is 8-10 % faster or slower to write depending on whether it shares one of the device's three
 * groups of memory with the matrix's value array (a property of the PAIR of allocations, constant for their lifetime, invisible in
 * any address the process can see: profiles/r04_placement.txt), and the vectors the loop reads beside the matrix pay a
 * little of the same.  Before the first iteration of a real solver whose A.x is lcg_hip_csr_ax the library therefore times y = A.B
 * into each work vector of the solve that it owns itself (and into idle vectors of its pool) and deals the ROLES by weight -- the
 * products' outputs (A.d; A.p and A.s), the vector the product reads, the rest -- the fastest vectors to the heaviest roles.  Where
 * the solve has too few fast vectors, once per matrix: chunks of 1 GiB allocated one after the other until the product into one of
 * them (every fourth, later every eighth, is timed) is clearly faster.  HARD BOUNDS of that walk, each looked at after every single
 * allocation: 128 chunks; 60 ms on the wall clock; at most min(64 GiB, a quarter of the memory that was free when it started) held
 * at once; never into the last 8 GiB of free memory (the clock is read between calls: another chunk is allocated only while the
 * time used plus the dearest allocation so far fits the 60 ms; a first chunk that takes more than 5 ms -- a device that clears
 * what it hands out because all of its memory has been in use since boot -- is the last); no walk at all on a device that is shared (more than 4 GiB were in use by others -- other ranks, the host program's own pool -- when the library was
 * initialised); and no walk once the library has given back more than 1 GiB in the process's life (matrices destroyed, vectors
 * trimmed, an earlier walk's chunks): out of recycled memory ONE 1 GiB allocation costs 30 ms .. 0.5 s, which no bound looked at
 * between calls can hold -- the walk belongs to the first large system of a process.  The chunk found is kept and cut
 * into work vectors (lcg_hip_trim gives it back), the others are given back at once.  A timing that fails means "not tried": the
 * solve goes on with its vectors as allocated.  Roles only: no arithmetic changes, iterates
 * are bit-identical (tests/test_gpu_placement.py).  Results are remembered per (matrix, vector), so later solves time nothing.
 * mode: -1 automatic (real; this process's product streams >= 768 MB), 0 never, 1 for every real matrix with the built-in callback
 * (no walk below that size).  LCG_HIP_PLACE in the environment sets the initial mode. */
int    lcg_hip_set_placement(int mode);
/* The latest solve's placement: vectors timed (0 = everything came from memory, or not tried), roles moved, and what the first
 * output role's product took as allocated / as placed, microseconds (0 when not tried).  Any pointer may be NULL. */
int    lcg_hip_last_placement(int *timed, int *moved, double *us_as_allocated, double *us_as_placed);
/* The pool of work vectors the solvers keep between solves (lcg_hip_trim gives the idle ones back): how many vectors, their bytes, and how
 * many of them are slots of an arena (a 1 GiB chunk the placement kept and cut up).  Any pointer may be NULL. */
int    lcg_hip_pool_info(int *vectors, int64_t *bytes, int *arena_slots);
/* Test hook: one allocation of slots x slot_bytes joins the pool as an arena of `slots` idle vectors -- what the placement's walk leaves
 * behind, without the walk (tests/test_gpu_placement.py: solves take their vectors from it, lcg_hip_trim gives it back as a whole). */
int    lcg_hip_pool_add_arena_for_test(uint64_t slot_bytes, int slots);
/* Test hook: the walk's bounds (0 / negative = the production value): least streamed bytes for a matrix to be placed at all, chunk
 * size, chunk limit, wall-clock limit, most bytes held; force_find_at >= 0 takes the k-th TIMED chunk as the faster place whatever the
 * clock says (so that arena creation and eviction run deterministically at a few million rows); allow_shared = 1 lifts the shared-device
 * rule, the fresh-allocator rule and the look-ahead of the clock (a test walks many times in one process), 2 the shared-device rule only.  tests/test_gpu_placement.py drives walk -> arena -> second matrix -> eviction -> trim with it. */
int    lcg_hip_placement_tune_for_test(uint64_t stream_min_bytes, uint64_t chunk_bytes, int max_chunks, double wall_ms, uint64_t hold_max_bytes,
                                       int force_find_at, int allow_shared);
/* The latest walk: chunks allocated, its wall time (ms), the most bytes it held at once, whether a place was kept, and which bound or
 * finding ended it ("found", "wall clock", "hold limit", "chunk limit", "free-memory floor", "no fresh memory", ...).  Any pointer may be NULL.  Returns the
 * number of walks this process has made so far. */
int    lcg_hip_last_placement_walk(int *chunks, double *wall_ms, int64_t *held_bytes, int *found, const char **ended);

/* ----------------------------------------------------------- solver entry */
/* lcg.h:71-72 lcg_solver() -> lcg.cpp:59-82.  solver_id: LCG_CG, LCG_CGS, LCG_BICGSTAB,
 * LCG_BICGSTAB2; anything else runs CGS exactly as the reference's default branch does. */
int lcg_hip_solver(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B,
                   int n_size, const lcg_para *param, void *instance, int solver_id, int mem);
/* lcg.h:90-91 lcg_solver_preconditioned() -> lpcg, lcg.cpp:293-434 (solver_id ignored, :90). */
int lcg_hip_solver_preconditioned(lcg_axfunc_ptr Afp, lcg_axfunc_ptr Mfp, lcg_progress_ptr Pfp,
                                  double *m, const double *B, int n_size, const lcg_para *param,
                                  void *instance, int solver_id, int mem);
/* lcg.h:111-113 lcg_solver_constrained() -> lcg.cpp:121-140: LCG_SPG runs lspg (lcg.cpp:1224-1446),
 * every other id lpg (lcg.cpp:1054-1204).  low/hig live where m and B live (`mem`). */
int lcg_hip_solver_constrained(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B,
                               const double *low, const double *hig, int n_size, const lcg_para *param,
                               void *instance, int solver_id, int mem);
/* lcg.h:135-137 lcg() with caller workspaces (DEVICE pointers or NULL), lcg.cpp:143-274. */
int lcg_hip_lcg(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, int n_size,
                const lcg_para *param, void *instance, double *Gk, double *Dk, double *ADk, int mem);
/* lcg.h:166-169 lcgs() with caller workspaces (DEVICE pointers or NULL), lcg.cpp:437-612. */
int lcg_hip_lcgs(lcg_axfunc_ptr Afp, lcg_progress_ptr Pfp, double *m, const double *B, int n_size,
                 const lcg_para *param, void *instance, double *RK, double *R0T, double *PK,
                 double *AX, double *UK, double *QK, double *WK, int mem);
/* clcg.h:74-76 clcg_solver() -> clcg.cpp:46-74.  solver_id: CLCG_BICG (the callback is asked
 * for A^H.x, clcg.cpp:187), CLCG_BICG_SYM, CLCG_CGS, CLCG_BICGSTAB, CLCG_TFQMR; other ids run
 * CGS like the reference's default branch.  The shadow residual of CGS/BiCGStab/TFQMR is drawn from
 * lcg_hip_set_shadow_seed() (default 1) instead of srand(time(0)) (lcg_complex.cpp:118-127). */
int clcg_hip_solver(clcg_hip_axfunc_ptr Afp, clcg_hip_progress_ptr Pfp, double *m, const double *B,
                    int n_size, const clcg_para *param, void *instance, int solver_id, int mem);
/* clcg_solver_preconditioned_cuda() (clcg_cuda.h:105-108) -> clpcg (clcg_cuda.cu:403-558): PCG for
 * complex-symmetric A with unconjugated products; monitors |r|^2/max(|m|^2,1) (or |r|/N), as that
 * CUDA loop does.  Mfp has the complex callback type; clcg_hip_jacobi_mx is the ready-made one.
 * solver_id = CLCG_PBICG runs clpbicg instead (clcg_solver_preconditioned_eigen's default, clcg_eigen.h:87-92,
 * clcg_eigen.cpp:685-802): preconditioned BiCG with two products per iteration, A.p and conj(A).ps (the callback's
 * conjugate flag), Eigen's conjugating dot and the CPU loops' 4th-power stop rule (|<r,r>|^2 / max(|<m,m>|^2, 1), or
 * |r|^2 / N with abs_diff); every other id runs clpcg.  Neither loop has a CPU twin in the reference that builds here. */
int clcg_hip_solver_preconditioned(clcg_hip_axfunc_ptr Afp, clcg_hip_axfunc_ptr Mfp, clcg_hip_progress_ptr Pfp,
                                   double *m, const double *B, int n_size, const clcg_para *param,
                                   void *instance, int solver_id, int mem);
int lcg_hip_set_shadow_seed(unsigned seed);
/* Replace the drawn shadow residual by an explicit vector (n complex, host memory) for the
 * next complex solve only; lets a test replay the reference's own rbar0. */
int lcg_hip_set_shadow_vector(const double *rbar0_host, int n_size);

/* --------------------------------------------------------------- matrices */
/* A CSR matrix resident in HBM: int32 rowptr[n+1] / col[nnz] (base 0), fp64 or c128 val.
 * This is what the reference's GPU samples assemble with cusparseXcoo2csr +
 * cusparseCreateCsr (sample8.cu:169-173, sample10.cu:177-181). */
typedef struct lcg_hip_csr *lcg_hip_csr_t;

/* Copy (mem == HOST, or DEVICE with adopt == 0) or adopt without copying (DEVICE, adopt != 0;
 * the caller keeps the arrays alive) a CSR matrix.  is_complex: val holds interleaved c128.
 * adopt == 2 additionally promises >= 64 readable bytes after col[nnz] and val[nnz], which
 * admits the fastest A.x kernel (copied matrices always have that slack). */
int lcg_hip_csr_create(lcg_hip_csr_t *A, int n_rows, int n_cols, int64_t nnz, const int *rowptr,
                       const int *col, const double *val, int is_complex, int mem, int adopt);
/* COO (row-sorted or not) -> CSR on the device: data/README:1-10 files, sample8.cu:30-64,169. */
int lcg_hip_csr_from_coo(lcg_hip_csr_t *A, int n, int64_t nnz, const int *row, const int *col,
                         const double *val, int is_complex, int mem);
int lcg_hip_csr_destroy(lcg_hip_csr_t A);
int lcg_hip_csr_rows(lcg_hip_csr_t A);
int64_t lcg_hip_csr_nnz(lcg_hip_csr_t A);
/* Device pointers of the arrays (for callers that want to inspect or reuse them). */
int lcg_hip_csr_arrays(lcg_hip_csr_t A, const int **rowptr, const int **col, const double **val);
/* Select the SpMV kernel: 0 auto, otherwise lanes per row (2,4,...,64) for the
 * wavefront kernel, or -1 for the LDS-staged variant. */
int lcg_hip_csr_set_kernel(lcg_hip_csr_t A, int variant);
/* Packed column indices for the LDS-staged real A.x (64 rows per block, block-relative columns of 18
 * or 21 bits, seven or six per 16 bytes: 10.3 / 10.7 instead of 12 bytes of stream per entry, -9 % time
 * on the headline system, bit-identical y).  Built on the device at the first product, kept beside the plain
 * column array (+2.67 B per entry).  mode: -1 automatic (matrices of >= 4M entries whose blocks
 * span < 2^21 columns), 0 never (frees the packed copy), 1 whenever eligible.  LCG_HIP_PACKED=0/1
 * overrides for the whole process. */
int lcg_hip_csr_set_packed(lcg_hip_csr_t A, int mode);
/* Blocks of the packed form stored as RUNS: every row of the block has the same number of entries and every entry's column is
 * one more than the entry in the same slot of the row above (constant diagonals, stencils away from the edges).  Such a
 * block carries row 0's columns only -- 8.06 instead of 10.3 bytes of stream per entry -- and its x gathers leave together
 * with the value stream (-22 % on the headline A.x, bit-identical y).  Returns the number of run blocks of the packed copy
 * (0 before the first product built it, or when there are none); *blocks_out (may be NULL) = all blocks of 64 rows.
 * (A LAB build's LCG_HIP_PACKED_RUNS=0 stores every block with its own columns.) */
int64_t lcg_hip_csr_packed_runs(lcg_hip_csr_t A, int64_t *blocks_out);
/* Blocks of 64 rows the packed form stores as TEMPLATE blocks: every entry on one of <= 64 diagonals (the union of the rows'), rows of
 * <= 32 entries, and a 64-bit mask per row saying which diagonals the row has -- what a stencil's blocks look like where grid boundaries pass
 * through them.  Like run blocks they stream values only and are bit-identical to the plain row-block kernel.
 * (A LAB build's LCG_HIP_PACKED_TEMPLATES=0 stores such blocks with packed columns.)  No reference counterpart. */
int64_t lcg_hip_csr_packed_templates(lcg_hip_csr_t A);
/* Two-pass "binned" A.x for matrices whose columns are scattered over more of x than any cache holds (the
 * arbitrary user CSR of sample8.cu:96-103 at its worst): pass 1 expands x into entry order with a 64 KB slice
 * of x in LDS per workgroup, pass 2 streams val and the expanded x and sums the rows in LDS (one wavefront per
 * 2048 rows, ds_add_f64) -- 28.5 streamed bytes per entry instead of 12 + a cache line per gather.  The plan
 * (re-ordered copy of the matrix: +26.5 B per entry) is built on the device at the first product.
 * mode: -1 automatic (real matrices of >= 4M entries and >= 1M columns whose 64-row blocks span on average
 * >= 2^20 columns, whose columns do not run along diagonals and
 * whose gathers mostly have a cache line of x of their own (share >= 0.5: block-structured
 * matrices stay with the row-block kernels), 0 never (frees the plan), 1 whenever eligible.
 * LCG_HIP_BINNED=0/1 overrides for the whole process.  y differs from the row-block kernels' y in the last
 * bits (products are rounded before the add); on gfx950 it is bit-identical from call to call and from plan to plan
 * (lanes of one ds_add_f64 that meet in a row are serialised by the LDS in lane order: verified by test, not an ISA promise).
 * The packed, tiled and binned forms are COPIES of the matrix made at the first product: a caller who rewrites the
 * arrays of an adopted matrix (lcg_hip_csr_create with adopt != 0) afterwards drops them with set_*(A, 0) and
 * re-arms the automatic choice with set_*(A, -1). */
int lcg_hip_csr_set_binned(lcg_hip_csr_t A, int mode);
/* One-pass "tiled" A.x for matrices whose rows draw their columns at random from a band: the row-block kernels
 * find x in the L2 there but move a 128-byte line per 8-byte gather.  A workgroup owns 8 x 1024 rows (sums in LDS),
 * walks the column tiles they touch (2048 columns; a loader wavefront copies the next tile of x into LDS by LDS-DMA while
 * eight consumer wavefronts stream the rows' entries of the current one: 2 KB per step of 192 entries = values + three
 * 21-bit (row, column) pairs per 64-bit word, 10.67 B per entry where CSR has 12).  mode: -1 automatic (real matrices of
 * >= 4M entries whose columns do not run along diagonals, whose gathers mostly have a cache line of x of their own
 * (share >= 0.18) and whose (workgroup, tile) pairs hold >= 700 entries on average), 0 never (frees the plan), 1 whenever eligible;
 * LCG_HIP_TILED=0/1 overrides for the process.  Same last-bit deviation from the row-block kernels as the binned product; y is bit-identical from call to
 * call and from plan to plan ON gfx950 (a row is summed by one wavefront in stream order; lanes of one ds_add_f64 that meet
 * in a row are serialised by the LDS in lane order -- verified by tests/test_gpu_binned.py, not promised by the ISA). */
int lcg_hip_csr_set_tiled(lcg_hip_csr_t A, int mode);
const char *lcg_hip_csr_tiled_status(lcg_hip_csr_t A);
/* Row ranges: a matrix whose rows fall into different column-pattern classes (a stencil in most rows, scattered columns in the
 * rest) is multiplied range by range, every range choosing its own kernel family among the ones above; the rows are cut where
 * the class of their 2048-row chunks changes (share of entries that continue a diagonal, mean column span of a 64-row block).
 * A range's product is bit-identical to the product of that range as a matrix of its own.  mode: -1 automatic (real matrices of
 * >= 4M entries, ranges of >= 256K entries, at most 8), 0 never, 1 whenever two classes are found; LCG_HIP_RANGES=0/1 overrides
 * for the process.  lcg_hip_csr_last_kernel then reads "rows [a, b): <kernel> | rows [b, c): <kernel> ...".
 * The reference has no counterpart (its A.x is the user's callback: lcg.h:37-38); speed only. */
int lcg_hip_csr_set_ranges(lcg_hip_csr_t A, int mode);
/* Number of row ranges the latest product used (0: one kernel family for all rows); first_row[0 .. min(cap, ranges)) = their first rows. */
int lcg_hip_csr_ranges(lcg_hip_csr_t A, int cap, int *first_row);
/* Why A has (or has not) a binned plan: "ready", or the reason it is not used (static string). */
const char *lcg_hip_csr_binned_status(lcg_hip_csr_t A);
/* Name of the kernel family the latest product with A used (static string; "" before the first product). */
const char *lcg_hip_csr_last_kernel(lcg_hip_csr_t A);
/* Bytes the latest product's kernels stream from / to memory per call BY CONSTRUCTION -- what must move for the kernel
 * family that ran: CSR row-block kernels 12 B per entry; packed columns 8 B + 16 B per group of 6-7 columns (run blocks:
 * row 0's columns only); tiled product 2 KB per step of 192 entries (10.67 B per entry) + tile lists; binned product both
 * passes (28.5 B per entry) -- each plus row pointers, x once and y.  0 for complex matrices and before the first product.
 * (SURVEY's algorithmic figure, 12 nnz + 4 (N + 1) + 16 N, is what bench.py's roofline.achieved is quoted on; this is the
 * physical companion: bench.py's roofline.must_move_bytes.) */
int64_t lcg_hip_csr_last_traffic_model(lcg_hip_csr_t A);
/* What the first product spent on A beside the CSR arrays: *build_ms = host time of choosing a kernel family and building
 * its copy of the matrix (packed columns / tiled stream / binned streams; the builders drain the stream), *extra_bytes =
 * device memory those copies hold.  Either pointer may be NULL. */
int lcg_hip_csr_plan_info(lcg_hip_csr_t A, double *build_ms, int64_t *extra_bytes);
/* Extract the diagonal and keep its reciprocal for lcg_hip_jacobi_mx
 * (lcg_smDcsr_get_diagonal algebra_cuda.cu:40-57,85-92; clcg_smZcsr_get_diagonal
 * lcg_complex_cuda.cu:46-63).  diag_out (device, n values) may be NULL. */
int lcg_hip_csr_build_jacobi(lcg_hip_csr_t A, double *diag_out);

/* Ready-made callbacks; pass the lcg_hip_csr_t as `instance`. */
void lcg_hip_csr_ax(void *instance, const double *x, double *prod_Ax, const int n_size);      /* cudaAx, sample8.cu:96-103 */
void lcg_hip_jacobi_mx(void *instance, const double *x, double *prod_Mx, const int n_size);   /* sample1.cpp:55-62 (z = x/diag, reciprocal form) */
void clcg_hip_csr_ax(void *instance, const double *x, double *prod_Ax, const int n_size,
                     int layout, int conjugate);                                                /* sample10.cu:90-97 */
void clcg_hip_jacobi_mx(void *instance, const double *x, double *prod_Mx, const int n_size,
                        int layout, int conjugate);                                             /* sample10.cu:99-120 (Jacobi branch) */

/* ---------------------------------------------------------------- kernels */
/* Stand-alone launches of the hot-path kernels on the current stream (device pointers).
 * Scalar results are written to host memory after a stream synchronise. */
int lcg_hip_spmv(lcg_hip_csr_t A, const double *x, double *y);                  /* y = A.x */
/* y = op(A).x, layout 1 = transpose, conjugate 1 = conjugated entries (A^H = both): the
 * MatTranspose / Conjugate products the reference's callbacks are asked for (clcg.h:40-41,
 * clcg.cpp:187; cusparseSpMV with CUSPARSE_OPERATION_*TRANSPOSE in sample9.cu:97). */
int lcg_hip_spmv_op(lcg_hip_csr_t A, const double *x, double *y, int layout, int conjugate);
/* (On a sharded matrix: layout = 1 only -- every rank multiplies the transpose of its rows with its slice of x and
 * ncclReduceScatter sums the contributions into the ranks' row blocks; conj(A).x alone returns LCG_HIP_E_RUNTIME.) */
/* y = A.x together with the sums the Krylov loops take right after it (lcg.cpp:234 d.Ad, :548-552 Ap.r0, :735-740 As.s and
 * As.As): result2[0] = y.u, result2[1] = y.y (host, after a stream synchronise; NULL = enqueue only).  Where the kernel family
 * allows, the sums ride in the product's epilogue (what the built-in solvers use on one GPU: lcg_hip_csr_last_kernel says
 * "carrying the dot"); otherwise product and reduction run as two launches.  Real matrices. */
int lcg_hip_spmv_dot(lcg_hip_csr_t A, const double *x, double *y, const double *u, double *result2);
int lcg_hip_dot(int n, const double *a, const double *b, double *result);      /* lcg_dot, algebra.cpp:154-163; cublasDdot lcg_cuda.cu:187 */
int lcg_hip_nrm2(int n, const double *a, double *result);                      /* cublasDznrm2-style 2-norm */
int lcg_hip_axpy(int n, double alpha, const double *x, double *y);             /* y += alpha*x, cublasDaxpy lcg_cuda.cu:190 */
int lcg_hip_scal(int n, double alpha, double *x);                              /* cublasDscal lcg_cuda.cu:203 */
int lcg_hip_set2box(int n, const double *low, const double *hig, double *a);    /* a = clamp(a): lcg_set2box_cuda, algebra_cuda.cu:26-38,79-85 */
int lcg_hip_vecmul(int n, const double *a, const double *b, double *c);        /* lcg_vecMvecD_element_wise, algebra_cuda.cu:59-67 */
int lcg_hip_vecdiv(int n, const double *a, const double *b, double *c);        /* lcg_vecDvecD_element_wise, algebra_cuda.cu:69-77 */
int clcg_hip_dot(int n, const double *a, const double *b, double *result2);    /* clcg_dot (unconjugated), lcg_complex.cpp:143-154 */
int clcg_hip_inner(int n, const double *a, const double *b, double *result2);  /* clcg_inner (conj a), lcg_complex.cpp:156-167 */
int clcg_hip_axpy(int n, const double *alpha2, const double *x, double *y);    /* cublasZaxpy clcg_cuda.cu */
int clcg_hip_vecdiv(int n, const double *a, const double *b, double *c);       /* clcg_vecDvecZ_element_wise, lcg_complex_cuda.cu:95-103 */

/* ----------------------------------------------------- synthetic systems */
/* The benchmark family of BASELINE.json configs 2-5 (definition: DESIGN.md, CPU twin:
 * oracle/csr_oracle.c).  Rows [r0,r1) of the n-row matrix are generated on the device with
 * GLOBAL column indices. band > 0: banded variant, band == 0: scrambled affine maps. */
int lcg_hip_csr_generate(lcg_hip_csr_t *A, int64_t n, int npairs, int64_t band, int symmetric,
                         uint64_t seed, double diag_shift, int64_t r0, int64_t r1);
/* The same family with the column pattern named (twin: orc_gen_init_ex):
 *   LCG_HIP_GEN_SCRAMBLED        columns (a_k*i + c_k) mod n and the inverse maps: anywhere in the matrix
 *   LCG_HIP_GEN_DIAGONALS        columns i +- c_k, the SAME npairs offsets c_k <= band in every row
 *                                (2*npairs constant diagonals: a DIA matrix stored as CSR)
 *   LCG_HIP_GEN_ROW_RANDOM_BAND  every row draws its own columns inside (i - band, i + band): rows are cut into
 *                                blocks of the largest power of two <= band/2 and map k sends block b onto block
 *                                b+1 by a keyed bijection of the in-block position (so the mirror entry is
 *                                computable and A stays symmetric without a transpose) */
enum { LCG_HIP_GEN_SCRAMBLED = 0, LCG_HIP_GEN_DIAGONALS = 1, LCG_HIP_GEN_ROW_RANDOM_BAND = 2 };
int lcg_hip_csr_generate_ex(lcg_hip_csr_t *A, int64_t n, int npairs, int pattern, int64_t band, int symmetric,
                            uint64_t seed, double diag_shift, int64_t r0, int64_t r1);
int lcg_hip_gen_xtrue(int64_t n, uint64_t seed, int64_t r0, int64_t r1, double *x_dev);
/* 5-point Laplacian on an nx x ny grid (BASELINE.json config 2), rows [r0,r1). */
int lcg_hip_csr_laplace2d(lcg_hip_csr_t *A, int nx, int ny, int64_t r0, int64_t r1);

/* ------------------------------------------------------------ multi-GPU */
/* One process per GPU.  The row range [r0,r1) of every vector and of A lives on this rank;
 * A.x all-gathers x over RCCL, every inner product is summed with an RCCL all-reduce
 * (nothing comparable exists in the reference: SURVEY.md sections 2 #22-23, 8e).
 * Rendezvous: rank 0 calls lcg_hip_comm_unique_id(), ships the 128 bytes to the other
 * ranks by any means (e.g. a torch.distributed broadcast), then all call comm_init. */
int lcg_hip_comm_unique_id(void *id128);
int lcg_hip_comm_init(int nranks, int rank, const void *id128);
int lcg_hip_comm_destroy(void);
int lcg_hip_comm_rank(void);
int lcg_hip_comm_size(void);
/* Path of the shared object the collectives were bound from ("" before the first lcg_hip_comm_* call).  By default the copy of
 * librccl already mapped into the process (the host program's), else librccl.so from the loader's path; LCG_HIP_RCCL_LIB=<path>
 * binds exactly that file instead, privately (a site's own RCCL build; tests/fake_rccl for several ranks on one GPU). */
const char *lcg_hip_comm_library(void);
/* Bind a row shard (created with GLOBAL column indices over n_global columns) to the
 * communicator: splits it into local-column and remote-column parts and sizes the gather
 * buffer.  rows_per_rank = ceil(n_global / nranks); rank r owns rows
 * [r*rows_per_rank, min(n_global,(r+1)*rows_per_rank)).  mode: 0 all-gather, 1 neighbour
 * (halo) exchange of only the referenced entries (both RCCL, on a second stream beside the
 * local product), 2 direct neighbour exchange: needs the mailboxes (lcg_hip_p2p_*), uses no
 * collective call and no second stream -- the owner's A.x kernel carries a few extra blocks that
 * write its boundary entries of x into the neighbours' receive buffers over the peer mappings
 * and raise a flag there; the neighbours' remote-column product waits for that flag.  At most 8
 * neighbours per rank; returns LCG_HIP_E_COMM on EVERY rank when any rank cannot set it up, so
 * that all of them can fall back to mode 1 or 0 together. */
int lcg_hip_csr_distribute(lcg_hip_csr_t A, int64_t n_global, int mode);
/* x entries this rank receives per A.x under the chosen mode (plan volume, for reporting). */
int64_t lcg_hip_csr_exchange_volume(lcg_hip_csr_t A);
int lcg_hip_allreduce_sum(double *dev_values, int count);
int lcg_hip_barrier(void);
/* Direct all-reduce over xGMI peer mappings (optional; RCCL stays the fallback and the courier
 * of x).  The <= 8 sums of a sync point are a latency problem, not a bandwidth one: every rank
 * owns a small uncached mailbox in its HBM, maps every peer's mailbox through HIP IPC, and the
 * scalar kernel of a sync point writes its sums straight into all peers' mailboxes, waits for
 * theirs and adds them in rank order -- reduce, exchange and scalar step in ONE kernel instead
 * of reduce | ncclAllReduce | scalar step, and the same bits on every rank.
 *   lcg_hip_p2p_export   allocate my mailbox, return its 64-byte IPC handle
 *   lcg_hip_p2p_connect  handles = nranks x 64 bytes in rank order (shipped like the unique id)
 *   lcg_hip_p2p_selftest `rounds` all-reduces of known values with a short timeout; 0 = all correct
 *   lcg_hip_p2p_enable   route the solvers' sync points (and lcg_hip_allreduce_sum) through it;
 *                        the caller makes sure all ranks pass the same value
 *   lcg_hip_p2p_status   0 = not connected, 1 = connected, 2 = enabled, -1 = an exchange timed out
 *                        (synchronises the stream; the path must not be used any more)
 * A contribution that does not arrive within the timeout (default 20 s, LCG_HIP_P2P_TIMEOUT_MS)
 * ends the solve on every rank with LCG_HIP_E_COMM instead of hanging. */
#define LCG_HIP_P2P_HANDLE_BYTES 64
int lcg_hip_p2p_export(void *handle64);
int lcg_hip_p2p_connect(int nranks, int rank, const void *handles);
int lcg_hip_p2p_selftest(int rounds);
int lcg_hip_p2p_enable(int on);
int lcg_hip_p2p_status(void);
/* Time-out of the waits on peer data (mailbox sums at once; a matrix's direct exchange from its next
 * lcg_hip_csr_distribute(A, n, 2) on): short while a node's links are being probed, long in production. */
int lcg_hip_p2p_set_timeout_ms(int ms);
int lcg_hip_p2p_disconnect(void);
/* Test hooks for the sharded product on ONE GPU: split a shard as rank `rank` of `nranks`
 * with no communicator; the caller fills the other ranks' slices of the gather buffer
 * (lcg_hip_csr_xfull) itself and then calls lcg_hip_spmv. */
int lcg_hip_csr_split_for_test(lcg_hip_csr_t A, int64_t n_global, int nranks, int rank);
double *lcg_hip_csr_xfull(lcg_hip_csr_t A);
/* Measurement hook: ONE PART of a sharded product alone (stream-ordered like the product; collective for part 1) -- 1: the x exchange
 * of modes 0 / 1 on the second stream, fork and join included; 2: the local-column product; 4: the remote-column part out of whatever
 * the gather buffer holds.  bench.py's comm_probe times them so that a multi-GPU line says where an iteration goes. */
int lcg_hip_csr_ax_part_for_probe(lcg_hip_csr_t A, const double *x, double *y, int part);
/* After split_for_test and after filling the gather buffer: run the direct exchange (mode 2) with
 * this rank standing in for its neighbours -- the real kernel chain (pushing blocks in the A.x grid,
 * flags, waiting remote-column product) on one GPU, with the true product as result. */
int lcg_hip_csr_direct_selfloop_for_test(lcg_hip_csr_t A, int nranks, int rank);
int64_t lcg_hip_csr_local_nnz(lcg_hip_csr_t A);
int lcg_hip_csr_need_ranges_for_test(lcg_hip_csr_t A, int nranks, int64_t *lohi /* [2*nranks] */);

#ifdef __cplusplus
}
#endif
#endif /* LCG_HIP_H */
