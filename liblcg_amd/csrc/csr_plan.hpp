// csr_plan.hpp -- what csr.hip (the row-block kernels, the dispatch, the handle's C ABI) and csr_choice.hip (which kernel family a
// matrix, or a stretch of its rows, gets) share.  Not installed.
#pragma once

#include <chrono>
#include <string>
#include <vector>

#include "devcommon.hpp"

namespace lcgh {

constexpr int PK_R = 64;            // rows per block of the packed form and of every per-block statistic

struct PlanTimer {      // adds the host time of a build (it ends on a drained stream) to the part's plan_ms
    const CsrPart &P; hipStream_t s; std::chrono::steady_clock::time_point t0;
    PlanTimer(const CsrPart &p, hipStream_t st) : P(p), s(st), t0(std::chrono::steady_clock::now()) {}
    ~PlanTimer() { (void)hipStreamSynchronize(s); P.plan_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

struct RangePlan {
    std::vector<CsrPart> parts;
    std::vector<int> r0;            // first row of each range
    std::vector<const char *> seen; // last_kernel of each range when `desc` was composed
    std::string desc;
};

// csr_choice.hip: true when P's products go through that format (statistics measured and plan built on first use)
bool binned_chosen(const CsrPart &P, hipStream_t s);
bool tiled_chosen(const CsrPart &P, hipStream_t s);
bool ranges_chosen(const CsrPart &P, hipStream_t s);
void ranges_free(const CsrPart &P);
void long_rows_free(const CsrPart &P);
int long_rows_launch(const CsrPart &P, const double *x, double *y, hipStream_t s, const int *done);
// csr.hip
template <class V> struct LdsCfg { static constexpr int CH = sizeof(V) == 8 ? 2240 : 1344; };      // entries of the largest LDS window
void launch_max_slice(int n, int R, const int *rowptr, int *out, hipStream_t s);     // out[0] = max over blocks of R rows of their entries (atomicMax; zero it first)

} // namespace lcgh
