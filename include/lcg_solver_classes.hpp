// lcg_solver_classes.hpp -- liblcg's class front ends (solver.h:32-283, solver.cpp:33-310) over
// the drop-in entry points: derive, override AxProduct / MxProduct (and Progress if wanted), call
// Minimize / MinimizePreconditioned / MinimizeConstrained.  Same member names, defaults and reporting behaviour;
// as everywhere in this library the vectors handed to AxProduct / MxProduct / Progress are DEVICE
// pointers and work is enqueued on lcg_hip_get_stream().
#ifndef LCG_SOLVER_CLASSES_HPP
#define LCG_SOLVER_CLASSES_HPP

#include <chrono>
#include <iostream>

#include "lcg_dropin.hpp"

class LCG_Solver {
protected:
    lcg_para param_;
    unsigned int inter_;
    bool silent_;

public:
    LCG_Solver() : param_(lcg_default_parameters()), inter_(1), silent_(false) {}     // solver.cpp:33-38
    virtual ~LCG_Solver() {}

    virtual void AxProduct(const lcg_float *a, lcg_float *b, const int num) = 0;
    virtual void MxProduct(const lcg_float *a, lcg_float *b, const int num) = 0;
    virtual int Progress(const lcg_float *, const lcg_float converge, const lcg_para *param, const int, const int k)
    {   // solver.cpp:40-54
        if ((inter_ > 0 && k % inter_ == 0) || converge <= param->epsilon)
            std::clog << "\rIteration-times: " << k << "\tconvergence: " << converge;
        return 0;
    }

    void silent() { silent_ = true; }
    void set_report_interval(unsigned int inter) { inter_ = inter; }
    void set_lcg_parameter(const lcg_para &in_param) { param_ = in_param; }

    void Minimize(lcg_float *m, const lcg_float *b, int x_size, lcg_solver_enum solver_id = LCG_CG,
                  bool verbose = true, bool er_throw = false)
    {   // solver.cpp:74-125
        run([&](lcg_progress_ptr P) { return lcg_solver(_AxProduct, P, m, b, x_size, &param_, this, solver_id); },
            name_of(solver_id), verbose, er_throw);
    }
    void MinimizePreconditioned(lcg_float *m, const lcg_float *b, int x_size, lcg_solver_enum solver_id = LCG_PCG,
                                bool verbose = true, bool er_throw = false)
    {   // solver.cpp:127-168
        run([&](lcg_progress_ptr P) { return lcg_solver_preconditioned(_AxProduct, _MxProduct, P, m, b, x_size, &param_, this, solver_id); },
            "PCG", verbose, er_throw);
    }

    void MinimizeConstrained(lcg_float *m, const lcg_float *b, const lcg_float *low, const lcg_float *hig, int x_size,
                             lcg_solver_enum solver_id = LCG_PG, bool verbose = true, bool er_throw = false)
    {   // solver.h:174-176, solver.cpp:171-212 (the report names PG-CG / SPG-CG, anything else "Unknown")
        run([&](lcg_progress_ptr P) { return lcg_solver_constrained(_AxProduct, P, m, b, low, hig, x_size, &param_, this, solver_id); },
            solver_id == LCG_PG ? "PG-CG" : (solver_id == LCG_SPG ? "SPG-CG" : "Unknown"), verbose, er_throw);
    }

    // thunks handed to the C entry points (solver.h:51-54,73-76,98-102)
    static void _AxProduct(void *instance, const lcg_float *a, lcg_float *b, const int num)
    {
        static_cast<LCG_Solver *>(instance)->AxProduct(a, b, num);
    }
    static void _MxProduct(void *instance, const lcg_float *a, lcg_float *b, const int num)
    {
        static_cast<LCG_Solver *>(instance)->MxProduct(a, b, num);
    }
    static int _Progress(void *instance, const lcg_float *m, const lcg_float converge, const lcg_para *param,
                         const int n_size, const int k)
    {
        return static_cast<LCG_Solver *>(instance)->Progress(m, converge, param, n_size, k);
    }

private:
    static const char *name_of(lcg_solver_enum id)
    {
        switch (id) {
        case LCG_CG: return "CG";
        case LCG_CGS: return "CGS";
        case LCG_BICGSTAB: return "BICGSTAB";
        case LCG_BICGSTAB2: return "BICGSTAB2";
        default: return "Unknown";
        }
    }
    template <class F> void run(F &&solve, const char *name, bool verbose, bool er_throw)
    {
        if (silent_) {      // solver.cpp:77-82: no progress callback => fully asynchronous device loop
            const int ret = solve(nullptr);
            if (ret < 0) lcg_error_str(ret, true);
            return;
        }
        const auto t0 = std::chrono::steady_clock::now();
        const int ret = solve(_Progress);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (!er_throw) std::clog << std::endl << "Solver: " << name << ". Time cost: " << ms << " ms" << std::endl;
        if (verbose || ret < 0) {
            if (!er_throw) std::clog << lcg_status_text(ret) << std::endl;
            lcg_error_str(ret, er_throw);
        }
    }
};

class CLCG_Solver {
protected:
    clcg_para param_;
    unsigned int inter_;
    bool silent_;

public:
    CLCG_Solver() : param_(clcg_default_parameters()), inter_(1), silent_(false) {}   // solver.cpp:214-219
    virtual ~CLCG_Solver() {}

    virtual void AxProduct(const lcg_complex *x, lcg_complex *prod_Ax, const int x_size, lcg_matrix_e layout,
                           clcg_complex_e conjugate) = 0;
    virtual int Progress(const lcg_complex *, const lcg_float converge, const clcg_para *param, const int, const int k)
    {   // solver.cpp:221-235
        if ((inter_ > 0 && k % inter_ == 0) || converge <= param->epsilon)
            std::clog << "\rIteration-times: " << k << "\tconvergence: " << converge;
        return 0;
    }

    void silent() { silent_ = true; }
    void set_report_interval(unsigned int inter) { inter_ = inter; }
    void set_clcg_parameter(const clcg_para &in_param) { param_ = in_param; }

    void Minimize(lcg_complex *m, const lcg_complex *b, int x_size, clcg_solver_enum solver_id = CLCG_CGS,
                  bool verbose = true, bool er_throw = false)
    {   // solver.cpp:255-310
        if (silent_) {
            const int ret = clcg_solver(_AxProduct, nullptr, m, b, x_size, &param_, this, solver_id);
            if (ret < 0) lcg_error_str(ret, true);
            return;
        }
        const auto t0 = std::chrono::steady_clock::now();
        const int ret = clcg_solver(_AxProduct, _Progress, m, b, x_size, &param_, this, solver_id);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        static const char *names[] = {"BI-CG", "BI-CG (symmetrically accelerated)", "CGS", "BICGSTAB", "TFQMR"};
        if (!er_throw)
            std::clog << std::endl << "Solver: " << (solver_id >= 0 && solver_id <= 4 ? names[solver_id] : "Unknown")
                      << ". Time cost: " << ms << " ms" << std::endl;
        if (verbose || ret < 0) {
            if (!er_throw) std::clog << lcg_status_text(ret) << std::endl;
            lcg_error_str(ret, er_throw);
        }
    }

    static void _AxProduct(void *instance, const lcg_complex *x, lcg_complex *prod_Ax, const int x_size,
                           lcg_matrix_e layout, clcg_complex_e conjugate)
    {
        static_cast<CLCG_Solver *>(instance)->AxProduct(x, prod_Ax, x_size, layout, conjugate);
    }
    static int _Progress(void *instance, const lcg_complex *m, const lcg_float converge, const clcg_para *param,
                         const int n_size, const int k)
    {
        return static_cast<CLCG_Solver *>(instance)->Progress(m, converge, param, n_size, k);
    }
};

#endif // LCG_SOLVER_CLASSES_HPP
