// sample_class.cpp -- liblcg's class-style usage (sample2.cpp / sample4.cpp: derive from LCG_Solver /
// CLCG_Solver, override AxProduct, call Minimize) on the bundled case_10K_A and case_1K_cA systems.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <vector>

#include "lcg_solver_classes.hpp"

template <class T>
static bool read_system(const std::string &path, int &n, std::vector<int> &row, std::vector<int> &col,
                        std::vector<T> &val, std::vector<T> &b)
{   // data/README:1-10
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    int nz = 0;
    in.read((char *)&n, sizeof(int)); in.read((char *)&nz, sizeof(int));
    row.resize(nz); col.resize(nz); val.resize(nz); b.resize(n);
    for (int i = 0; i < nz; i++) {
        in.read((char *)&row[i], sizeof(int)); in.read((char *)&col[i], sizeof(int)); in.read((char *)&val[i], sizeof(T));
    }
    in.read((char *)b.data(), sizeof(T) * n);
    return (bool)in;
}
template <class T> static std::vector<T> read_answer(const std::string &path)
{
    std::ifstream in(path, std::ios::binary);
    int n = 0; in.read((char *)&n, sizeof(int));
    std::vector<T> x(n); in.read((char *)x.data(), sizeof(T) * n);
    return x;
}

class RealDemo : public LCG_Solver {
public:
    lcg_hip_csr_t A = nullptr;
    void AxProduct(const lcg_float *x, lcg_float *Ax, const int) override { lcg_hip_spmv(A, x, Ax); }
    void MxProduct(const lcg_float *x, lcg_float *Mx, const int n) override { lcg_hip_jacobi_mx(A, x, Mx, n); }
};

class ComplexDemo : public CLCG_Solver {
public:
    lcg_hip_csr_t A = nullptr;
    void AxProduct(const lcg_complex *x, lcg_complex *Ax, const int n, lcg_matrix_e layout, clcg_complex_e conj) override
    {
        clcg_csr_ax(A, x, Ax, n, layout, conj);
    }
};

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "tests/golden";
    int bad = 0;
    {
        int n; std::vector<int> row, col; std::vector<double> val, b;
        if (!read_system(dir + "/case_10K_A", n, row, col, val, b)) return 2;
        std::vector<double> ans = read_answer<double>(dir + "/case_10K_B"), m(n, 0.0);
        RealDemo s;
        if (lcg_hip_csr_from_coo(&s.A, n, (int64_t)val.size(), row.data(), col.data(), val.data(), 0, LCG_HIP_MEM_HOST)) return 3;
        lcg_hip_csr_build_jacobi(s.A, nullptr);
        lcg_para p = lcg_default_parameters(); p.epsilon = 1e-10; p.abs_diff = 1;
        s.set_lcg_parameter(p);
        s.set_report_interval(50);
        s.Minimize(m.data(), b.data(), n, LCG_CG);
        double e = 0; for (int i = 0; i < n; i++) e += (m[i] - ans[i]) * (m[i] - ans[i]);
        std::printf("class CG: iterations=%d error=%.3e\n", lcg_hip_last_iterations(), std::sqrt(e));
        if (!(std::sqrt(e) < 1e-4)) bad++;
        std::fill(m.begin(), m.end(), 0.0);
        s.silent();
        s.MinimizePreconditioned(m.data(), b.data(), n);
        e = 0; for (int i = 0; i < n; i++) e += (m[i] - ans[i]) * (m[i] - ans[i]);
        std::printf("class PCG (silent): iterations=%d error=%.3e\n", lcg_hip_last_iterations(), std::sqrt(e));
        if (!(std::sqrt(e) < 1e-4)) bad++;
        // sample2.cpp:143-149: the box-constrained solvers through the class (box [-5, 8] cuts the known answer)
        std::vector<double> low(n, -5.0), hig(n, 8.0);
        p.max_iterations = 40;
        s.set_lcg_parameter(p);
        for (lcg_solver_enum id : {LCG_PG, LCG_SPG}) {
            std::fill(m.begin(), m.end(), 0.0);
            // silent mode + a negative code (here: 40 iterations reached) throws, exactly as solver.cpp:174-178 does
            try { s.MinimizeConstrained(m.data(), b.data(), low.data(), hig.data(), n, id, false); bad++; }
            catch (const std::runtime_error &e) { if (std::string(e.what()) != "Reached the maximal iteration times.") bad++; }
            double mn = m[0], mx = m[0];
            for (double v : m) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
            std::printf("class %s: iterations=%d residual=%.10e min=%g max=%g\n", id == LCG_PG ? "PG" : "SPG", lcg_hip_last_iterations(),
                        lcg_hip_last_residual(), mn, mx);
            if (lcg_hip_last_iterations() != 40 || mn < -5.0 || mx > 8.0) bad++;
        }
        lcg_hip_csr_destroy(s.A);
    }
    {
        int n; std::vector<int> row, col; std::vector<lcg_complex> val, b;
        if (!read_system(dir + "/case_1K_cA", n, row, col, val, b)) return 2;
        std::vector<lcg_complex> ans = read_answer<lcg_complex>(dir + "/case_1K_cB"), m(n, lcg_complex(0, 0));
        ComplexDemo s;
        if (lcg_hip_csr_from_coo(&s.A, n, (int64_t)val.size(), row.data(), col.data(), (const double *)val.data(), 1, LCG_HIP_MEM_HOST)) return 3;
        clcg_para p = clcg_default_parameters(); p.epsilon = 1e-10; p.abs_diff = 1;
        s.set_clcg_parameter(p);
        s.set_report_interval(0);
        s.Minimize(m.data(), b.data(), n, CLCG_TFQMR);
        double e = 0; for (int i = 0; i < n; i++) e += std::norm(m[i] - ans[i]);
        std::printf("class TFQMR: iterations=%d error=%.3e\n", lcg_hip_last_iterations(), std::sqrt(e));
        if (!(std::sqrt(e) < 2e-3)) bad++;
        lcg_hip_csr_destroy(s.A);
    }
    return bad ? 1 : 0;
}
