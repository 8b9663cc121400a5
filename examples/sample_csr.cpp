// sample_csr.cpp -- the workload of liblcg's sample8.cu (read data/case_10K_A, convert COO -> CSR,
// solve with CG, CGS and PCG, report the error against data/case_10K_B; sample8.cu:133-279),
// written against liblcg's own entry points as re-exported by include/lcg_dropin.hpp.
// Plain C++: no HIP headers, no vendor handles -- compile with g++ and link liblcg_hip.so.
//
//   g++ -O2 -std=c++11 -Iinclude examples/sample_csr.cpp -Lliblcg_amd/lib -llcg_hip
//       -Wl,-rpath,$PWD/liblcg_amd/lib -o sample_csr && ./sample_csr tests/golden
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <vector>

#include "lcg_dropin.hpp"

static bool read_system(const std::string &path, int &n, std::vector<int> &row, std::vector<int> &col,
                        std::vector<double> &val, std::vector<double> &b)
{   // data/README:1-10
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    int nz = 0;
    in.read((char *)&n, sizeof(int)); in.read((char *)&nz, sizeof(int));
    row.resize(nz); col.resize(nz); val.resize(nz); b.resize(n);
    for (int i = 0; i < nz; i++) {
        in.read((char *)&row[i], sizeof(int)); in.read((char *)&col[i], sizeof(int)); in.read((char *)&val[i], sizeof(double));
    }
    in.read((char *)b.data(), sizeof(double) * n);
    return (bool)in;
}

static double avg_error(const std::vector<double> &a, const std::vector<double> &b)
{   // sample8.cu:66-74
    double s = 0.0;
    for (size_t i = 0; i < a.size(); i++) s += (a[i] - b[i]) * (a[i] - b[i]);
    return std::sqrt(s) / a.size();
}

static int progress(void *, const lcg_float *, const lcg_float converge, const lcg_para *param, const int, const int k)
{   // sample8.cu:122-129
    if (converge <= param->epsilon) std::clog << "Iteration-times: " << k << "\tconvergence: " << converge << std::endl;
    return 0;
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "tests/golden";
    int n = 0, n2 = 0;
    std::vector<int> row, col;
    std::vector<double> val, b, ans;
    if (!read_system(dir + "/case_10K_A", n, row, col, val, b)) { std::cerr << "cannot read " << dir << "/case_10K_A\n"; return 2; }
    {
        std::ifstream in(dir + "/case_10K_B", std::ios::binary);
        in.read((char *)&n2, sizeof(int)); ans.resize(n2); in.read((char *)ans.data(), sizeof(double) * n2);
    }
    lcg_hip_csr_t A = nullptr;
    int rc = lcg_hip_csr_from_coo(&A, n, (int64_t)val.size(), row.data(), col.data(), val.data(), 0, LCG_HIP_MEM_HOST);
    if (rc) { std::cerr << "csr_from_coo: " << lcg_hip_last_error() << "\n"; return 3; }
    lcg_hip_csr_build_jacobi(A, nullptr);

    lcg_para para = lcg_default_parameters();
    para.epsilon = 1e-10; para.abs_diff = 1;
    std::vector<double> m(n);
    int bad = 0;
    struct { const char *name; lcg_solver_enum id; } runs[] = {{"CG", LCG_CG}, {"CGS", LCG_CGS}, {"BICGSTAB", LCG_BICGSTAB}, {"PCG", LCG_PCG}};
    for (auto &r : runs) {
        std::fill(m.begin(), m.end(), 0.0);
        int ret = r.id == LCG_PCG
                      ? lcg_solver_preconditioned(lcg_hip_csr_ax, lcg_hip_jacobi_mx, progress, m.data(), b.data(), n, &para, A)
                      : lcg_solver(lcg_hip_csr_ax, progress, m.data(), b.data(), n, &para, A, r.id);
        const double err = avg_error(m, ans);
        std::printf("%s: ret=%d (%s) iterations=%d averaged_error=%.3e\n", r.name, ret, lcg_status_text(ret),
                    lcg_hip_last_iterations(), err);
        if (ret != 0 || !(err < 1e-6)) bad++;
    }
    lcg_hip_csr_destroy(A);
    return bad ? 1 : 0;
}
