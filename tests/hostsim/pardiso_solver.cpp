// CPU BASELINE / TEST-ONLY.  NOT part of the product (see backend_host.cpp).
//
// MKL PARDISO as the LinearSolver of the host harness: the solver the reference factorises its Jacobian with,
// with the reference's settings (libsanm/sparse_solver.cpp:107-127: pardisoinit for mtype 11, iparm[17] =
// iparm[18] = 0, zero-based indexing iparm[34] = 1, the parallel nested dissection iparm[1] = 3 when it runs
// threaded) and the reference's call pattern: phase 12 (analysis + numerical factorisation) at every prepare
// (:336), phase 33 per solve (:154-180), phase -1 on destruction.  The image ships MKL 2021.4 under
// /opt/conda/lib without headers; the two prototypes below are MKL's documented C interface and the
// library is bound at run time with dlopen, so nothing links against it.
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "anm.h"

namespace sanm_hip {
namespace {
using mkl_int = int32_t;  // LP64 interface (libsanm/CMakeLists.txt:41-43)
using pardisoinit_fn = void (*)(void* pt, const mkl_int* mtype, mkl_int* iparm);
using pardiso_fn = void (*)(void* pt, const mkl_int* maxfct, const mkl_int* mnum, const mkl_int* mtype,
                            const mkl_int* phase, const mkl_int* n, const void* a, const mkl_int* ia,
                            const mkl_int* ja, mkl_int* perm, const mkl_int* nrhs, mkl_int* iparm,
                            const mkl_int* msglvl, void* b, void* x, mkl_int* error);
using set_threads_fn = void (*)(int);

struct Mkl {
    pardisoinit_fn init = nullptr;
    pardiso_fn run = nullptr;
    set_threads_fn set_threads = nullptr;
    static const Mkl& get() {
        static Mkl m = [] {
            Mkl r;
            const char* paths[] = {"/opt/conda/lib/libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.1", "libmkl_rt.so",
                                   "libmkl_rt.so.1", "libmkl_rt.so.2"};
            for (const char* p : paths) {
                void* h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
                if (!h) continue;
                r.init = reinterpret_cast<pardisoinit_fn>(dlsym(h, "pardisoinit"));
                r.run = reinterpret_cast<pardiso_fn>(dlsym(h, "pardiso"));
                r.set_threads = reinterpret_cast<set_threads_fn>(dlsym(h, "MKL_Set_Num_Threads"));
                if (r.init && r.run) break;
                r = Mkl{};
            }
            return r;
        }();
        return m;
    }
};

class PardisoSolver final : public LinearSolver {
    const JacobianPattern& m_pat;
    const int m_threads;
    std::vector<mkl_int> m_ia, m_ja;
    void* m_pt[64];
    mkl_int m_iparm[64];
    mkl_int m_mtype = 11;
    bool m_factored = false;

    void call(mkl_int phase, const double* b, double* x) {
        mkl_int maxfct = 1, mnum = 1, n = (mkl_int)m_pat.n(), nrhs = 1, msglvl = 0, error = 0;
        Mkl::get().run(m_pt, &maxfct, &mnum, &m_mtype, &phase, &n, m_pat.csr().val, m_ia.data(), m_ja.data(), nullptr,
                       &nrhs, m_iparm, &msglvl, const_cast<double*>(b), x, &error);
        if (error != 0) sanm_throw(SANM_ERR_NUMERICAL, "pardiso phase=%d failed: error=%d", (int)phase, (int)error);
    }

public:
    PardisoSolver(const JacobianPattern& pat, int threads) : m_pat{pat}, m_threads{threads} {
        const Mkl& mkl = Mkl::get();
        if (!mkl.run) sanm_throw(SANM_ERR_UNSUPPORTED, "MKL (libmkl_rt.so) not found: no PARDISO for the CPU baseline");
        m_ia.assign(pat.h_rowptr().begin(), pat.h_rowptr().end());
        m_ja.assign(pat.h_col().begin(), pat.h_col().end());
        for (int64_t i = 0; i < pat.n(); ++i)
            for (mkl_int p = m_ia[i] + 1; p < m_ia[i + 1]; ++p)
                sanm_check(m_ja[p - 1] < m_ja[p], "PARDISO needs ascending column indices in every row");
        if (mkl.set_threads) mkl.set_threads(threads);
        std::memset(m_pt, 0, sizeof(m_pt));
        std::memset(m_iparm, 0, sizeof(m_iparm));
        mkl.init(m_pt, &m_mtype, m_iparm);
        m_iparm[17] = 0;
        m_iparm[18] = 0;
        m_iparm[34] = 1;
        if (threads > 1) m_iparm[1] = 3;
    }
    ~PardisoSolver() override {
        if (m_factored) {
            try {
                call(-1, nullptr, nullptr);
            } catch (...) {
            }
        }
    }
    void prepare() override {
        call(12, nullptr, nullptr);
        m_factored = true;
    }
    void solve(const double* b, double* x) override {
        for (int64_t i = 0; i < m_pat.n(); ++i)  // sparse_solver.cpp:160-161
            if (!std::isfinite(b[i])) sanm_throw(SANM_ERR_NUMERICAL, "b[%ld]=%g", (long)i, b[i]);
        if (b == x) {
            std::vector<double> tmp(b, b + m_pat.n());
            call(33, tmp.data(), x);
        } else {
            call(33, b, x);
        }
        ++nr_solve;
    }
};
}  // namespace

LinearSolver* hostsim_make_pardiso(const JacobianPattern& pat, int threads) { return new PardisoSolver(pat, threads); }
}  // namespace sanm_hip
