// TEST-ONLY host harness.  NOT part of the product.
//
// The authoring container has no GPU.  This file implements the Backend
// interface with plain CPU loops over the *same* per-tet / per-row bodies the
// HIP kernels run (tet_ops.h, row_ops.h), so that graph compilation, the
// assembly pattern, the ANM driver and the Pade logic can be debugged against
// the oracle before spending GPU minutes.  It is compiled only into
// tests/hostsim/libsanm_hostsim.so by tests/hostsim/build.py; libsanm_hip.so
// never contains it and sanm_amd never loads it.  GPU parity is proven by the
// `-m gpu` tests, not by this harness.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "backend.h"
#include "graph.h"
#include "row_ops.h"
#include "tet_ops.h"

namespace sanm_hip {
namespace {
class HostSimBackend final : public Backend {
public:
    const char* name() const override { return "hostsim"; }
    void* alloc(size_t bytes) override { return std::malloc(bytes ? bytes : 8); }
    void free(void* p) override { std::free(p); }
    void h2d(void* d, const void* s, size_t b) override { if (b) std::memcpy(d, s, b); }
    void d2h(void* d, const void* s, size_t b) override { if (b) std::memcpy(d, s, b); }
    void d2d(void* d, const void* s, size_t b) override { if (b) std::memmove(d, s, b); }
    void zero(void* d, size_t b) override { if (b) std::memset(d, 0, b); }
    void sync() override {}

    void run_pass(const ProgramDev& P, int mode, int order, const double* xvec) override {
        for (int64_t t = 0; t < P.T; ++t) exec_program_tet(P, mode, order, t, xvec);
    }
    void gather_rows(const SparseRowsDev& R, const double* src, double* dst) override {
        for (int64_t i = 0; i < R.nrows; ++i) dst[i] = gather_row(R, src, i);
    }
    void assemble(const AssemblyDev& A, const double* jac, double* val) override {
        for (int64_t s = 0; s < A.nslots; ++s) val[s] = assemble_slot(A, jac, s);
    }
    void spmv(const CsrDev& A, const double* x, double* y) override {
        for (int64_t i = 0; i < A.n; ++i) y[i] = spmv_row(A, x, i);
    }
    double dot(size_t n, const double* x, const double* y) override {
        double s = 0;
        for (size_t i = 0; i < n; ++i) s += x[i] * y[i];
        return s;
    }
    void axpby(size_t n, double a, const double* x, double b, const double* y,
               double* out) override {
        for (size_t i = 0; i < n; ++i) out[i] = b == 0.0 ? a * x[i] : a * x[i] + b * y[i];
    }
    void vmul(size_t n, const double* x, const double* y, double* out) override {
        for (size_t i = 0; i < n; ++i) out[i] = x[i] * y[i];
    }
    void csr_inv_diag(const CsrDev& A, double scale, double* d) override {
        for (int64_t i = 0; i < A.n; ++i) d[i] = 1.0 / (scale * csr_diag(A, i));
    }
    int64_t count_nonfinite(size_t n, const double* x) override {
        int64_t c = 0;
        for (size_t i = 0; i < n; ++i) c += std::isfinite(x[i]) ? 0 : 1;
        return c;
    }
    double allclose_excess(size_t n, const double* a, const double* b, double eps) override {
        double m = -1e300;
        for (size_t i = 0; i < n; ++i) m = std::fmax(m, allclose_excess1(a[i], b[i], eps));
        return m;
    }
    double t0v_excess(size_t n, const double* fx, const double* v, double t0,
                      double tol) override {
        double m = -1e300;
        for (size_t i = 0; i < n; ++i) {
            double a = fx[i], b = v[i] * t0;
            double me = std::fmax(std::fmin(std::fabs(a), std::fabs(b)), 1.0) * tol;
            double d = std::fabs(a + b);
            m = std::fmax(m, (d == d) ? d - me : 1e300);
        }
        return m;
    }
};
}  // namespace

Backend* make_backend(int) { return new HostSimBackend(); }
}  // namespace sanm_hip
