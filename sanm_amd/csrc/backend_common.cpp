#include <cmath>

#include "backend.h"
#include "graph.h"

namespace sanm_hip {

void Backend::comm_unique_id(void*) { sanm_throw(SANM_ERR_UNSUPPORTED, "backend %s has no native collective", name()); }
void Backend::comm_init(int, int, const void*) {
    sanm_throw(SANM_ERR_UNSUPPORTED, "backend %s has no native collective", name());
}
void Backend::allreduce_sum(double*, int64_t) {
    sanm_throw(SANM_ERR_UNSUPPORTED, "backend %s has no native collective (pass an all-reduce callback)", name());
}

void Backend::comm_exchange(double*, const MfSchedule::Xfer*, int) {
    sanm_throw(SANM_ERR_UNSUPPORTED, "backend %s has no native point-to-point transfers", name());
}

void Backend::residual(const CsrDev& A, const double* b, const double* x, double* r) {
    double* ax = static_cast<double*>(alloc(A.n * 8));
    spmv(A, x, ax);
    axpby(A.n, 1.0, b, -1.0, ax, r);
    free(ax);
}

void Backend::run_gs_phase(const GsPhase& ph) {
    switch (ph.kind) {
        case 1:
            multi_dot_async(ph.n, ph.x, ph.nvec, ph.vecs.data(), ph.red_out, ph.norm2, ph.nn2, ph.eps);
            break;
        case 2:
            gs_update_async(ph.n, ph.x, ph.nvec, ph.vecs.data(), ph.coefs, ph.first, ph.out, ph.red_out);
            break;
        case 3:
            scale_rsqrt_async(ph.n, ph.out, ph.norm2, ph.eps, ph.red_out);
            break;
        default:
            sanm_throw(SANM_ERR_ASSERT, "bad Gram-Schmidt phase %d", ph.kind);
    }
}

void Backend::mf_factor_piece(const MfDev&, const MfSchedule&, const CsrDev&, int, int, bool) {
    sanm_throw(SANM_ERR_UNSUPPORTED, "backend %s: no piecewise multifrontal factorisation", name());
}
void Backend::mf_factor_status(const MfDev&, double*) {
    sanm_throw(SANM_ERR_UNSUPPORTED, "backend %s: no piecewise multifrontal factorisation", name());
}
void Backend::mf_solve_piece(const MfDev&, const MfSchedule&, bool, int, int) {
    sanm_throw(SANM_ERR_UNSUPPORTED, "backend %s: no piecewise multifrontal solve", name());
}
void Backend::mf_permute(const MfDev&, const double*, double*) {
    sanm_throw(SANM_ERR_UNSUPPORTED, "backend %s: no piecewise multifrontal solve", name());
}
void Backend::copy2d_batch(const MfCopy2D*, int, int, int, const double*, double*) {
    sanm_throw(SANM_ERR_UNSUPPORTED, "backend %s: no batched block copy", name());
}
void Backend::mf_solve_fused(const MfDev& mf, const MfSchedule& sch, const double* b, double* x, const double* dot_y,
                             double* dot_out) {
    if (!b) sanm_throw(SANM_ERR_ASSERT, "mf_solve_fused: this backend needs the right-hand side");
    mf_solve(mf, sch, b, x);
    if (dot_y) dot_async(mf.n, x, dot_y, dot_out);
}

void Backend::lincomb(size_t n, int nvec, const double* const* ptrs, const double* coefs,
                      double* out) {
    if (nvec == 0) {
        zero(out, n * 8);
        return;
    }
    axpby(n, coefs[0], ptrs[0], 0, nullptr, out);
    for (int j = 1; j < nvec; ++j) axpby(n, 1.0, out, coefs[j], ptrs[j], out);
}

void Backend::multi_dot(size_t n, const double* x, int nvec, const double* const* ys,
                        double* out_host) {
    for (int j = 0; j < nvec; ++j) out_host[j] = dot(n, x, ys[j]);
}

void Backend::lincomb2_diff_norms(size_t n, int nvec, const double* const* ptrs, const double* c1,
                                  const double* c2, double scale, double out_host[2]) {
    double* u = static_cast<double*>(alloc(n * 8));
    double* w = static_cast<double*>(alloc(n * 8));
    lincomb(n, nvec, ptrs, c1, u);
    lincomb(n, nvec, ptrs, c2, w);
    axpby(n, scale, w, -1.0, u, w);
    out_host[0] = dot(n, w, w);
    out_host[1] = dot(n, u, u);
    free(u);
    free(w);
}

void Backend::lincomb2_diff_norms_multi(size_t n, int nvec, const double* const* ptrs, int ncand,
                                        const double* c1, const double* c2, const double* scale,
                                        double* out_host) {
    for (int c = 0; c < ncand; ++c)
        lincomb2_diff_norms(n, nvec, ptrs, c1 + (size_t)c * nvec, c2 + (size_t)c * nvec, scale[c], out_host + 2 * c);
}

void Backend::pcg(const CsrDev& A, double sign, const double* dinv, const double* b, double* x,
                  double rtol, int maxit, int* iters, double* relres) {
    const size_t n = A.n;
    double* r = static_cast<double*>(alloc(n * 8));
    double* z = static_cast<double*>(alloc(n * 8));
    double* p = static_cast<double*>(alloc(n * 8));
    double* q = static_cast<double*>(alloc(n * 8));
    zero(x, n * 8);
    axpby(n, sign, b, 0, nullptr, r);  // r = sign*b - M*0
    double bnorm = std::sqrt(dot(n, r, r));
    int it = 0;
    double rr = bnorm * bnorm;
    if (bnorm > 0) {
        vmul(n, dinv, r, z);
        d2d(p, z, n * 8);
        double rz = dot(n, r, z);
        for (it = 1; it <= maxit; ++it) {
            spmv(A, p, q);
            double pq = sign * dot(n, p, q);
            if (!(pq > 0)) {
                it = -it;
                break;
            }
            double alpha = rz / pq;
            axpby(n, 1.0, x, alpha, p, x);
            axpby(n, 1.0, r, -alpha * sign, q, r);
            rr = dot(n, r, r);
            if (std::sqrt(rr) <= rtol * bnorm) break;
            vmul(n, dinv, r, z);
            double rz_new = dot(n, r, z);
            axpby(n, 1.0, z, rz_new / rz, p, p);
            rz = rz_new;
        }
    }
    *iters = it;
    *relres = bnorm > 0 ? std::sqrt(rr) / bnorm : 0.0;
    free(r);
    free(z);
    free(p);
    free(q);
}

}  // namespace sanm_hip
