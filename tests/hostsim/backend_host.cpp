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
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "backend.h"
#include "graph.h"
#include "row_ops.h"
#include "vecprog.h"
#include "tet_ops.h"

namespace sanm_hip {
int hostsim_mf_factor(const MfDev& mf, const MfSchedule& sch, const CsrDev& A);
void hostsim_mf_solve(const MfDev& mf, const MfSchedule& sch, const double* b, double* x);
void hostsim_mf_factor_piece(const MfDev& mf, const MfSchedule& sch, const CsrDev& A, int l0, int l1, bool prologue);
void hostsim_mf_solve_piece(const MfDev& mf, const MfSchedule& sch, bool fwd, int l0, int l1);
LinearSolver* hostsim_make_pardiso(const JacobianPattern& pat, int threads);  // pardiso_solver.cpp
namespace {
// Worker threads over contiguous ranges, like the reference's ParallelTaylorCoeffProp
// (libsanm/symbolic.cpp:525-536: worker w owns [w*T/nr, (w+1)*T/nr)); SANM_CPU_THREADS sets the count
// (default 1: the tests stay serial and deterministic).  Used by the CPU-baseline leg of bench.py.
class RangePool {
    std::vector<std::thread> m_threads;
    std::mutex m_mtx;
    std::condition_variable m_cv_job, m_cv_done;
    std::function<void(int64_t, int64_t)> m_fn;
    int64_t m_n = 0;
    uint64_t m_gen = 0;
    int m_pending = 0;
    bool m_stop = false;
    const int m_nr;

    void worker(int id) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk{m_mtx};
            m_cv_job.wait(lk, [&] { return m_stop || m_gen != seen; });
            if (m_stop) return;
            seen = m_gen;
            const int64_t b = id * m_n / m_nr, e = (id + 1) * m_n / m_nr;
            lk.unlock();
            if (e > b) m_fn(b, e);
            lk.lock();
            if (--m_pending == 0) m_cv_done.notify_one();
        }
    }

public:
    explicit RangePool(int nr) : m_nr{nr} {
        for (int i = 1; i < nr; ++i) m_threads.emplace_back([this, i] { worker(i); });
    }
    ~RangePool() {
        {
            std::lock_guard<std::mutex> lk{m_mtx};
            m_stop = true;
        }
        m_cv_job.notify_all();
        for (auto& t : m_threads) t.join();
    }
    int size() const { return m_nr; }
    void run(int64_t n, std::function<void(int64_t, int64_t)> fn) {
        if (m_nr == 1 || n < 2 * m_nr) {
            fn(0, n);
            return;
        }
        {
            std::lock_guard<std::mutex> lk{m_mtx};
            m_fn = std::move(fn);
            m_n = n;
            m_pending = m_nr - 1;
            ++m_gen;
        }
        m_cv_job.notify_all();
        m_fn(0, n / m_nr);  // the calling thread is worker 0
        std::unique_lock<std::mutex> lk{m_mtx};
        m_cv_done.wait(lk, [&] { return m_pending == 0; });
    }
};

class HostSimBackend final : public Backend {
    RangePool m_pool{std::max(1, std::getenv("SANM_CPU_THREADS") ? std::atoi(std::getenv("SANM_CPU_THREADS")) : 1)};
    struct Bracket {
        std::string tag;
        std::chrono::steady_clock::time_point t0;
    };
    std::vector<Bracket> m_open;
    std::map<std::string, double> m_acc, m_cnt;

public:
    const char* name() const override { return "hostsim"; }
    // vector graphs (vecprog.h): the device kernel's schedule as loops -- every "thread" e runs an operator before
    // the next operator starts
    void run_vec_pass(const VecProgDev& P, int mode, int order, const double* xin) override {
        std::vector<double> g(P.grad_total);
        for (int64_t b = 0; b < P.B; ++b) {
            if (mode == PASS_GRAD) {
                for (int r = 0; r < P.odim; ++r) {
                    std::fill(g.begin(), g.end(), 0.0);
                    g[P.vars[P.out_var].grad + r] = 1.0;
                    for (int i = P.nops - 1; i >= 0; --i)
                        for (int e = 0; e < VEC_MAX_SIZE; ++e) vec_backward(P, P.ops[i], b, e, g.data());
                    for (int e = 0; e < P.idim; ++e)
                        P.arena[P.jac + (b * P.odim + r) * P.idim + e] = g[P.vars[P.in_var].grad + e];
                }
                continue;
            }
            for (int i = 0; i < P.nops; ++i)
                for (int e = 0; e < VEC_MAX_SIZE; ++e) vec_forward(P, P.ops[i], mode, order, b, e, xin);
        }
    }
    LinearSolver* make_external_solver(const JacobianPattern& pat, const HyperParam&) override {
        return hostsim_make_pardiso(pat, m_pool.size());
    }
    // everything runs synchronously here: the brackets are host clocks
    void phase_begin(const char* tag) override { m_open.push_back({tag, std::chrono::steady_clock::now()}); }
    void phase_end() override {
        if (m_open.empty()) return;
        m_acc[m_open.back().tag] +=
                std::chrono::duration<double>(std::chrono::steady_clock::now() - m_open.back().t0).count();
        m_cnt[m_open.back().tag] += 1;
        m_open.pop_back();
    }
    void phase_collect(std::map<std::string, double>& acc, std::map<std::string, double>* cnt) override {
        for (auto& kv : m_acc) acc[kv.first] += kv.second;
        if (cnt)
            for (auto& kv : m_cnt) (*cnt)[kv.first] += kv.second;
        m_acc.clear();
        m_cnt.clear();
    }
    void* alloc(size_t bytes) override { return std::malloc(bytes ? bytes : 8); }
    void free(void* p) override { std::free(p); }
    void h2d(void* d, const void* s, size_t b) override { if (b) std::memcpy(d, s, b); }
    void d2h(void* d, const void* s, size_t b) override { if (b) std::memcpy(d, s, b); }
    void d2d(void* d, const void* s, size_t b) override { if (b) std::memmove(d, s, b); }
    void zero(void* d, size_t b) override { if (b) std::memset(d, 0, b); }
    void sync() override {}

    void run_pass(const ProgramDev& P, int mode, int order, const double* xvec) override {
        m_pool.run(P.T, [&](int64_t tb, int64_t te) {
            std::vector<double> cur(P.cur_size + 1);
            if (mode == PASS_GRAD) {  // one reverse sweep per row of the Jacobian (the row travels in `order`)
                for (int64_t t = tb; t < te; ++t)
                    for (int r = 0; r < P.odim; ++r) exec_program_tet(P, mode, r, t, xvec, cur.data(), 1);
                return;
            }
            for (int64_t t = tb; t < te; ++t) exec_program_tet(P, mode, order, t, xvec, cur.data(), 1);
        });
    }
    void dense_lu_factor(const CsrDev& A, double* lu, int32_t* piv, double* status) override {
        const int64_t n = A.n;
        std::fill(lu, lu + n * n, 0.0);
        for (int64_t i = 0; i < n; ++i)
            for (uint32_t p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p) lu[i * n + A.col[p]] += A.val[p];
        double pmin = std::numeric_limits<double>::infinity(), pmax = 0;
        bool bad = false;
        for (int64_t c = 0; c < n; ++c) {
            // the device kernel's rule (backend_hip.hip dense_lu_kernel): NaN entries never win; a column of NaNs
            // alone is left as it is and reported through status[0] = NaN
            int64_t p = -1;
            double best = -1.0;
            for (int64_t r = c; r < n; ++r)
                if (std::fabs(lu[r * n + c]) > best) {
                    best = std::fabs(lu[r * n + c]);
                    p = r;
                }
            if (p < 0) {
                p = c;
                best = 0.0;
                bad = true;
            }
            piv[c] = (int32_t)p;
            pmin = std::min(pmin, best);
            pmax = std::max(pmax, best);
            if (p != c)
                for (int64_t j = 0; j < n; ++j) std::swap(lu[c * n + j], lu[p * n + j]);
            if (best == 0.0) continue;
            const double pv = lu[c * n + c];
            for (int64_t r = c + 1; r < n; ++r) {
                const double l = lu[r * n + c] / pv;
                lu[r * n + c] = l;
                if (l == 0.0) continue;
                for (int64_t j = c + 1; j < n; ++j) lu[r * n + j] = __builtin_fma(-l, lu[c * n + j], lu[r * n + j]);
            }
        }
        status[0] = bad ? std::numeric_limits<double>::quiet_NaN() : pmin;
        status[1] = pmax;
    }
    void dense_lu_solve(int64_t n, const double* lu, const int32_t* piv, const double* b, double* x) override {
        std::vector<double> y(b, b + n);
        // (the factor swapped whole rows, the finished part of L included: all interchanges first, as getrs does)
        for (int64_t c = 0; c < n; ++c) std::swap(y[c], y[piv[c]]);
        for (int64_t c = 0; c < n; ++c)
            for (int64_t r = c + 1; r < n; ++r) y[r] = __builtin_fma(-lu[r * n + c], y[c], y[r]);
        for (int64_t c = n - 1; c >= 0; --c) {
            y[c] /= lu[c * n + c];
            for (int64_t r = 0; r < c; ++r) y[r] = __builtin_fma(-lu[r * n + c], y[c], y[r]);
        }
        std::copy(y.begin(), y.end(), x);
    }
    void gather_rows(const SparseRowsDev& R, const double* src, double* dst, const int32_t* perm,
                     double* dst2) override {
        for (int64_t i = 0; i < R.nrows; ++i) {
            dst[i] = gather_row(R, src, i);
            if (perm) dst2[perm[i]] = dst[i];
        }
    }
    void assemble(const AssemblyDev& A, const double* jac, double* val, double* grad_t) override {
        // worker-parallel by rows in the reference (anm.cpp:384-390)
        m_pool.run(A.n, [&](int64_t sb, int64_t se) {
            std::vector<int32_t> pos(A.n + 1, -1);
            for (int64_t i = sb; i < se; ++i) assemble_row(A, jac, i, val, grad_t, pos.data());
        });
    }
    void gather(size_t n, const double* src, const uint32_t* idx, double* dst) override {
        for (size_t i = 0; i < n; ++i) dst[i] = src[idx[i]];
    }
    void ata(const CsrDev& At, const CsrDev& M, const uint32_t* mrow, double lambda) override {
        for (int64_t e = 0; e < M.nnz; ++e) M.val[e] = ata_entry(At, mrow[e], M.col[e], lambda);
    }
    void residual(const CsrDev& A, const double* b, const double* x, double* r) override {
        for (int64_t i = 0; i < A.n; ++i) {  // in extended precision, like the device's double-double kernel
            long double s = b[i];
            for (uint32_t p = A.rowptr[i], e = A.rowptr[i + 1]; p < e; ++p) s -= (long double)A.val[p] * x[A.col[p]];
            r[i] = (double)s;
        }
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
    double* alloc_host(size_t n) override { return static_cast<double*>(std::malloc(std::max<size_t>(n, 1) * 8)); }
    void free_host(double* p) override { std::free(p); }
    void dot_async(size_t n, const double* x, const double* y, double* out) override { *out = dot(n, x, y); }
    void multi_dot_async(size_t n, const double* x, int nvec, double* const* ys, double* out,
                         const double* last_norm2, const double* last_nn2, double eps) override {
        if (last_norm2 && nvec > 0) gs_renorm_async(n, ys[nvec - 1], last_norm2, last_nn2, eps);
        multi_dot(n, x, nvec, ys, out);
    }
    void gs_update_async(size_t n, const double* x, int nvec, const double* const* qs, const double* coefs,
                         int first, double* out, double* norm2) override {
        for (size_t i = 0; i < n; ++i) {
            double acc = x[i];
            for (int j = first; j < nvec; ++j) acc += -coefs[j] * qs[j][i];
            out[i] = acc;
        }
        *norm2 = dot(n, out, out);
    }
    void scale_rsqrt_async(size_t n, double* v, const double* norm2, double eps, double* nn2) override {
        const double s = 1.0 / std::max(std::sqrt(*norm2), eps);
        for (size_t i = 0; i < n; ++i) v[i] *= s;
        *nn2 = dot(n, v, v);
    }
    void gs_renorm_async(size_t n, double* v, const double* norm2, const double* nn2, double eps) override {
        if (std::sqrt(*norm2) >= eps) return;
        const double s = 1.0 / std::sqrt(*nn2);
        for (size_t i = 0; i < n; ++i) v[i] *= s;
    }
    void next_coeff_async(const NextCoeff& nc) override {
        const double scale = nc.sc ? 1.0 / (nc.sc[0] - nc.sc[1]) : nc.scale;
        const double t = *nc.num * scale;
        axpby_tail(nc.n, -t, nc.xg, -1.0, nc.xb, nc.out, t);
        *nc.t_out = t;
    }
    void sanity_check_async(const CsrDev& A, const double* xi, const double* grad_t, const double* bi,
                            double eps, size_t n1, const double* x1, double* tmp0, double* tmp1,
                            double* out2) override {
        sanity_check(A, xi, xi[A.n], grad_t, bi, eps, n1, x1, tmp0, tmp1, out2);
    }
    void axpby_tail(size_t n, double a, const double* x, double b, const double* y, double* out,
                    double tail) override {
        axpby(n, a, x, b, y, out);
        out[n] = tail;
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
    int mf_factor(const MfDev& mf, const MfSchedule& sch, const CsrDev& A) override {
        return hostsim_mf_factor(mf, sch, A);
    }
    void mf_solve_fused(const MfDev& mf, const MfSchedule& sch, const double* b, double* x, const double* dot_y,
                        double* dot_out) override {
        hostsim_mf_solve(mf, sch, b, x);
        if (dot_y) *dot_out = dot(mf.n, x, dot_y);
    }
    void mf_solve(const MfDev& mf, const MfSchedule& sch, const double* b, double* x) override {
        hostsim_mf_solve(mf, sch, b, x);
    }
    void mf_factor_piece(const MfDev& mf, const MfSchedule& sch, const CsrDev& A, int l0, int l1, bool prologue) override {
        hostsim_mf_factor_piece(mf, sch, A, l0, l1, prologue);
    }
    void mf_factor_status(const MfDev& mf, double* out) override { *out = (double)*mf.status; }
    void mf_solve_piece(const MfDev& mf, const MfSchedule& sch, bool fwd, int l0, int l1) override {
        hostsim_mf_solve_piece(mf, sch, fwd, l0, l1);
    }
    void mf_permute(const MfDev& mf, const double* b, double* x) override {
        if (b)
            for (int64_t i = 0; i < mf.n; ++i) mf.work[mf.perm[i]] = b[i];
        if (x)
            for (int64_t i = 0; i < mf.n; ++i) x[i] = mf.work[mf.perm[i]];
    }
    void copy2d_batch(const MfCopy2D* d, int count, int, int, const double* src, double* dst) override {
        for (int q = 0; q < count; ++q)
            for (int i = 0; i < d[q].rows; ++i)
                for (int j = 0; j < d[q].cols; ++j)
                    dst[d[q].dst + (int64_t)i * d[q].ldd + j] = src[d[q].src + (int64_t)i * d[q].lds + j];
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

// ---- multifrontal: straightforward serial reference of the numeric phases ----
// (same data structures as the HIP kernels; validates the host symbolic analysis)
namespace sanm_hip {
namespace {
struct HostMf {
    static int factor(const MfDev& mf, const MfSchedule& sch, const CsrDev& A) {
        factor_piece(mf, sch, A, 0, (int)sch.levels.size(), true);
        return *mf.status;
    }
    static void factor_piece(const MfDev& mf, const MfSchedule& sch, const CsrDev& A, int l0, int l1, bool prologue) {
        if (prologue) {
            if (sch.dist.enabled) {
                // only the fronts this rank factors are zeroed (MfSchedule::Dist::own_store); the harness POISONS the
                // rest, so that a read of another rank's front that no exchange delivered shows up as NaN
                for (int64_t q = 0; q < mf.front_store_size; ++q) mf.front_store[q] = std::nan("");
                for (const auto& r : sch.dist.own_store)
                    std::memset(mf.front_store + r.first, 0, (size_t)(r.second - r.first) * sizeof(double));
            } else {
                std::memset(mf.front_store, 0, mf.front_store_size * sizeof(double));
            }
            if (!sch.a_dst_ready) {  // (as the HIP backend: where the entries of A go, on the first factorisation)
                for (int64_t i = 0; i < mf.n; ++i)
                    for (uint32_t p = A.rowptr[i]; p < A.rowptr[i + 1]; ++p)
                        mf.a_dst[p] = mf_scatter_slot(mf.fronts, mf.own_front, mf.bnd_idx, mf.perm[i], mf.perm[A.col[p]], mf.front_here);
                sch.a_dst_ready = true;
            }
            for (int64_t p = 0; p < mf.nnzA; ++p)
                if (mf.a_dst[p] >= 0) mf.front_store[mf.a_dst[p]] += A.val[p];  // (< 0: an entry of another rank's front)
            double amax = 0;
            for (int64_t p = 0; p < mf.nnzA; ++p) amax = std::fmax(amax, std::fabs(A.val[p]));
            *mf.piv_amax = amax;
            *mf.status = 0;
        }
        int bad = 0;
        const double thr = MF_PIVOT_EPS * *mf.piv_amax;  // static pivot perturbation, as in the HIP kernels (mf_kernels.h)
        for (int li = l0; li < l1; ++li) {
            const auto& L = sch.levels[li];
            // extend-add the children of this level's fronts
            for (auto [b, e] : L.ea_rounds)
                for (int32_t q = b; q < e; ++q) {
                    const MfFrontDev& c = mf.fronts[sch.ea_children[q]];
                    const MfFrontDev& p = mf.fronts[c.parent];
                    const int32_t* rel = mf.rel + c.rel_off;
                    int nb = c.m - c.k;
                    for (int i = 0; i < nb; ++i)
                        for (int j = 0; j < nb; ++j)
                            mf.front_store[p.off + (int64_t)rel[i] * p.ld + rel[j]] +=
                                    mf.front_store[c.off + (int64_t)(2 * c.k + i) * c.ld + 2 * c.k + j];
                }
            for (int32_t q = L.front_begin; q < L.front_end; ++q) {
                const MfFrontDev& f = mf.fronts[mf.level_fronts[q]];
                double* F = mf.front_store + f.off;
                // physical position of logical index i: [pivot | augmentation (unused here) | boundary]
                const int m = f.m, k = f.k;
                const int64_t ld = f.ld;
                auto P = [k](int i) { return i < k ? i : i + k; };
                for (int j = 0; j < k; ++j) {
                    double piv = F[P(j) * ld + P(j)];
                    if (!(std::fabs(piv) > thr)) {
                        ++bad;
                        piv = thr > 0 ? std::copysign(thr, piv) : 1.0;
                        F[P(j) * ld + P(j)] = piv;
                    }
                    double inv = 1.0 / piv;
                    for (int i = j + 1; i < m; ++i) {
                        double l = F[P(i) * ld + P(j)] * inv;
                        F[P(i) * ld + P(j)] = l;
                        if (l != 0)
                            for (int c2 = j + 1; c2 < m; ++c2) F[P(i) * ld + P(c2)] -= l * F[P(j) * ld + P(c2)];
                    }
                }
            }
        }
        *mf.status += bad;
    }

    static void solve(const MfDev& mf, const MfSchedule& sch, const double* b, double* x) {
        double* w = mf.work;
        if (b)  // (nullptr: the caller already put the permuted right-hand side into mf.work)
            for (int64_t i = 0; i < mf.n; ++i) w[mf.perm[i]] = b[i];
        solve_piece(mf, sch, true, 0, (int)sch.levels.size());
        solve_piece(mf, sch, false, 0, (int)sch.levels.size());
        for (int64_t i = 0; i < mf.n; ++i) x[i] = w[mf.perm[i]];
    }
    static void solve_piece(const MfDev& mf, const MfSchedule& sch, bool fwd, int l0, int l1) {
        double* w = mf.work;
        std::vector<double> t;
        if (fwd) {
            for (int li = l0; li < l1; ++li) {
                const auto& L = sch.levels[li];
                for (int32_t q = L.front_begin; q < L.front_end; ++q) {
                    const MfFrontDev& f = mf.lfronts[q];
                    const double* F = mf.front_store + f.off;
                    const int m = f.m, k = f.k;
                    t.assign(m, 0.0);
                    for (int r = 0; r < k; ++r) t[r] = w[f.own_start + r];
                    // children's update entries arrive through the inbox (one slot per child and row)
                    for (int j = 0; j < f.nch; ++j)
                        for (int r = 0; r < m; ++r) t[r] += mf.inbox_store[f.inbox_off + (int64_t)j * m + r];
                    for (int r = 0; r < k; ++r) {  // unit lower L11
                        double v = t[r];
                        for (int c2 = 0; c2 < r; ++c2) v -= F[(int64_t)r * f.ld + c2] * t[c2];
                        t[r] = v;
                    }
                    for (int r = k; r < m; ++r) {
                        double v = t[r];
                        for (int c2 = 0; c2 < k; ++c2) v -= F[(int64_t)(r + k) * f.ld + c2] * t[c2];
                        mf.inbox_store[mf.upd_dst[f.bnd_off + r - k]] = v;
                    }
                    for (int r = 0; r < k; ++r) w[f.own_start + r] = t[r];
                }
            }
            return;
        }
        for (int li = l1 - 1; li >= l0; --li) {  // backward
            const auto& L = sch.levels[li];
            for (int32_t q = L.front_begin; q < L.front_end; ++q) {
                const MfFrontDev& f = mf.fronts[mf.level_fronts[q]];
                const double* F = mf.front_store + f.off;
                const int m = f.m, k = f.k;
                const int32_t* bi = mf.bnd_idx + f.bnd_off;
                for (int r = k - 1; r >= 0; --r) {
                    double v = w[f.own_start + r];
                    for (int c2 = k; c2 < m; ++c2) v -= F[(int64_t)r * f.ld + c2 + k] * w[bi[c2 - k]];
                    for (int c2 = r + 1; c2 < k; ++c2) v -= F[(int64_t)r * f.ld + c2] * w[f.own_start + c2];
                    w[f.own_start + r] = v / F[(int64_t)r * f.ld + r];
                }
            }
        }
    }
};
}  // namespace
int hostsim_mf_factor(const MfDev& mf, const MfSchedule& sch, const CsrDev& A) { return HostMf::factor(mf, sch, A); }
void hostsim_mf_solve(const MfDev& mf, const MfSchedule& sch, const double* b, double* x) { HostMf::solve(mf, sch, b, x); }
void hostsim_mf_factor_piece(const MfDev& mf, const MfSchedule& sch, const CsrDev& A, int l0, int l1, bool prologue) {
    HostMf::factor_piece(mf, sch, A, l0, l1, prologue);
}
void hostsim_mf_solve_piece(const MfDev& mf, const MfSchedule& sch, bool fwd, int l0, int l1) {
    HostMf::solve_piece(mf, sch, fwd, l0, l1);
}
}  // namespace sanm_hip

// the run-time compiler belongs to the product library only (sanm_amd/csrc/rtc.cpp)
extern "C" int sanm_rtc_compile_check(const char*, char*, size_t, size_t*) { return 2; }
extern "C" int sanm_rtc_cache_stats(int64_t* compiled, int64_t* memory_hits, int64_t* disk_hits) {
    if (compiled) *compiled = 0;
    if (memory_hits) *memory_hits = 0;
    if (disk_hits) *disk_hits = 0;
    return 0;
}
extern "C" int sanm_rtc_cache_probe(const char*) { return 2; }
extern "C" int sanm_rtc_cache_drop_memory(void) { return 0; }
extern "C" int sanm_rtc_source_key(const char*, char* key33) { key33[0] = 0; return 2; }
extern "C" int sanm_rtc_compile_to_file(const char*, const char*, char*, size_t) { return 2; }
extern "C" int sanm_rtc_embedded_hits(int64_t* hits) { *hits = 0; return 0; }
