#include "anm.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <exception>
#include <functional>
#include <cstddef>
#include <future>
#include <cstdlib>
#include <cstring>
#include <limits>

#include "multifrontal.h"
#include "host_parallel.h"
#include "vecprog_host.h"
#include "poly.h"
#include "tet_ops.h"

namespace sanm_hip {
namespace {
std::string ssprintf(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
std::string ssprintf(const char* fmt, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    const int k = std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return std::string(buf, buf + std::min<int>(std::max(k, 0), (int)sizeof buf - 1));
}
}  // namespace


// ------------------------------------------------------------ profiling --
class AnmDriver::ScopedTimer {
    AnmDriver* m_d;
    const char* m_tag;
    std::chrono::steady_clock::time_point m_t0;
    int64_t m_l0 = 0;

public:
    // mode 1: host clock around a synchronised region (the ScopedProfiler of the reference, utils.h:225-249);
    // mode 2: device events, no synchronisation (the launch mix of the timed run is undisturbed)
    ScopedTimer(AnmDriver* d, const char* tag) : m_d{d}, m_tag{tag} {
        if (m_d->m_profile_mode) m_l0 = m_d->m_be->launch_count();
        if (m_d->m_profile_mode == 1) {
            m_d->m_be->sync();
            m_t0 = std::chrono::steady_clock::now();
        } else if (m_d->m_profile_mode == 2) {
            m_d->m_be->phase_begin(tag);
        }
    }
    ~ScopedTimer() {
        if (m_d->m_profile_mode) m_d->m_profile_launches[m_tag] += double(m_d->m_be->launch_count() - m_l0);
        if (m_d->m_profile_mode == 1) {
            m_d->m_be->sync();
            m_d->m_profile[m_tag] +=
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - m_t0).count();
            m_d->m_profile_cnt[m_tag] += 1;
        } else if (m_d->m_profile_mode == 2) {
            m_d->m_be->phase_end();
        }
    }
};

// ---- host-side trace of the step's tail (SANM_TAIL_TRACE=1; scripts/tail_trace.py) ----------------------------
// mark(label): the host clock at a point of the step; at exit the mean interval between consecutive marks is printed.
// What it is for: the device sits idle between the probe kernels of the Pade range estimate and the restart (the
// kernel trace shows the gaps, not what the host does in them).
namespace {
struct HostTrace {
    const bool on = std::getenv("SANM_TAIL_TRACE") != nullptr;
    std::chrono::steady_clock::time_point last;
    const char* last_label = nullptr;
    std::vector<std::pair<std::string, std::pair<double, int>>> acc;
    void mark(const char* label) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        if (last_label) {
            const std::string key = std::string(last_label) + " -> " + label;
            const double us = std::chrono::duration<double, std::micro>(now - last).count();
            auto it = std::find_if(acc.begin(), acc.end(), [&](const auto& e) { return e.first == key; });
            if (it == acc.end()) acc.push_back({key, {us, 1}});
            else {
                it->second.first += us;
                it->second.second += 1;
            }
        }
        last = now;
        last_label = label;
    }
    ~HostTrace() {
        if (!on) return;
        for (const auto& e : acc)
            std::fprintf(stderr, "tail_trace %-58s %9.1f us mean over %d\n", e.first.c_str(),
                         e.second.first / e.second.second, e.second.second);
    }
};
HostTrace g_trace;
}  // namespace

// ------------------------------------------------------------------ PCG --
namespace {
class PcgSolver final : public LinearSolver {
    Backend* m_be;
    const JacobianPattern& m_pat;
    const HyperParam m_hp;
    DVec m_dinv;
    double m_sign = -1;

public:
    PcgSolver(Backend* be, const JacobianPattern& pat, const HyperParam& hp)
            : m_be{be}, m_pat{pat}, m_hp{hp}, m_dinv{be, (size_t)pat.n()} {}

    void prepare() override {
        // dinv = 1/diag(A); a definite matrix has a diagonal of one sign, which
        // tells which of +A / -A conjugate gradients must run on
        CsrDev A = m_pat.csr();
        m_be->csr_inv_diag(A, 1.0, m_dinv.p());
        double first;
        m_be->d2h(&first, m_dinv.p(), 8);
        m_sign = first < 0 ? -1.0 : 1.0;
        if (m_sign < 0) m_be->axpby(A.n, -1.0, m_dinv.p(), 0, nullptr, m_dinv.p());
    }

    void solve(const double* b, double* x) override {
        int iters = 0;
        double relres = 0;
        m_be->pcg(m_pat.csr(), m_sign, m_dinv.p(), b, x, m_hp.solver_rtol, m_hp.solver_maxit, &iters,
                  &relres);
        if (iters < 0) {
            sanm_throw(SANM_ERR_NUMERICAL,
                       "PCG breakdown after %d iterations: the Jacobian is not definite (use the "
                       "direct solver)",
                       -iters);
        }
        if (!(relres <= std::max(m_hp.solver_rtol * 100, 1e-8))) {
            sanm_throw(SANM_ERR_NUMERICAL, "PCG did not converge: %d iterations, relres=%g", iters,
                       relres);
        }
        ++nr_solve;
        tot_iters += iters;
        last_iters = iters;
        last_relres = relres;
    }
};
}  // namespace

namespace {
class DirectSolver final : public LinearSolver {
    Backend* m_be;
    const JacobianPattern& m_pat;
    std::unique_ptr<Multifrontal> m_mf_own;  // analysed here, or handed in (analysed beside the driver's other tables)
    Multifrontal& m_mf;
    // Iterative refinement (x += A^-1 (b - A x), residual in double-double): `m_refine_always` steps per solve on
    // request (HyperParam::solver_refine / SANM_SOLVER_REFINE: brings the solution within ~1e-14 of the exact one
    // where the plain LU of an elasticity Jacobian of condition 5e8 is off by 4e-11 and PARDISO by 4e-13), and
    // 2 steps after a factorisation that had to perturb pivots -- what PARDISO does with the reference's
    // settings (pardisoinit defaults for mtype 11: iparm[9] = 13, iparm[7] = 0; sparse_solver.cpp:107-127).
    const int m_refine_always;
    int m_refine = 0;
    bool m_residual_checked = true;
    DVec m_r, m_d;

    void refine(const double* b, double* x) {
        const size_t n = m_pat.n();
        if (m_r.empty()) {
            m_r = DVec{m_be, n};
            m_d = DVec{m_be, n};
        }
        for (int s = 0; s < m_refine; ++s) {
            m_be->residual(m_pat.csr(), b, x, m_r.p());
            if (dist()) dist_solve(m_r.p(), m_d.p());
            else m_be->mf_solve(m_mf.dev(), m_mf.schedule(), m_r.p(), m_d.p());
            m_be->axpby(n, 1.0, x, 1.0, m_d.p(), x);
        }
        if (!m_residual_checked) {
            // the first solve after perturbed pivots: the refined solution must actually solve the system
            m_residual_checked = true;
            m_be->residual(m_pat.csr(), b, x, m_r.p());
            const double rr = m_be->dot(n, m_r.p(), m_r.p()), bb = m_be->dot(n, b, b);
            if (!(rr <= 1e-16 * bb))  // relative residual 1e-8
                sanm_throw(SANM_ERR_NUMERICAL,
                           "multifrontal LU: pivots were perturbed and iterative refinement stalls at a relative "
                           "residual of %g; the Jacobian is numerically singular", std::sqrt(rr / std::max(bb, 1e-300)));
        }
    }

    // ---- distributed over the ranks (MfSchedule::Dist, mf_types.h): every front has one owner, the rank's own levels
    //      run stage by stage with the exchanges between them.  Every entry of every exchange has exactly one writer, so
    //      the factors and solutions are those of the single-rank run bit for bit.  m_p2p (the backend's own
    //      communicator): grouped send / receive pairs and broadcasts in place; else the C ABI's all-reduce callback
    //      over a zeroed staging buffer.
    const Collective m_coll;
    const PointToPoint m_p2p;
    DVec m_dist_status;
    bool dist() const { return m_mf.schedule().dist.enabled; }
    //! one staged exchange: pack what this rank sends (src_base -> stage), move it, unpack what it receives
    void exchange(const MfSchedule::Exchange& E, const double* src_base, double* dst_base) {
        const auto& D = m_mf.schedule().dist;
        if (E.doubles == 0) return;
        if (!m_p2p) m_be->zero(D.stage, (size_t)E.doubles * 8);
        m_be->copy2d_batch(E.pack, E.n_pack, E.max_rows, E.max_cols, src_base, D.stage);
        if (m_p2p) m_p2p(D.stage, E.xfers.data(), (int)E.xfers.size());
        else m_coll(D.stage, E.doubles);
        m_be->copy2d_batch(E.unpack, E.n_unpack, E.max_rows, E.max_cols, D.stage, dst_base);
    }
    void dist_factor(double* status) {
        const MfDev& mf = m_mf.dev();
        const MfSchedule& sch = m_mf.schedule();
        const auto& D = sch.dist;
        for (int st = 0; st < D.nr_stage; ++st) {
            // the Schur complements of the children another rank factored
            if (st > 0) exchange(D.schur[st], mf.front_store, mf.front_store);
            m_be->mf_factor_piece(mf, sch, m_pat.csr(), D.stage_level[st], D.stage_level[st + 1], st == 0);
        }
        // perturbed pivots: every rank counts those of its own fronts; anywhere decides for everybody (the refinement
        // they switch on contains collectives)
        if (m_dist_status.empty()) m_dist_status = DVec{m_be, 1};
        m_be->mf_factor_status(mf, m_dist_status.p());
        m_coll(m_dist_status.p(), 1);
        m_be->d2h_async(status, m_dist_status.p(), 8);
    }
    void dist_solve(const double* b, double* x) {
        const MfDev& mf = m_mf.dev();
        const MfSchedule& sch = m_mf.schedule();
        const auto& D = sch.dist;
        m_be->mf_permute(mf, b, nullptr);
        for (int st = 0; st < D.nr_stage; ++st) {
            // the update rows of those children in their parents' inboxes
            if (st > 0) exchange(D.inbox[st], mf.inbox_store, mf.inbox_store);
            m_be->mf_solve_piece(mf, sch, true, D.stage_level[st], D.stage_level[st + 1]);
        }
        for (int st = D.nr_stage - 1; st >= 0; --st) {
            m_be->mf_solve_piece(mf, sch, false, D.stage_level[st], D.stage_level[st + 1]);
            // the stage's pivots to everyone: the levels below read their boundary values there, and every rank ends
            // with the whole solution
            const auto& X = D.sol[st];
            if (X.doubles == 0) continue;
            if (m_p2p) {
                m_p2p(mf.work, X.xfers.data(), (int)X.xfers.size());  // broadcasts in place
            } else if (st > 0) {
                exchange(X, mf.work, mf.work);
            } else {
                for (const auto& r : D.zero_ranges) m_be->zero(mf.work + r.first, (size_t)(r.second - r.first) * 8);
                m_coll(mf.work, mf.n);
            }
        }
        m_be->mf_permute(mf, nullptr, x);
    }

public:
    DirectSolver(Backend* be, const JacobianPattern& pat, const HyperParam& hp, const double* coords, int rank,
                 int world, Collective coll, PointToPoint p2p, std::unique_ptr<Multifrontal> analysed)
            : m_be{be}, m_pat{pat},
              m_mf_own{analysed ? std::move(analysed)
                                : std::make_unique<Multifrontal>(be, pat.n(), pat.h_rowptr(), pat.h_col(), coords, rank, world)},
              m_mf{*m_mf_own},
              m_refine_always{std::getenv("SANM_SOLVER_REFINE") ? std::atoi(std::getenv("SANM_SOLVER_REFINE"))
                                                                : hp.solver_refine},
              m_coll{std::move(coll)},
              m_p2p{std::move(p2p)} {
        sanm_check(!dist() || m_coll, "distributed direct solver without a collective");
        nnz_factors = m_mf.nnz_factors;
        nr_front = m_mf.nr_front;
        nr_level = m_mf.nr_level;
        max_front = m_mf.max_front;
        factor_flops = m_mf.factor_flops;
        const auto& D = m_mf.schedule().dist;
        factor_flops_own = D.flops_own;
        factor_flops_top = D.flops_top;
        factor_flops_top_own = D.flops_top_own;
        factor_flops_critical = D.flops_critical;
        nr_dist_stage = D.enabled ? D.nr_stage : 0;
        front_store_doubles = m_mf.front_doubles;
        nr_subtree = D.nr_subtree;
        nr_subtree_own = D.nr_subtree_own;
        dist_schur_doubles = D.schur_doubles;
        dist_inbox_doubles = D.inbox_doubles;
        m_refine = m_refine_always;
    }
    void prepare() override {
        if (dist()) {
            double* st = m_be->alloc_host(1);
            dist_factor(st);
            m_be->sync();
            const double v = *st;
            m_be->free_host(st);
            (void)check_prepared(&v);
            return;
        }
        int bad = m_be->mf_factor(m_mf.dev(), m_mf.schedule(), m_pat.csr());
        double st = bad;
        (void)check_prepared(&st);
    }
    void prepare_async(double* status) override {
        m_refine = m_refine_always;
        if (dist()) dist_factor(status);
        else m_be->mf_factor_async(m_mf.dev(), m_mf.schedule(), m_pat.csr(), status);
    }
    bool check_prepared(const double* status) override {
        nr_perturbed_pivots = (int64_t)*status;
        if (*status != *status) sanm_throw(SANM_ERR_NUMERICAL, "multifrontal LU: the factorisation status is not a number");
        if (*status == 0) {
            m_refine = m_refine_always;
            return false;
        }
        if (m_refine >= 2) return false;  // already refining: the solves so far stand
        m_refine = 2;
        m_residual_checked = false;
        return true;
    }
    void solve(const double* b, double* x) override {
        if (dist()) dist_solve(b, x);
        else m_be->mf_solve(m_mf.dev(), m_mf.schedule(), b, x);
        if (m_refine > 0) refine(b, x);
        ++nr_solve;
    }
    // (with refinement the right-hand side is needed again after the solve: no fused ends then)
    const int32_t* rhs_perm() const override { return m_refine > 0 || dist() ? nullptr : m_mf.dev().perm; }
    double* rhs_work() const override { return m_mf.dev().work; }
    void solve_fused(const double* b, double* x, const double* dot_y, double* dot_out) override {
        sanm_check(m_refine == 0 || b, "solve_fused without a right-hand side while refining");
        if (m_refine > 0) {
            solve(b, x);
            if (dot_y) m_be->dot_async(m_pat.n(), x, dot_y, dot_out);
            return;
        }
        m_be->mf_solve_fused(m_mf.dev(), m_mf.schedule(), b, x, dot_y, dot_out);
        ++nr_solve;
    }
};
}  // namespace

namespace {
// Tikhonov path (libsanm/sparse_solver.cpp:366-395, :162-176; HyperParam::xcoeff_l2_penalty): the coefficients
// minimise |A x - b|^2 + lambda |x|^2, i.e. solve (A'A + lambda I) x = A'b.  The reference forms A'A with
// mkl_sparse_syrk and factors it as SPD (PARDISO mtype 2) at every step; here the pattern of A'A is built once on
// the host (the Jacobian's pattern is fixed), its values are sparse column dot products on the device
// (Backend::ata), and the same multifrontal code factors it -- an LU of an SPD matrix needs no pivoting.
class TikhonovSolver final : public LinearSolver {
    Backend* m_be;
    const JacobianPattern& m_pat;
    const double m_lambda;
    std::vector<uint32_t> m_h_mrowptr, m_h_mcol;
    CsrDev m_at{}, m_m{};
    uint32_t *m_perm = nullptr, *m_mrow = nullptr;
    DVec m_atval, m_mval, m_rhs;
    std::unique_ptr<Multifrontal> m_mf;
    std::vector<void*> m_bufs;

    template <class T>
    T* upload(const std::vector<T>& v) {
        void* p = m_be->alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
        if (!v.empty()) m_be->h2d(p, v.data(), v.size() * sizeof(T));
        m_bufs.push_back(p);
        return static_cast<T*>(p);
    }

public:
    TikhonovSolver(Backend* be, const JacobianPattern& pat, double lambda, const double* coords)
            : m_be{be}, m_pat{pat}, m_lambda{lambda} {
        const int64_t n = pat.n();
        const auto& rp = pat.h_rowptr();
        const auto& col = pat.h_col();
        const size_t nnz = col.size();
        // A' in CSR form (= A in CSC), with the permutation that carries the values over
        std::vector<uint32_t> cp(n + 1, 0), rows(nnz), perm(nnz);
        for (size_t p = 0; p < nnz; ++p) cp[col[p] + 1]++;
        for (int64_t i = 0; i < n; ++i) cp[i + 1] += cp[i];
        {
            std::vector<uint32_t> fill(cp.begin(), cp.end() - 1);
            for (int64_t r = 0; r < n; ++r)
                for (uint32_t p = rp[r]; p < rp[r + 1]; ++p) {
                    const uint32_t q = fill[col[p]]++;
                    rows[q] = (uint32_t)r;  // ascending: rows are visited in order
                    perm[q] = p;
                }
        }
        // pattern of A'A: column j belongs to row i iff some row of A holds both
        m_h_mrowptr.assign(n + 1, 0);
        std::vector<uint32_t> mrow, tmp;
        for (int64_t i = 0; i < n; ++i) {
            tmp.clear();
            for (uint32_t q = cp[i]; q < cp[i + 1]; ++q) {
                const uint32_t r = rows[q];
                tmp.insert(tmp.end(), col.begin() + rp[r], col.begin() + rp[r + 1]);
            }
            tmp.push_back((uint32_t)i);  // the diagonal carries lambda even where A'A has none
            std::sort(tmp.begin(), tmp.end());
            tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
            m_h_mcol.insert(m_h_mcol.end(), tmp.begin(), tmp.end());
            mrow.insert(mrow.end(), tmp.size(), (uint32_t)i);
            m_h_mrowptr[i + 1] = m_h_mcol.size();
            sanm_check(m_h_mcol.size() < (size_t(1) << 31), "A'A exceeds 32-bit indexing");
        }
        m_atval = DVec{be, std::max<size_t>(nnz, 1)};
        m_mval = DVec{be, m_h_mcol.size()};
        m_rhs = DVec{be, (size_t)n};
        m_at = {upload(cp), upload(rows), m_atval.p(), n, (int64_t)nnz};
        m_m = {upload(m_h_mrowptr), upload(m_h_mcol), m_mval.p(), n, (int64_t)m_h_mcol.size()};
        m_perm = upload(perm);
        m_mrow = upload(mrow);
        m_mf = std::make_unique<Multifrontal>(be, n, m_h_mrowptr, m_h_mcol, coords);
        nnz_factors = m_mf->nnz_factors;
        nr_front = m_mf->nr_front;
        nr_level = m_mf->nr_level;
        max_front = m_mf->max_front;
        factor_flops = m_mf->factor_flops;
    }
    ~TikhonovSolver() override {
        for (void* p : m_bufs) m_be->free(p);
    }
    void prepare() override {
        m_be->gather(m_at.nnz, m_pat.csr().val, m_perm, m_atval.p());
        m_be->ata(m_at, m_m, m_mrow, m_lambda);
        const int bad = m_be->mf_factor(m_mf->dev(), m_mf->schedule(), m_m);
        if (bad)
            sanm_throw(SANM_ERR_NUMERICAL, "Tikhonov path: %d pivot(s) of A'A + %g I below the threshold", bad, m_lambda);
    }
    void solve(const double* b, double* x) override {
        m_be->spmv(m_at, b, m_rhs.p());  // A'b
        m_be->mf_solve(m_mf->dev(), m_mf->schedule(), m_rhs.p(), x);
        ++nr_solve;
    }
};
}  // namespace

namespace {
//! Dense LU with partial pivoting (Backend::dense_lu_factor) for the small general systems of graphs on the vector
//! interpreter: the stand-in for the pivoting PARDISO does on them in the reference (sparse_solver.cpp:107-127)
class DenseLuSolver final : public LinearSolver {
    Backend* m_be;
    const JacobianPattern& m_pat;
    DVec m_lu, m_status;
    void* m_piv = nullptr;
    double* m_host_status = nullptr;

public:
    DenseLuSolver(Backend* be, const JacobianPattern& pat) : m_be{be}, m_pat{pat} {
        const size_t n = pat.n();
        m_lu = DVec{be, n * n};
        m_status = DVec{be, 2};
        m_piv = be->alloc(n * sizeof(int32_t));
        m_host_status = be->alloc_host(2);
        nnz_factors = (int64_t)(n * n);
        factor_flops = 2.0 / 3.0 * (double)n * (double)n * (double)n;
        nr_front = nr_level = 1;
        max_front = (int64_t)n;
    }
    ~DenseLuSolver() override {
        m_be->free(m_piv);
        m_be->free_host(m_host_status);
    }
    void prepare() override {
        prepare_async(nullptr);
        m_be->sync();
        check_prepared(nullptr);
    }
    void prepare_async(double* status) override {
        (void)status;
        m_be->dense_lu_factor(m_pat.csr(), m_lu.p(), static_cast<int32_t*>(m_piv), m_status.p());
        m_be->d2h_async(m_host_status, m_status.p(), 16);
    }
    bool check_prepared(const double* status) override {
        (void)status;
        const double pmin = m_host_status[0], pmax = m_host_status[1];
        // (PARDISO reports a zero pivot the same way: sparse_solver.cpp:118-127)
        if (!(pmin > 0) || !std::isfinite(pmax) || pmin < 1e-14 * pmax)
            sanm_throw(SANM_ERR_NUMERICAL, "dense LU: pivot %g of %g: the Jacobian is numerically singular", pmin, pmax);
        return false;
    }
    void solve(const double* b, double* x) override {
        m_be->dense_lu_solve(m_pat.n(), m_lu.p(), static_cast<const int32_t*>(m_piv), b, x);
        ++nr_solve;
    }
};
}  // namespace

std::unique_ptr<LinearSolver> make_dense_solver(Backend* be, const JacobianPattern& pat) {
    return std::make_unique<DenseLuSolver>(be, pat);
}

std::unique_ptr<LinearSolver> make_direct_solver(Backend* be, const JacobianPattern& pat,
                                                 const HyperParam& hp, const double* coords, int rank, int world,
                                                 Collective coll, PointToPoint p2p,
                                                 std::unique_ptr<Multifrontal> analysed) {
    // (the regularised path factors A'A on every rank: replicated)
    if (hp.xcoeff_l2_penalty != 0) return std::make_unique<TikhonovSolver>(be, pat, hp.xcoeff_l2_penalty, coords);
    return std::make_unique<DirectSolver>(be, pat, hp, coords, rank, world, std::move(coll), std::move(p2p),
                                          std::move(analysed));
}

std::unique_ptr<LinearSolver> make_pcg_solver(Backend* be, const JacobianPattern& pat,
                                              const HyperParam& hp) {
    return std::make_unique<PcgSolver>(be, pat, hp);
}

// ----------------------------------------------------------------- Pade --
void PadeWorkspace::ensure(Backend* be_, int nx, size_t len) {
    if (!orth.empty()) {
        sanm_check((int)orth.size() == nx && orth[1].size() == len, "pade workspace mismatch");
        return;
    }
    be = be_;
    orth.resize(nx);
    for (int i = 1; i < nx; ++i) orth[i] = DVec{be, len};
    acoef = DVec{be, (size_t)nx * nx + nx};
    tmp_row = DVec{be, (size_t)nx + 1};
    host_acoef = be->alloc_host((size_t)nx * nx);
}

// SANM_PADE_ORTH=cgs2 (opt-in, round 5): the Pade basis by classical Gram-Schmidt WITH RE-ORTHOGONALISATION ("twice is
// enough") instead of the reference's single classical sweep (libsanm/pade.cpp:30-55), whose loss of orthogonality --
// kappa^2 eps on the nearly parallel series vectors -- is what makes the fp64 range decisions of the reference's
// algorithm differ from their rounding-free outcome (profiles/r05_pade_arbiter_cgs2.json).  The
// default stays the reference's algorithm: the contract is parity with it.
int pade_orth_mode() {
    static const int mode = [] {
        const char* e = std::getenv("SANM_PADE_ORTH");
        if (!e || !*e || !std::strcmp(e, "cgs")) return 0;
        if (!std::strcmp(e, "cgs2")) return 1;
        sanm_throw(SANM_ERR_ASSERT, "SANM_PADE_ORTH=%s: cgs (the reference's) or cgs2", e);
    }();
    return mode;
}

void PadeWorkspace::step(const std::vector<DVec>& xs, int i, bool anm_cond) {
    if (pade_orth_mode() == 0 || i < 2) {
        for (int k = 1; k <= 3; ++k) phase(xs, i, k, anm_cond, false);
        return;
    }
    // u = x_i - sum_j (x_i . q_j) q_j as the reference does, then once more on the result: u -= sum_j (u . q_j) q_j;
    // the coefficients of the Pade system are the sums of both passes' projections
    phase(xs, i, 1, anm_cond, false);
    phase(xs, i, 2, anm_cond, false);
    const int nx = orth.size();
    double* row = acoef.p() + (size_t)i * nx;
    GsPhase p1;
    p1.kind = 1;
    p1.n = orth[1].size();
    p1.x = orth[i].p();
    p1.nvec = i - 1;
    p1.vecs.resize(p1.nvec);
    for (int j = 1; j < i; ++j) p1.vecs[j - 1] = orth[j].p();
    p1.eps = std::numeric_limits<double>::epsilon();
    p1.red_out = tmp_row.p() + 1;
    be->run_gs_phase(p1);
    GsPhase p2 = p1;
    p2.kind = 2;
    p2.coefs = tmp_row.p() + 1;
    p2.first = anm_cond ? 1 : 0;
    p2.out = orth[i].p();
    p2.red_out = row + i;
    be->run_gs_phase(p2);
    be->axpby((size_t)(i - 1), 1.0, row + 1, 1.0, tmp_row.p() + 1, row + 1);
    phase(xs, i, 3, anm_cond, false);
}

void PadeWorkspace::phase(const std::vector<DVec>& xs, int i, int k, bool anm_cond, bool defer) {
    // Classical Gram-Schmidt (the projections use xs[i], not the running uii).  Nothing here waits for the
    // device: projections and squared norms stay in device memory for the update / scaling kernels that
    // consume them and come back in one copy when the basis is complete.
    const int nx = orth.size(), n = nx - 1;
    sanm_check(i >= 1 && i <= n && ((k == 1 && i == done + 1) || (k > 1 && i == done + 1)), "pade basis: step %d after %d", i,
               done);
    double* nn2 = acoef.p() + (size_t)nx * nx;  // per vector: squared norm after the first scaling
    double* row = acoef.p() + (size_t)i * nx;
    GsPhase ph;
    ph.kind = k;
    ph.n = orth[1].size();
    ph.x = xs[i].p();
    ph.nvec = i - 1;
    // (more than GsPhase::kMaxVec vectors -- orders beyond 25 --: the backend runs the phase in chunks)
    ph.vecs.resize(ph.nvec);
    for (int j = 1; j < i; ++j) ph.vecs[j - 1] = orth[j].p();
    ph.eps = std::numeric_limits<double>::epsilon();
    if (k == 1) {
        // the projection kernel also completes the normalisation of the previous basis vector
        ph.norm2 = i >= 2 ? acoef.p() + (size_t)(i - 1) * nx + (i - 1) : nullptr;
        ph.nn2 = nn2 + (i - 1);
        ph.red_out = row + 1;
    } else if (k == 2) {
        // under the ANM condition the projection on the first basis vector is dropped (checked by the reader)
        ph.coefs = row + 1;
        ph.first = anm_cond ? 1 : 0;
        ph.out = orth[i].p();
        ph.red_out = row + i;
    } else {
        ph.out = orth[i].p();
        ph.norm2 = row + i;
        ph.red_out = nn2 + i;
    }
    if (defer) be->defer_gs_phase(ph);
    else be->run_gs_phase(ph);
    if (k == 3) {
        if (i == n) {
            be->flush_deferred();
            be->gs_renorm_async(ph.n, orth[n].p(), acoef.p() + (size_t)n * nx + n, nn2 + n, ph.eps);
        }
        done = i;
        done_anm_cond = anm_cond;
    }
}

PadeApproximation::PadeApproximation(Backend* be, const std::vector<DVec>& xs,
                                     const std::vector<double>& t_coeffs, bool anm_cond, PadeWorkspace* ws)
        : m_be{be}, m_xs{xs}, m_len{xs[0].size()} {
    // libsanm/pade.cpp:13-105
    const int nx = xs.size();
    sanm_check(nx >= 3, "pade needs at least 3 terms");
    if (m_len < (size_t)nx * 2 || nx <= 4) return;
    const int n = nx - 1;
    std::vector<double> a((size_t)nx * nx, 0.0);
    auto A = [&](int i, int j) -> double& { return a[(size_t)i * nx + j]; };
    // The basis (PadeWorkspace::step).  With a workspace from the driver most or all of its steps are already
    // queued beside the order loop.
    PadeWorkspace local;
    if (!ws) ws = &local;
    ws->ensure(be, nx, m_len);
    if (ws->done > 0 && ws->done_anm_cond != anm_cond) ws->done = 0;
    if (ws->done == 0 && ws != &local && !std::getenv("SANM_NO_PADE_GRAPH") && std::getenv("SANM_PADE_GRAPH")) {
        // (the whole sweep as one graph replay: kept as an experiment switch; the steps queued by the driver
        // beside the order loop made it redundant)
        if (!ws->graph && be->graph_capture_begin()) {
            for (int i = 1; i <= n; ++i) ws->step(xs, i, anm_cond);
            ws->graph = be->graph_capture_end();
            ws->done = 0;
        }
        if (ws->graph) {
            be->graph_launch(ws->graph);
            ws->done = n;
        }
    }
    // the steps the driver has not queued yet (all of them for a stand-alone PadeApproximation)
    for (int i = ws->done + 1; i <= n; ++i) ws->step(xs, i, anm_cond);
    be->side_join();
    ws->done = 0;  // consumed: the next series starts over
    {
        std::vector<double> h((size_t)nx * nx);
        if (ws->host_valid) std::copy(ws->host_acoef, ws->host_acoef + h.size(), h.begin());  // (synchronised since)
        else be->d2h(h.data(), ws->acoef.p(), h.size() * 8);
        ws->host_valid = false;
        for (int i = 1; i <= n; ++i) {
            for (int j = 1; j < i; ++j) {
                A(i, j) = h[(size_t)i * nx + j];
                if (anm_cond && j == 1) {
                    sanm_check(std::fabs(A(i, j)) < 1e-4, "pade: anm condition violated: %g", A(i, j));
                    A(i, j) = 0;
                }
            }
            const double aii = std::sqrt(h[(size_t)i * nx + i]);
            if (aii == 0) {
                m_d.clear();
                return;
            }
            A(i, i) = aii;
        }
    }
    auto solve_d = [&](std::vector<double>& d, int nn) {
        d.assign(nn, 0.0);
        d[0] = 1;
        for (int i = 1; i < nn; ++i) {
            double s = 0;
            for (int j = 0; j < i; ++j) s += A(nn - j, nn - i) * d[j];
            double y = A(nn - i, nn - i);
            d[i] = -s * y / (y * y + 1e-20);
        }
    };
    g_trace.mark("pade: basis on host");
    solve_d(m_d, n);
    solve_d(m_d_lo, n - 1);
    m_t_nume.assign(n, 0.0);
    for (int i = 0; i < n; ++i) {
        double ti = t_coeffs[i];
        if (!i) {
            m_t0 = ti;
        } else {
            for (int j = 0; j < n - i; ++j) m_t_nume[i + j] += m_d[j] * ti;
        }
    }
}

void PadeApproximation::nume_coefs(double a, const double* d, int n, double* coefs) const {
    // libsanm/pade.cpp:181-189: the numerator is sum_{i=1}^{n} a^(i-1) * poly(d[0..n-i], a) * xs[i]
    double ap = 1.0;
    for (int i = 1; i <= n; ++i) {
        coefs[i - 1] = ap * poly::eval(d, n - i + 1, a);
        ap *= a;
    }
}

void PadeApproximation::eval_nume(double a, const double* d, int n, double* out) const {
    // evaluated as one fused linear combination instead of a Horner chain of axpys
    std::vector<const double*> ptrs(n);
    std::vector<double> coefs(n);
    for (int i = 1; i <= n; ++i) ptrs[i - 1] = m_xs[i].p();
    nume_coefs(a, d, n, coefs.data());
    m_be->lincomb(m_len, n, ptrs.data(), coefs.data(), out);
}

double PadeApproximation::eval_t(double a) const {
    return poly::eval(m_t_nume, a) / poly::eval(m_d, a) + m_t0;
}

bool PadeApproximation::estimate_valid_range(double start, double eps, double limit) {
    // libsanm/pade.cpp:107-173
    sanm_check(start > 0 && eps > 0, "pade: bad start/eps");
    m_diag = PadeDiag{};
    m_diag.attempted = 1;
    m_diag.start = start;
    if (m_d.empty()) return false;
    m_diag.built = 1;
    m_diag.d = m_d;
    std::vector<double> roots;
    g_trace.mark("pade: denominators");
    const bool roots_ok = poly::real_roots(m_d, roots);
    g_trace.mark("pade: roots");
    if (!roots_ok) return false;
    m_diag.roots_valid = 1;
    double pole = 0;
    for (double r : roots)
        if (r > 0 && (pole == 0 || r < pole)) pole = r;
    if (pole == 0) pole = start * 4;
    m_diag.pole = pole;
    if (pole <= start) return false;

    const int n = (int)m_xs.size() - 2;
    const double eps2 = eps * eps;
    std::vector<const double*> ptrs(n);
    std::vector<double> c_n(n), c_lo(n);
    for (int i = 1; i <= n; ++i) ptrs[i - 1] = m_xs[i].p();
    // check(a): |P_n(a)/D_n(a) - P_{n-1}(a)/D_{n-1}(a)| <= eps |P_n(a)/D_n(a)| on the numerators.  Several
    // points are probed in one pass over the series and one synchronisation; each point's arithmetic is the
    // same as if it were probed alone.
    auto check_many = [&](const std::vector<double>& as) {
        const int nc = as.size();
        std::vector<double> c1((size_t)nc * n), c2((size_t)nc * n, 0.0), scale(nc), r(2 * nc);
        for (int c = 0; c < nc; ++c) {
            nume_coefs(as[c], m_d.data(), n, c1.data() + (size_t)c * n);
            nume_coefs(as[c], m_d_lo.data(), n - 1, c2.data() + (size_t)c * n);
            scale[c] = poly::eval(m_d, as[c]) / poly::eval(m_d_lo, as[c]);
        }
        g_trace.mark("pade: probe batch prepared");
        m_be->lincomb2_diff_norms_multi(m_len, n, ptrs.data(), nc, c1.data(), c2.data(), scale.data(), r.data());
        g_trace.mark("pade: probe batch returned");
        std::vector<char> ok(nc);
        for (int c = 0; c < nc; ++c) ok[c] = r[2 * c] <= r[2 * c + 1] * eps2;
        m_probe_margin.resize(nc);
        for (int c = 0; c < nc; ++c) m_probe_margin[c] = r[2 * c] / (r[2 * c + 1] * eps2);
        return ok;
    };
    auto note = [&](double a, int slot, bool ok) { m_diag.probes.push_back({a, m_probe_margin[slot], ok ? 1 : 0}); };
    double left = start * 1.001, right = start + (pole - start) * 0.99;
    if (limit && right > limit) right = limit;
    // The probes of pade.cpp:143-165 in TWO batches (two host round trips instead of up to ten): every probe point
    // of the bisection is a function of the interval it starts from, so the midpoints the next `depth` decisions can
    // ask for form a binary tree that is probed as a whole (heap order: node v, children 2v = "failed: right = mid"
    // and 2v + 1 = "passed: left = mid") and the decisions then walk down it.  Batch A = the two opening probes and
    // three levels of the bisection for BOTH outcomes of the doubling probe (16 points), batch B = the remaining
    // five levels (31 points).  Which probes count, and in which order, is decided by the reference's control
    // flow below; the unused ones are discarded.
    auto build_tree = [](double lo0, double hi0, int depth, std::vector<double>& mid) {
        const int nn = (1 << depth) - 1;
        std::vector<double> lo(nn + 1), hi(nn + 1);
        mid.assign(nn + 1, 0.0);
        lo[1] = lo0;
        hi[1] = hi0;
        for (int v = 1; v <= nn; ++v) {
            mid[v] = (lo[v] + hi[v]) / 2;
            if (2 * v + 1 <= nn) {
                lo[2 * v] = lo[v];
                hi[2 * v] = mid[v];
                lo[2 * v + 1] = mid[v];
                hi[2 * v + 1] = hi[v];
            }
        }
    };
    int iter = 0;
    // walks a probed tree from its root; ok[] / margins are indexed like `as` of the batch, base = index of node 1 - 1
    auto walk = [&](const std::vector<double>& mid, int depth, const std::vector<char>& ok, int base) {
        const int nn = (1 << depth) - 1;
        for (int v = 1; v <= nn && iter < 8 && right - left > 1e-3; ++iter) {
            const bool pass = ok[base + v];
            note(mid[v], base + v, pass);
            if (pass) left = mid[v];
            else right = mid[v];
            v = 2 * v + (pass ? 1 : 0);
        }
    };
    {
        const bool dbl = right > start * 2;
        std::vector<double> as{left};
        if (dbl) as.push_back(start * 2);
        const int first_tree = (int)as.size();
        std::vector<double> midA, midB;
        constexpr int kDepthA = 3;
        if (dbl) {
            build_tree(start * 2, right, kDepthA, midA);  // the doubling probe passed: left = 2 start
            build_tree(left, start * 2, kDepthA, midB);   // it failed: right = 2 start
            as.insert(as.end(), midA.begin() + 1, midA.end());
            as.insert(as.end(), midB.begin() + 1, midB.end());
        } else {
            build_tree(left, right, kDepthA, midA);
            as.insert(as.end(), midA.begin() + 1, midA.end());
        }
        const std::vector<char> ok = check_many(as);
        note(left, 0, ok[0]);
        if (!ok[0]) return false;
        if (dbl) {
            note(start * 2, 1, ok[1]);
            if (ok[1]) {
                left = start * 2;
                walk(midA, kDepthA, ok, first_tree - 1);
            } else {
                right = start * 2;
                walk(midB, kDepthA, ok, first_tree - 1 + (int)midA.size() - 1);
            }
        } else {
            walk(midA, kDepthA, ok, first_tree - 1);
        }
    }
    if (iter < 8 && right - left > 1e-3) {
        const int depth = 8 - iter;
        std::vector<double> mid;
        build_tree(left, right, depth, mid);
        const std::vector<char> ok = check_many(std::vector<double>(mid.begin() + 1, mid.end()));
        walk(mid, depth, ok, -1);
    }
    m_t_max_a = left;
    m_t_max = eval_t(left);
    m_diag.accepted = 1;
    m_diag.t_max_a = left;
    return true;
}

double PadeApproximation::solve_a(double t) const {
    // libsanm/pade.cpp:191-201
    sanm_check(t >= m_t0 && t <= m_t_max, "pade solve_a: t=%g out of [%g, %g]", t, m_t0, m_t_max);
    if (t == m_t_max) return m_t_max_a;
    std::vector<double> c = m_t_nume;
    for (size_t i = 0; i < c.size(); ++i) c[i] -= (t - m_t0) * m_d[i];
    return poly::solve_eqn(c, 0, m_t_max_a, 0);
}

void PadeApproximation::eval_xt(double a, double* out) const {
    // libsanm/pade.cpp:214-219
    eval_nume(a, m_d.data(), (int)m_xs.size() - 2, out);
    m_be->axpby(m_len, a / poly::eval(m_d, a), out, 1.0, m_xs[0].p(), out);
}

// ------------------------------------------------------------ AnmDriver --
// ---- spatial tet order ------------------------------------------------------
// The Taylor passes stream the per-tet state and do not care about the order of the tets, but remap_out
// and the Jacobian assembly GATHER single doubles from per-tet arrays: a 64-byte line holds 8 consecutive
// tets, and it only serves more than one of the ~10^7 gathered entries if consecutive tets are neighbours
// in the mesh.  Mesh generators do not number them that way, so when the caller supplied the positions of
// the unknowns (sanm_sparse_desc_set_out_coords) the tets are renumbered along a Morton curve through
// their centroids.  The renumbering is internal to the driver: everything it exposes lives in the space
// of the unknowns.
namespace {
std::vector<int64_t> spatial_tet_order(const SparseDesc& remap_inp, const std::vector<double>& coords,
                                       int64_t n) {
    const int64_t T = remap_inp.out_size / 9;
    std::vector<double> cen((size_t)T * 3, 0.0);
    constexpr int kMaxThr = 64;
    double tlo[kMaxThr][3], thi[kMaxThr][3];
    for (int t = 0; t < kMaxThr; ++t)
        for (int d = 0; d < 3; ++d) tlo[t][d] = 1e300, thi[t][d] = -1e300;
    parallel_ranges(T, 4096, [&](int64_t e0, int64_t e1, int t) {
        double* lo = tlo[t % kMaxThr];  // min / max are exact: any grouping gives the same box
        double* hi = thi[t % kMaxThr];
        for (int64_t e = e0; e < e1; ++e) {
            double acc[3] = {0, 0, 0};
            int64_t cnt = 0;
            for (uint64_t q = remap_inp.rowptr[e * 9]; q < remap_inp.rowptr[e * 9 + 9]; ++q) {
                const int64_t u = remap_inp.idx[q];
                if (u >= n) continue;  // the t column of the implicit solver
                for (int d = 0; d < 3; ++d) acc[d] += coords[u * 3 + d];
                ++cnt;
            }
            for (int d = 0; d < 3; ++d) {
                cen[e * 3 + d] = cnt ? acc[d] / cnt : 0.0;
                lo[d] = std::min(lo[d], cen[e * 3 + d]);
                hi[d] = std::max(hi[d], cen[e * 3 + d]);
            }
        }
    });
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int t = 0; t < kMaxThr; ++t)
        for (int d = 0; d < 3; ++d) lo[d] = std::min(lo[d], tlo[t][d]), hi[d] = std::max(hi[d], thi[t][d]);
    auto spread = [](uint64_t v) {  // 21 bits -> every third bit
        v &= 0x1fffff;
        v = (v | v << 32) & 0x1f00000000ffffULL;
        v = (v | v << 16) & 0x1f0000ff0000ffULL;
        v = (v | v << 8) & 0x100f00f00f00f00fULL;
        v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
        v = (v | v << 2) & 0x1249249249249249ULL;
        return v;
    };
    std::vector<std::pair<uint64_t, int64_t>> key(T);
    parallel_ranges(T, 4096, [&](int64_t e0, int64_t e1, int) {
        for (int64_t e = e0; e < e1; ++e) {
            uint64_t k = 0;
            for (int d = 0; d < 3; ++d) {
                const double ext = hi[d] - lo[d];
                const double f = ext > 0 ? (cen[e * 3 + d] - lo[d]) / ext : 0.0;
                k |= spread((uint64_t)(f * 2097151.0)) << d;
            }
            key[e] = {k, e};
        }
    });
    // (the keys carry the tet number: a total order)
    parallel_sort(key.begin(), key.end(), std::less<std::pair<uint64_t, int64_t>>{});
    std::vector<int64_t> order(T);
    for (int64_t e = 0; e < T; ++e) order[e] = key[e].second;  // new tet e = old tet order[e]
    return order;
}

}  // namespace

AnmDriver::AnmDriver(Backend* be, const Graph& g_in, int out_var, const SparseDesc& remap_inp_in,
                     const SparseDesc& remap_out_in, int64_t nr_unknown, const HyperParam& hp,
                     const ShardInfo& shard)
        : m_be{be}, m_hp{hp}, m_n{nr_unknown}, m_max_a_bound{poly::stable_x_range(hp.order)},
          m_shard{shard}, m_profile_mode{hp.profile} {
    sanm_check(hp.order >= 2, "order=%d", hp.order);  // anm.cpp:108-110
    if (graph_is_vector(g_in, out_var)) {
        construct_on_vector_interpreter(g_in, out_var, remap_inp_in, remap_out_in);
        return;
    }
    sanm_check(remap_inp_in.out_size % 9 == 0, "remap_inp must produce a (T,3,3) tensor");
    if (hp.xcoeff_l2_penalty != 0 && hp.solver_kind != 1)
        sanm_throw(SANM_ERR_UNSUPPORTED, "xcoeff_l2_penalty (Tikhonov path) needs the direct solver (solver_kind 1)");
    auto clk = [] { return std::chrono::steady_clock::now(); };
    auto lap = [&](const char* name, std::chrono::steady_clock::time_point& t0) {
        const auto t1 = clk();
        const double sec = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        for (auto& kv : m_setup)
            if (kv.first == name) {  // (a phase in two pieces: one entry)
                kv.second += sec;
                return;
            }
        m_setup.emplace_back(name, sec);
    };
    auto t_setup = clk();
    const Graph& g = g_in;
    const SparseDesc& remap_inp = remap_inp_in;
    const SparseDesc& remap_out = remap_out_in;
    const int64_t T = remap_inp.out_size / 9;
    // this rank's tets: contiguous ranges like the reference's worker shards
    // (libsanm/symbolic.cpp:525-536)
    int64_t tb = 0, te = T;
    if (m_shard.active()) {
        sanm_check(m_shard.world >= 1 && m_shard.rank >= 0 && m_shard.rank < m_shard.world,
                   "invalid shard description");
        if (!m_shard.allreduce)
            sanm_check(be->comm_world() == m_shard.world && be->comm_rank() == m_shard.rank,
                       "sharded solver without a callback: the backend's communicator (sanm_hip_comm_init) has rank "
                       "%d of %d, the solver was given rank %d of %d",
                       be->comm_rank(), be->comm_world(), m_shard.rank, m_shard.world);
        tb = (int64_t)m_shard.rank * T / m_shard.world;
        te = (int64_t)(m_shard.rank + 1) * T / m_shard.world;
        sanm_check(te > tb, "more ranks than tets");
    }
    // the renumbering is applied while the device tables are built (Program, DeviceRows, JacobianPattern read the
    // caller's graph constants and remap tables through it): no permuted copies of either
    std::vector<int64_t> order, inv;
    const bool reorder = remap_out_in.out_coords.size() == (size_t)m_n * 3 &&
                         remap_out_in.in_size == remap_inp_in.out_size && !std::getenv("SANM_NO_TET_ORDER");
    // The Jacobian's pattern first (host only), so that the direct solver's analysis -- host only as well, and the
    // longest piece of the constructor -- runs on a thread of its own beside the program, the remap tables and the
    // assembly lists; its device copies are made by this thread once the others are done.  The host pattern of ALL tets
    // does not depend on their numbering: it is built beside the renumbering (round 6; a shard's pattern needs the
    // numbering first).  SANM_SETUP_SERIAL=1: one thing after the other.
    const bool serial = std::getenv("SANM_SETUP_SERIAL") != nullptr;
    const double* coords = remap_out.out_coords.size() == (size_t)m_n * 3 ? remap_out.out_coords.data() : nullptr;
    // (the merged top block, SANM_MF_TOP > 0, multiplies device blocks out while the analysis builds it, i.e. it is
    // not deferred: it must run on this thread, the owner of the backend, whose pool and staging buffers have no lock)
    const bool mf_top = std::getenv("SANM_MF_TOP") && std::atoi(std::getenv("SANM_MF_TOP")) > 0;
    const bool beside = hp.solver_kind == 1 && hp.xcoeff_l2_penalty == 0 && !serial && !mf_top;
    const bool dist = m_shard.active() && m_shard.world > 1;
    // The analysis starts as soon as the pattern's BLOCK rows exist (sparse.h, on_blocks: the Jacobian of a mesh with three
    // unknowns per vertex), i.e. while the rows of the unknowns are still being written out and the tets renumbered
    // (round 6); a pattern without that structure hands the analysis its rows when they are complete.
    std::future<std::unique_ptr<Multifrontal>> analysis;
    // (whatever fails below, the thread is joined before what it reads goes away)
    struct Join {
        std::future<std::unique_ptr<Multifrontal>>& f;
        ~Join() {
            if (f.valid()) f.wait();
        }
    } join{analysis};
    auto on_blocks = [&](JacobianPattern::BlockRows qptr, JacobianPattern::BlockRows qcol) {
        if (!beside || std::getenv("SANM_ANALYSIS_FROM_ROWS")) return;
        analysis = std::async(std::launch::async, [this, be, coords, dist, qptr, qcol] {
            return std::make_unique<Multifrontal>(be, m_n, Multifrontal::BlockPattern{3, qptr, qcol}, coords,
                                                  dist ? m_shard.rank : 0, dist ? m_shard.world : 1, /*defer_device=*/true);
        });
    };
    auto make_pattern = [&](const int64_t* tet_order, const int64_t* tet_inv) {
        m_pattern = std::make_unique<JacobianPattern>(be, remap_out, remap_inp, m_n, T, 0, 9, tb, te, 9, tet_order, tet_inv,
                                                      /*defer_device=*/true, on_blocks);
    };
    const bool pattern_beside = reorder && !serial && tb == 0 && te == T;
    {
        std::future<void> pattern_job;  // (joined when the scope ends, whatever ends it)
        if (pattern_beside) pattern_job = std::async(std::launch::async, [&] { make_pattern(nullptr, nullptr); });
        if (reorder) {
            order = spatial_tet_order(remap_inp_in, remap_out_in.out_coords, m_n);
            inv.resize(order.size());
            for (size_t e = 0; e < order.size(); ++e) inv[order[e]] = (int64_t)e;
        }
        lap("tet_order", t_setup);
        if (pattern_job.valid()) pattern_job.get();  // (rethrows)
    }
    const int64_t* tet_order = reorder ? order.data() : nullptr;
    const int64_t* tet_inv = reorder ? inv.data() : nullptr;
    if (pattern_beside) m_pattern->set_tet_order(tet_order, tet_inv);
    else make_pattern(tet_order, tet_inv);
    lap("pattern", t_setup);
    if (beside && !analysis.valid()) {
        analysis = std::async(std::launch::async, [this, be, coords, dist] {
            return std::make_unique<Multifrontal>(be, m_n, m_pattern->h_rowptr(), m_pattern->h_col(), coords,
                                                  dist ? m_shard.rank : 0, dist ? m_shard.world : 1, /*defer_device=*/true);
        });
    }
    m_prog = std::make_unique<Program>(be, g, out_var, te - tb, hp.order, tb, T, /*full_history=*/false, tet_order);
    lap("program", t_setup);
    m_setup.back().second -= m_prog->jit_seconds;
    m_setup.emplace_back("jit", m_prog->jit_seconds);
    m_setup.emplace_back(m_prog->jit_source == 4 ? "jit_embedded" : m_prog->jit_source == 3 ? "jit_compiled" : m_prog->jit_source == 2 ? "jit_disk_hit"
                         : m_prog->jit_source == 1 ? "jit_memory_hit" : "jit_none", 1.0);
    sanm_check(m_prog->dev().odim == 9, "the ANM solvers take a graph whose output is a batched 3x3 matrix");
    if (serial) {
        m_prog->set_remap_in(remap_inp.in_size, remap_inp.rowptr.data(), remap_inp.idx.data(),
                             remap_inp.coef.data());
        m_remap_out = std::make_unique<DeviceRows>(be, remap_out, te - tb, m_prog->Tpad(), tb, te, 9, tet_inv);
        lap("remap_tables", t_setup);
        m_pattern->finish_device(remap_out, remap_inp);
        lap("pattern", t_setup);
    } else {
        // the host halves of the two remap tables on a thread beside the device side of the pattern (round 6); their
        // uploads follow on this thread
        auto tables = std::async(std::launch::async, [&] {
            return std::make_pair(m_prog->prepare_remap_in(remap_inp.in_size, remap_inp.rowptr.data(), remap_inp.idx.data(),
                                                           remap_inp.coef.data()),
                                  DeviceRows::pack_host(remap_out, te - tb, m_prog->Tpad(), tb, te, 9, tet_inv));
        });
        m_pattern->finish_device(remap_out, remap_inp);
        lap("pattern", t_setup);
        auto host = tables.get();  // (rethrows)
        m_prog->set_remap_in(std::move(host.first));
        m_remap_out = std::make_unique<DeviceRows>(be, std::move(host.second));
        lap("remap_tables", t_setup);
    }
    std::unique_ptr<Multifrontal> analysed;
    if (analysis.valid()) {
        analysed = analysis.get();  // (rethrows what the analysis threw)
        m_setup.emplace_back("analysis_thread", analysed->analysis_seconds);  // its own clock; "analysis" is the wait for it
        lap("analysis", t_setup);
        analysed->finish_device();
        lap("analysis_device", t_setup);  // the queued uploads and allocations, the front store among them
    }
    construct_solver_and_vectors(coords, std::move(analysed));
    lap(beside ? "solver_vectors" : "analysis", t_setup);
    if (std::getenv("SANM_DEBUG_SETUP")) {
        for (const auto& kv : m_setup) std::fprintf(stderr, "[setup] %s %.4f\n", kv.first.c_str(), kv.second);
        std::fprintf(stderr, "[setup] driver constructor, first member to here: %.4f\n",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - m_ctor_begin).count());
    }
}

void AnmDriver::construct_on_vector_interpreter(const Graph& g, int out_var, const SparseDesc& remap_inp,
                                                const SparseDesc& remap_out) {
    // tests/symbolic.cpp:140-560, :835-900: the ANM solvers over graphs of arbitrary (batch, ...) tensors
    Backend* be = m_be;
    if (m_shard.active())
        sanm_throw(SANM_ERR_UNSUPPORTED, "tet-sharded solvers take (T,3,3) graphs (the vector interpreter is not sharded)");
    if (m_hp.xcoeff_l2_penalty != 0 && m_hp.solver_kind != 1)
        sanm_throw(SANM_ERR_UNSUPPORTED, "xcoeff_l2_penalty (Tikhonov path) needs the direct solver (solver_kind 1)");
    int idim = 0;
    for (const GraphOp& op : g.ops)
        if (op.type == OP_PLACEHOLDER) idim = g.vars[op.out[0]].size;
    sanm_check(idim > 0, "the graph has no placeholder");
    sanm_check(remap_inp.out_size % idim == 0, "remap_inp produces %ld elements: not a batch of the placeholder's %d",
               (long)remap_inp.out_size, idim);
    const int64_t B = remap_inp.out_size / idim;
    m_vprog = std::make_unique<VecProgram>(be, g, out_var, B, m_hp.order);
    sanm_check(m_vprog->idim() == idim, "the graph holds several placeholders of different sizes (%d, %d): only the "
               "one the output depends on may be declared", m_vprog->idim(), idim);
    const int odim = m_vprog->odim();
    sanm_check(remap_out.in_size == B * odim, "remap_out takes %ld elements, the graph produces (%ld, %d)",
               (long)remap_out.in_size, (long)B, odim);
    // x (n or n+1 entries) -> placeholder rows: one "batch item" holding the whole vector
    m_vec_remap_in = std::make_unique<DeviceRows>(be, remap_inp, 1, 1, 0, 1, remap_inp.in_size);
    m_vec_xin = DVec{be, (size_t)(B * idim)};
    m_remap_out = std::make_unique<DeviceRows>(be, remap_out, B, B, 0, B, odim);
    m_pattern = std::make_unique<JacobianPattern>(be, remap_out, remap_inp, m_n, B, B, odim, 0, B, idim);
    construct_solver_and_vectors(nullptr);
}

void AnmDriver::construct_solver_and_vectors(const double* coords, std::unique_ptr<Multifrontal> analysed) {
    Backend* be = m_be;
    const HyperParam& hp = m_hp;
    // Graphs on the vector interpreter: general small systems (a transpose or a random sparse map puts zeros on
    // the diagonal) -- dense LU with partial pivoting up to kDenseMaxN unknowns, the multifrontal solver beyond
    // and on the regularised path
    constexpr int64_t kDenseMaxN = 4096;
    if (hp.solver_kind == 1 && m_vprog && m_n <= kDenseMaxN && hp.xcoeff_l2_penalty == 0 &&
        !std::getenv("SANM_VEC_MULTIFRONTAL")) {
        m_solver = make_dense_solver(be, *m_pattern);
    } else if (hp.solver_kind == 1) {
        // tet-sharded over several ranks: factorisation and solves by subtrees where that pays (multifrontal.cpp)
        if (m_shard.active() && m_shard.world > 1) {
            // point-to-point transfers where the collective is the backend's own communicator (SANM_DIST_P2P=0: the
            // all-reduce form of the exchanges there too); the callback of the C ABI offers the all-reduce only
            PointToPoint p2p;
            const char* env_p2p = std::getenv("SANM_DIST_P2P");
            if (test_p2p().fn) {  // (tests: the point-to-point branch over a callback, sanm_hip_test.h)
                static_assert(sizeof(MfSchedule::Xfer) == 32 && offsetof(MfSchedule::Xfer, off) == 8 &&
                                      offsetof(MfSchedule::Xfer, cnt) == 16 && offsetof(MfSchedule::Xfer, src_stage) == 24,
                              "sanm_test_xfer mirrors MfSchedule::Xfer");
                const TestP2p hook = test_p2p();
                p2p = [be, hook](double* base, const MfSchedule::Xfer* x, int n) {
                    be->sync();
                    const int rc = hook.fn(hook.user, base, x, n);
                    sanm_check(rc == 0, "point-to-point callback failed (%d)", rc);
                };
            } else if (!m_shard.allreduce && be->comm_p2p_available() && !(env_p2p && std::atoi(env_p2p) == 0))
                p2p = [this](double* base, const MfSchedule::Xfer* x, int n) { exchange_p2p(base, x, n); };
            m_solver = make_direct_solver(be, *m_pattern, hp, coords, m_shard.rank, m_shard.world,
                                          [this](double* p, int64_t c) { allreduce(p, c); }, std::move(p2p),
                                          std::move(analysed));
        } else {
            m_solver = make_direct_solver(be, *m_pattern, hp, coords, 0, 1, {}, {}, std::move(analysed));
        }
    } else if (hp.solver_kind == 0) {
        m_solver = make_pcg_solver(be, *m_pattern, hp);
    } else if (hp.solver_kind == 2) {
        m_solver.reset(be->make_external_solver(*m_pattern, hp));
        if (!m_solver)
            sanm_throw(SANM_ERR_UNSUPPORTED, "solver_kind 2 (host MKL PARDISO) exists in the CPU baseline harness only");
    } else {
        sanm_throw(SANM_ERR_ASSERT, "unknown solver_kind %d", hp.solver_kind);
    }
    const size_t n1 = m_n + 1;
    m_xt0 = DVec{be, n1};
    m_fx0 = DVec{be, (size_t)m_n};
    m_bi = DVec{be, (size_t)m_n};
    m_xbi = DVec{be, (size_t)m_n};
    m_xgt = DVec{be, (size_t)m_n};
    m_grad_t_buf = DVec{be, (size_t)m_n};
    m_tmp0 = DVec{be, n1};
    m_tmp1 = DVec{be, n1};
    m_dev_scalars = DVec{be, (size_t)hp.order + 2};
    // per order: t_i, then (after all of them) the two results of its sanity check
    m_host_scalars = be->alloc_host(5 * ((size_t)hp.order + 2) + 8);
    if (hp.sanity_check && !hp.xcoeff_l2_penalty) {  // anm.cpp:271: no check on the regularised path
        m_bi_all.resize(hp.order + 1);
        for (int i = 1; i <= hp.order; ++i) m_bi_all[i] = DVec{be, (size_t)m_n};
    }
    m_xt_coeffs.resize(hp.order + 1);
    for (auto& v : m_xt_coeffs) v = DVec{be, n1};
}

void AnmDriver::run_pass(int mode, int order, const double* x) {
    if (m_prog) {
        m_be->run_pass(m_prog->dev(), (PassMode)mode, order, x);
        return;
    }
    const double* xin = nullptr;
    if (x) {  // remap_inp (anm.cpp:362-438's input side) as a gather in front of the interpreter
        m_be->gather_rows(m_vec_remap_in->dev(), x, m_vec_xin.p());
        xin = m_vec_xin.p();
    }
    if (mode == PASS_COEFF_BIAS) {
        m_be->run_vec_pass(m_vprog->dev(), PASS_COEFF, order, xin);
        m_be->run_vec_pass(m_vprog->dev(), PASS_BIAS, order + 1, nullptr);
        return;
    }
    m_be->run_vec_pass(m_vprog->dev(), mode, order, xin);
}

// (the per-tet programs stage value and bias in one tet-major buffer)
const double* AnmDriver::out_value0() const { return m_prog ? m_prog->out_coef0() : m_vprog->out_coef0_dev(); }
const double* AnmDriver::out_bias() const { return m_prog ? m_prog->out_bias() : m_vprog->out_bias_dev(); }

double* AnmDriver::pow_flag_words() const {
    if (m_vprog) return m_vprog->flag_dev();
    return m_prog->pow_flags().empty() ? nullptr : m_prog->arena_dev() + m_prog->pow_flags()[0].off;
}

std::string AnmDriver::pow_exponent_list() const {
    std::string exps;
    if (m_prog)
        for (const auto& f : m_prog->pow_flags()) exps += (exps.empty() ? "" : ", ") + std::to_string(f.exponent);
    else
        for (double e : m_vprog->pow_exponents()) exps += (exps.empty() ? "" : ", ") + std::to_string(e);
    return exps;
}

const double* AnmDriver::jacobian_blocks() const { return m_prog ? m_prog->placeholder_jac() : m_vprog->jac_dev(); }

int64_t AnmDriver::batch() const { return m_prog ? m_prog->T() : m_vprog->B(); }
size_t AnmDriver::arena_bytes() const { return m_prog ? m_prog->arena_bytes() : m_vprog->arena_bytes(); }

const std::map<std::string, double>& AnmDriver::profile() {
    if (m_profile_mode == 2) m_be->phase_collect(m_profile, &m_profile_cnt);
    return m_profile;
}

AnmDriver::~AnmDriver() {
    if (m_host_scalars) m_be->free_host(m_host_scalars);
}

void AnmDriver::apply_injection(double* vec, int64_t len) {
    // test hook (sanm_anm_debug_inject): overwrite or scale one entry of a device vector, once
    const int64_t cnt = m_inject.kind == 3 ? std::max(m_inject.order, 1) : 1;  // kind 3: `order` entries in a row
    sanm_check(m_inject.index >= 0 && m_inject.index + cnt <= len, "injection index out of range");
    std::vector<double> v(cnt);
    m_be->d2h(v.data(), vec + m_inject.index, 8 * cnt);
    for (double& e : v) e = m_inject.scale ? e * m_inject.value : m_inject.value;
    m_be->h2d(vec + m_inject.index, v.data(), 8 * cnt);
    m_inject.kind = 0;
}

void AnmDriver::allreduce(double* buf, int64_t count) {
    if (!m_shard.active()) return;
    ScopedTimer t{this, "allreduce"};
    if (!m_shard.allreduce) {
        m_be->allreduce_sum(buf, count);  // queued on the backend's stream
        return;
    }
    m_be->sync();  // the callback's collective runs outside this backend's stream
    int rc = m_shard.allreduce(m_shard.user, buf, count);
    if (rc != 0) sanm_throw(SANM_ERR_HIP, "all-reduce callback failed with code %d", rc);
}

TestP2p& test_p2p() {
    static TestP2p hook;
    return hook;
}

void AnmDriver::exchange_p2p(double* base, const MfSchedule::Xfer* x, int n) {
    ScopedTimer t{this, "allreduce"};  // (booked with the collectives)
    m_be->comm_exchange(base, x, n);   // queued on the backend's stream
}

void AnmDriver::init_xt0(const double* x_host, double t) {
    std::vector<double> h(m_n + 1);
    std::copy(x_host, x_host + m_n, h.begin());
    h[m_n] = t;
    m_be->h2d(m_xt0.p(), h.data(), h.size() * 8);
    m_t0_host = t;
    m_t0_known = true;
}

void AnmDriver::solve_expansion_coeffs() {
    // libsanm/anm.cpp:193-312
    ScopedTimer timer_all{this, "solve_expansion_coeffs"};
    const int N = m_hp.order;
    const size_t n = m_n, n1 = m_n + 1;
    Backend* be = m_be;

    const auto inject_at_entry = m_inject;
    be->d2d(m_xt_coeffs[0].p(), m_xt0.p(), n1 * 8);
    m_nr_valid_coeffs = 1;
    m_t_coeffs.assign(1, 0.0);
    // t_0: known on the host when the caller has just set it (no round trip to the device then)
    if (m_t0_known) m_t_coeffs[0] = m_t0_host;
    else be->d2h(&m_t_coeffs[0], m_xt0.p() + n, 8);
    m_t0_known = false;
    trace_b_norm.clear();
    trace_x_norm.clear();
    trace_t.clear();
    m_trace_xbi_norm.clear();
    static const bool env_verbose = std::getenv("SANM_VERBOSE") != nullptr;
    const bool verbose = env_verbose || m_profile_mode == 1;

    {
        ScopedTimer t{this, "taylor_order0"};
        run_pass(PASS_EVAL0, 0, m_xt0.p());
        {
            ScopedTimer t2{this, "remap_out"};
            be->gather_rows(m_remap_out->dev(), out_value0(), m_fx0.p());
        }
        allreduce(m_fx0.p(), n);
    }
    // 0^p in a pow operator (analytic_unary.cpp:112-131): the order-0 pass raised a flag in the arena; its square
    // travels to the host with the next synchronisation
    double* const host_powflag = m_host_scalars + 5 * ((size_t)N + 2) + 4;
    *host_powflag = 0;
    double* const flag_words = pow_flag_words();
    const bool has_powflag = flag_words != nullptr;
    auto check_powflag = [&]() {
        if (!has_powflag || *host_powflag == 0) return;
        // |flags|^2 = 1 or 5: a non-integer exponent met a zero (the reference's error; it wins); 4: only the order limit
        const bool unsupported = *host_powflag == 4.0;
        const double zero[2] = {0, 0};
        be->h2d(flag_words, zero, 16);
        const std::string exps = pow_exponent_list();
        if (unsupported && !m_prog)
            sanm_throw(SANM_ERR_UNSUPPORTED, "integer power other than a square of a series through zero on the vector "
                                             "interpreter (exponents: %s)", exps.c_str());
        if (unsupported)
            sanm_throw(SANM_ERR_UNSUPPORTED, "integer power of a series through zero beyond order %d (exponents: %s)",
                       POW_INT_MAX_ORDER, exps.c_str());
        sanm_throw(SANM_ERR_NUMERICAL, "0^p when p is not integer (pow exponents in the graph: %s)", exps.c_str());
    };
    if (has_powflag) be->dot_async(2, flag_words, flag_words, host_powflag);
    if (!on_fx0_computed(m_fx0.p())) {
        check_powflag();
        return;
    }
    if (has_powflag) be->sync();  // (most on_fx0_computed variants have waited for the device already)
    check_powflag();

    double t1 = 0, xgt_dot_x1 = 0;
    bool order1_on_device = false;  // t_1 and xgt . x_1 never came to the host (Backend::x1_async)
    const double* grad_t = nullptr;
    const int32_t* rhs_perm = nullptr;
    // COEFF(i) and BIAS(i+1) are back to back: one launch among the kernels compiled for this graph
    const bool fuse_passes = m_prog && m_prog->dev().spec_id >= 0 && !std::getenv("SANM_NO_FUSED_PASS");
    bool bias_done = false;
    double* const host_sanity = m_host_scalars + 3 * ((size_t)N + 2);  // [2 (i-1)], [2 (i-1) + 1]
    // asynchronous results examined after the loop: non-finite Jacobian entries, rejected pivots, |x_1|^2, |x_N|^2
    double* const host_checks = m_host_scalars + 5 * ((size_t)N + 2);
    host_checks[0] = host_checks[1] = 0;
    // The Pade basis (pade.cpp:36-70) grows by one Gram-Schmidt step per order instead of being built after the
    // loop -- speculatively: estimate_valid_range decides later whether it is used.  The step for x_{i-1} is
    // handed to the backend in its three phases at the points of order i where the kernels that can carry them
    // are launched (remap_out gather, last kernel of the solve, next_coeff: Backend::defer_gs_phase); a backend
    // that cannot carry them runs them as launches of their own at those points, which is the same arithmetic.
    // SANM_GS_MODE: "tail" = the whole basis after the loop, "side" = steps on a second queue beside the loop
    // (measured: the cross-queue traffic slows the solve's latency-bound launches by more than it hides),
    // default = riders.
    static const bool env_pade = getenv("SANM_PADE") != nullptr;
    static const int gs_mode = [] {
        const char* e = getenv("SANM_GS_MODE");
        // (the re-orthogonalised basis is built step by step behind the order loop: no riders)
        if (pade_orth_mode() != 0) return 0;
        return !e ? 2 : (!strcmp(e, "tail") ? 0 : (!strcmp(e, "side") ? 1 : 2));
    }();
    const bool anm_cond = !m_hp.xcoeff_l2_penalty;
    const bool pade_steps = (m_hp.use_pade || env_pade) && gs_mode != 0 && n1 >= 2 * ((size_t)N + 1) && N + 1 > 4;
    const bool pade_riders = pade_steps && gs_mode == 2, pade_side = pade_steps && gs_mode == 1;
    m_pade_ws.done = 0;
    if (pade_steps) m_pade_ws.ensure(be, N + 1, n1);
    const bool do_sanity = m_hp.sanity_check && !m_hp.xcoeff_l2_penalty;
    int sanity_done = 0;  // orders 1 .. sanity_done are queued
    auto queue_sanity = [&](int upto, const double* grad_t_dev) {
        // anm.cpp:271-285: A x_i = -(t_i g_t + b_i) and x_1 . x_i = delta_1i, in one pass over the matrix per 10
        // orders (the reference checks each order as it goes; a failure surfaces after the loop here)
        ScopedTimer t{this, "anm_sanity_check"};
        std::vector<const double*> xs, bs;
        for (int q = sanity_done + 1; q <= upto; ++q) {
            xs.push_back(m_xt_coeffs[q].p());
            bs.push_back(m_bi_all[q].p());
        }
        be->sanity_check_batch_async(m_pattern->csr(), (int)xs.size(), xs.data(), grad_t_dev, bs.data(), 1e-4, n1,
                                     m_xt_coeffs[1].p(), m_tmp0.p(), m_tmp1.p(), host_sanity + 2 * sanity_done);
        sanity_done = upto;
    };
    for (int i = 1; i <= N; ++i) {
        // (with the checks on, every order keeps its b_i: they are all verified in one pass after the loop)
        double* const bi = do_sanity ? m_bi_all[i].p() : m_bi.p();
        if (i == 1) {
            ScopedTimer t{this, "jacobian"};
            run_pass(PASS_GRAD, 0, nullptr);
        }
        {
            ScopedTimer t{this, "taylor_next_order"};
            if (!bias_done) run_pass(PASS_BIAS, i, nullptr);
            // (orders >= 2, single rank: remap_out drops b_i where the direct solver reads its right-hand side)
            rhs_perm = (i > 1 && !m_shard.active()) ? m_solver->rhs_perm() : nullptr;
            if (pade_riders && i >= 2) m_pade_ws.phase(m_xt_coeffs, i - 1, 1, anm_cond, true);
            {
                ScopedTimer t2{this, "remap_out"};
                be->gather_rows(m_remap_out->dev(), out_bias(), bi, rhs_perm,
                                rhs_perm ? m_solver->rhs_work() : nullptr);
            }
            // the one collective per Taylor order: sum of the per-shard nodal bias (n doubles)
            if (i > 1) allreduce(bi, n);
            if (m_inject.kind == 2 && m_inject.order == i) {
                apply_injection(bi, n);
                rhs_perm = nullptr;  // the solver takes the corrupted vector, not the copy the gather left it
            }
        }
        // Orders >= 2 queue their kernels without ever waiting for the device: t_i is formed on the
        // device from the reduction's result (next_coeff_async), lands in x_i[n] for the kernels that need
        // it and in pinned host memory for the checks below the loop.  (Order 1 needs t_1 on the host for
        // the scale factors; the sharded mode synchronises at its all-reduce anyway.)
        double ti = 0;
        const double* xbi;
        double* xi = m_xt_coeffs[i].p();
        bool pass_done = false;  // COEFF(i) + BIAS(i + 1) already queued together with next_coeff
        if (i == 1) {
            {
                ScopedTimer t{this, "build_sparse_coeff"};
                be->assemble(m_pattern->assembly(), jacobian_blocks(), m_pattern->csr().val,
                             m_pattern->has_t() ? m_grad_t_buf.p() : nullptr);
                allreduce(m_pattern->csr().val, m_pattern->nnz());
                if (m_pattern->has_t()) allreduce(m_grad_t_buf.p(), n);
                if (m_inject.kind == 3) apply_injection(m_pattern->csr().val, m_pattern->nnz());
                // sparse_solver.cpp:288-289: the coefficients must be finite (examined at the first synchronisation)
                be->count_nonfinite_async(m_pattern->nnz(), m_pattern->csr().val, host_checks);
            }
            grad_t = get_grad_t();
            {
                ScopedTimer t{this, "sparse_prep"};
                m_solver->prepare_async(host_checks + 1);
            }
            double xgt2 = 0;
            xbi = bi;  // zero at first order (anm.cpp:235)
            // Order 1 needs two reductions (|xgt|^2 for t_1, xgt . x_1 for the scale of every later t_i).  Where the
            // backend offers it they stay on the device (Backend::x1_async; NextCoeff::sc) and the host queues the
            // whole order loop behind the factorisation without waiting once: the factor's status and the Jacobian's
            // finiteness are then examined with everything else after the loop.  SANM_ORDER1_HOST=1 (and the
            // printout / profile mode that needs the scalars) takes them to the host as before -- same arithmetic.
            static const bool order1_host = std::getenv("SANM_ORDER1_HOST") != nullptr || std::getenv("SANM_X1_DOT_ANALYTIC") != nullptr;
            {
                ScopedTimer t{this, "sparse_solve"};
                m_solver->solve(grad_t, m_xgt.p());
                g_trace.mark("order 1 queued");
                if (!order1_host && !verbose && !m_force_order1_host) {
                    if (m_order1_sc.empty()) m_order1_sc = DVec{be, 4};
                    double* sc = m_order1_sc.p();
                    be->dot_async(n, m_xgt.p(), m_xgt.p(), sc + 2);
                    order1_on_device = be->x1_async(n, sc + 2, m_xgt.p(), xbi, xi, sc, m_host_scalars + 3 * i);
                    if (order1_on_device) be->dot_async(n, xi, m_xgt.p(), sc + 1);
                }
                if (!order1_on_device) {
                    xgt2 = be->dot(n, m_xgt.p(), m_xgt.p());  // (waits for the device: the factor's status is in)
                    g_trace.mark("order 1: |xgt|^2 returned");
                    // sparse_solver.cpp:288-289: the coefficients must be finite
                    sanm_check(host_checks[0] == 0, "non-finite Jacobian coefficient");
                    if (m_solver->check_prepared(host_checks + 1)) {
                        // perturbed pivots: the solver refines from now on (sparse_solver.cpp:107-127: PARDISO's
                        // behaviour); this first solution is computed again
                        m_solver->solve(grad_t, m_xgt.p());
                        xgt2 = be->dot(n, m_xgt.p(), m_xgt.p());
                    }
                }
            }
            if (!order1_on_device) {
                t1 = ti = 1.0 / std::sqrt(xgt2 + 1.0);
                // x_1 = -t1*xgt - xbi ; t_1 appended  (anm.cpp:261-264)
                be->axpby_tail(n, -ti, m_xgt.p(), -1.0, xbi, xi, ti);
                m_host_scalars[3 * i] = ti;
                // xgt . x_1 with x_1 = -t_1 xgt - 0 (the order-1 bias is exactly zero, anm.cpp:235): -t_1 |xgt|^2, the
                // reduction already on the host, instead of another launch and host round trip
                static const bool analytic = std::getenv("SANM_X1_DOT_ANALYTIC") != nullptr;
                xgt_dot_x1 = analytic ? -ti * xgt2 : be->dot(n, xi, m_xgt.p());
                g_trace.mark("order 1: xgt.x1 returned");
            }
        } else {
            // t_i = (xb_i . x_1) / (t1 - xgt . x_1);  x_i = -t_i*xgt - xb_i  (anm.cpp:246-264)
            if (pade_riders) m_pade_ws.phase(m_xt_coeffs, i - 1, 2, anm_cond, true);
            {
                ScopedTimer t{this, "sparse_solve"};
                if (rhs_perm) {  // ... and the solver's last kernel forms xb_i . x_1 on the way out
                    m_solver->solve_fused(nullptr, m_xbi.p(), m_xt_coeffs[1].p(), m_dev_scalars.p() + i);
                } else {
                    m_solver->solve(bi, m_xbi.p());
                    be->dot_async(n, m_xbi.p(), m_xt_coeffs[1].p(), m_dev_scalars.p() + i);
                }
            }
            xbi = m_xbi.p();
            if (pade_riders) m_pade_ws.phase(m_xt_coeffs, i - 1, 3, anm_cond, true);
            // next_coeff and the COEFF(i) + BIAS(i + 1) pass that consumes x_i as ONE launch where the backend offers
            // it (the pass forms x_i in its gather, rider workgroups store it: Backend::run_pass_next_coeff) and
            // nothing between the two looks at x_i; two launches otherwise -- the same arithmetic
            const NextCoeff nc{n, m_dev_scalars.p() + i, order1_on_device ? 0.0 : 1.0 / (t1 - xgt_dot_x1), m_xgt.p(), xbi, xi,
                               m_host_scalars + 3 * i, order1_on_device ? m_order1_sc.p() : nullptr};
            static const bool sanity_side_env = std::getenv("SANM_SANITY_SIDE") != nullptr;
            const bool nothing_between = !(m_inject.kind == 1 && m_inject.order == i) && !pade_side && !verbose &&
                                         !(do_sanity && sanity_side_env);
            if (fuse_passes && i < N && nothing_between && !m_pattern->has_t() && m_prog) {
                ScopedTimer t{this, "taylor_push"};
                pass_done = be->run_pass_next_coeff(m_prog->dev(), i, nc);
                if (pass_done) bias_done = true;
            }
            if (!pass_done) be->next_coeff_async(nc);
        }
        m_nr_valid_coeffs = i + 1;
        if (m_inject.kind == 1 && m_inject.order == i) apply_injection(xi, n1);
        if (pade_side) {
            be->side_fork();
            m_pade_ws.step(m_xt_coeffs, i, anm_cond);
            be->side_end();
        }
        // SANM_SANITY_SIDE: the checks of the orders finished so far, ten at a time, on the second queue beside the
        // solves that follow instead of after the loop.  Measured on armadillo_small: the tail gets 0.12 ms
        // shorter and the solves 0.26 ms longer (the pass over the matrix competes with the latency-bound level
        // kernels for the whole of its 60 us, twice): off by default.
        static const bool sanity_side = std::getenv("SANM_SANITY_SIDE") != nullptr;
        if (do_sanity && sanity_side && !pade_side && i < N && (i - sanity_done == 10 || i == N - 1)) {
            be->side_fork();
            queue_sanity(i, grad_t);
            be->side_end();
        }

        if (verbose) {
            trace_b_norm.push_back(std::sqrt(be->dot(n, bi, bi)));
            trace_x_norm.push_back(std::sqrt(be->dot(n1, xi, xi)));
            trace_t.push_back(m_host_scalars[3 * i]);  // valid: the dot above synchronised
            m_trace_xbi_norm.push_back(std::sqrt(be->dot(n, xbi, xbi)));
            if (i == 1) {  // anm.cpp:247-250: |gt|, |xgt|, SparseSolver::coeff_l2 (sparse_solver.cpp:217-223)
                m_trace_gt = std::sqrt(be->dot(n, grad_t, grad_t));
                m_trace_xgt = std::sqrt(be->dot(n, m_xgt.p(), m_xgt.p()));
                m_trace_jacob = std::sqrt(be->dot(m_pattern->nnz(), m_pattern->csr().val, m_pattern->csr().val));
            }
        }
        if (i < N && !pass_done) {
            ScopedTimer t{this, "taylor_push"};
            run_pass(fuse_passes ? PASS_COEFF_BIAS : PASS_COEFF, i, xi);
            bias_done = fuse_passes;
        }
    }
    // The checks of all orders are two passes over the matrix for ten right-hand sides each (0.12 ms on the armadillo
    // mesh) whose verdict nothing needs before the step is declared good.  SANM_SANITY_BESIDE puts them on the
    // second queue beside the Pade estimate that follows -- small kernels and host round trips that leave the
    // device idle most of the time -- and waits for them after it (a failed check is still reported before
    // anything the estimate may have had to say).  Measured: the tail gets 0.1 ms shorter, but once a second
    // hardware queue has been in use every launch of the latency-bound chains of the NEXT step takes longer
    // (solves 2.20 -> 2.39 ms per step): off by default, like the other uses of the second queue.
    static const bool sanity_beside_env = std::getenv("SANM_SANITY_BESIDE") != nullptr;
    const bool sanity_beside = do_sanity && sanity_beside_env && !pade_side && m_hp.use_pade;
    // Default since round 4: the checks are queued LAST, behind a mark in the (one) queue, and the host only waits
    // for the mark -- the scalars of the order loop, the two norms and the Pade table are in by then -- so the root
    // finder and the preparation of the first probe batch (0.08 ms of host time, profiles/r04_tail_trace.txt) run
    // while the device is busy with the checks; their verdict is read at the range estimate's first synchronisation
    // (or one of its own) and still reported first.  SANM_SANITY_FIRST=1: the checks in front, everything waited for.
    static const bool sanity_first_env = std::getenv("SANM_SANITY_FIRST") != nullptr;
    const bool sanity_behind = do_sanity && !sanity_beside && !pade_side && !sanity_first_env;
    if (do_sanity && !sanity_behind) {
        if (sanity_beside) be->side_fork();
        queue_sanity(N, grad_t);  // the orders not checked beside the loop
        if (sanity_beside) {
            be->side_end();
            be->side_detach();
        }
    }
    // the two norms of estimate_valid_range travel with the rest
    be->dot_async(n1, m_xt_coeffs[1].p(), m_xt_coeffs[1].p(), host_checks + 2);
    be->dot_async(n1, m_xt_coeffs[N].p(), m_xt_coeffs[N].p(), host_checks + 3);
    // ... and so does what is left of the Pade basis (the step of x_N) with its coefficient table: queued here,
    // before the synchronisation, instead of by PadeApproximation behind one of its own
    m_pade_ws.host_valid = false;
    if (pade_steps && !pade_side) {
        be->flush_deferred();
        for (int i = m_pade_ws.done + 1; i <= N; ++i) m_pade_ws.step(m_xt_coeffs, i, anm_cond);
        be->d2h_async(m_pade_ws.host_acoef, m_pade_ws.acoef.p(), (size_t)(N + 1) * (N + 1) * 8);
        m_pade_ws.host_valid = true;
    }
    if (sanity_behind) {
        be->mark();
        queue_sanity(N, grad_t);
    }
    g_trace.mark("loop queued");
    if (sanity_behind) be->wait_mark();
    else be->sync();
    g_trace.mark("loop-end sync returned");
    if (order1_on_device) {
        // what order 1 would have looked at before going on (sparse_solver.cpp:288-289, :107-127)
        sanm_check(host_checks[0] == 0, "non-finite Jacobian coefficient");
        if (m_solver->check_prepared(host_checks + 1)) {
            // The factorisation perturbed pivots: every solve of this expansion should have been refined (PARDISO's
            // behaviour with the reference's settings).  Rare; the expansion is simply taken again along the
            // synchronous order-1 path, which re-solves with refinement as soon as it sees the status.
            m_force_order1_host = true;
            m_inject = inject_at_entry;  // (a test's fault belongs to the expansion, not to its first attempt)
            try {
                solve_expansion_coeffs();
            } catch (...) {
                m_force_order1_host = false;
                throw;
            }
            m_force_order1_host = false;
            return;
        }
    }
    auto check_sanity = [&]() {
        for (int i = 1; i <= N && do_sanity; ++i) {
            const double ex = host_sanity[2 * (i - 1)], xdot = host_sanity[2 * (i - 1) + 1];
            sanm_check(ex < 0, "ANM check coeff eqn: order %d: excess %g", i, ex);
            if (i == 1) sanm_check(std::fabs(xdot - 1) < 1e-4, "xdot=%g", xdot);
            else sanm_check(std::fabs(xdot) < 1e-4, "i=%d: xdot=%g", i, xdot);
        }
    };
    try {
        sanm_check(host_checks[0] == 0, "non-finite Jacobian coefficient");
        for (int i = 1; i <= N; ++i) {
            const double ti = m_host_scalars[3 * i];
            // the reference asserts a finite right-hand side before solving (sparse_solver.cpp:160-161);
            // a non-finite b_i or solution makes t_i non-finite, which is checked instead of a separate pass
            if (!std::isfinite(ti))
                sanm_throw(SANM_ERR_NUMERICAL, "non-finite right-hand side / solution at order %d", i);
            m_t_coeffs.push_back(ti);
        }
        if (!sanity_beside && !sanity_behind) check_sanity();
    } catch (...) {
        be->side_wait();
        if (sanity_behind) be->sync();  // (the checks are still running on buffers of this driver)
        throw;
    }
    g_trace.mark("checks done");
    std::exception_ptr held;
    try {
        ScopedTimer t{this, "estimate_valid_range"};
        estimate_valid_range();
    } catch (...) {
        held = std::current_exception();
    }
    if (sanity_beside) {
        be->side_wait();
        check_sanity();
    }
    if (sanity_behind) {
        be->sync();  // (returns at once when the estimate has synchronised since)
        check_sanity();
    }
    g_trace.mark("range estimate done");
    if (held) std::rethrow_exception(held);
    if (verbose) {
        // the reference's printout, anm.cpp:200-203, :247-259, :295-309 (same format strings)
        std::string& o = m_verbose_text;
        o = ssprintf("=== ANM iter %zu:\n", m_iter);
        o += ssprintf("gt=%g xgt=%g jacob=%g", m_trace_gt, m_trace_xgt, m_trace_jacob);
        for (int i = 1; i <= N; ++i) o += ssprintf(" %d:(bi=%g xbi=%g)", i, trace_b_norm[i - 1], m_trace_xbi_norm[i - 1]);
        o += ssprintf("\nbound=%g t=%g\n", m_t_max_a, m_t_max);
        o += "x(a):";
        o += ssprintf(" %.3g", std::sqrt(be->dot(n1, m_xt_coeffs[0].p(), m_xt_coeffs[0].p())));
        for (int i = 1; i <= N; ++i) o += ssprintf(" %.3g", trace_x_norm[i - 1]);
        o += "\nt(a):";
        for (double t : m_t_coeffs) o += ssprintf(" %.3g,", t);
        o += "\n";
        if (m_hp.xcoeff_l2_penalty) o += ssprintf("xcoeff_l2_penalty=%g\n", m_hp.xcoeff_l2_penalty);
        if (env_verbose) {
            std::fputs(o.c_str(), stdout);
            std::fflush(stdout);
        }
    }
    ++m_iter;
}

void AnmDriver::estimate_valid_range() {
    // libsanm/anm.cpp:117-154
    const size_t n1 = m_n + 1;
    const int N = m_hp.order;
    // |x_1|, |x_N|: queued at the end of the order loop (solve_expansion_coeffs), here after its synchronisation
    const double* norms2 = m_host_scalars + 5 * ((size_t)N + 2) + 2;
    (void)n1;
    double x1 = std::sqrt(norms2[0]);
    double xback = std::max(std::sqrt(norms2[1]), 1e-15);
    double a_bound = std::pow(m_hp.maxr / xback * x1, 1.0 / double(N - 1));
    a_bound = std::min(a_bound, m_max_a_bound);
    sanm_check((int)m_t_coeffs.size() == N + 1, "t coefficients incomplete");
    sanm_check(m_t_coeffs[1] > 0, "t1=%g is not positive", m_t_coeffs[1]);
    m_t_max_a = a_bound;
    m_t_max = poly::eval(m_t_coeffs, a_bound);
    sanm_check(m_t_max > m_t_coeffs[0], "t does not incr at iter %zu: t0=%g tmax=%g bound=%g", m_iter,
               m_t_coeffs[0], m_t_max, a_bound);
    m_pade.reset();
    m_pade_diag = PadeDiag{};
    static const bool env_pade = getenv("SANM_PADE") != nullptr;
    m_be->side_join();
    if ((m_hp.use_pade || env_pade) && a_bound < m_max_a_bound) {
        auto pade = std::make_unique<PadeApproximation>(m_be, m_xt_coeffs, m_t_coeffs,
                                                        !m_hp.xcoeff_l2_penalty, &m_pade_ws);
        const bool ok = pade->estimate_valid_range(a_bound, m_hp.maxr, m_max_a_bound);
        m_pade_diag = pade->diag();
        m_pade_diag.attempted = 1;
        m_pade_diag.start = a_bound;
        if (ok) {
            m_t_max_a = pade->get_t_max_a();
            m_t_max = pade->get_t_max();
            m_pade = std::move(pade);
        }
    }
}

void AnmDriver::update_approx() {
    // libsanm/anm.cpp:156-159
    eval_xt(m_t_max_a, m_tmp0.p());
    m_be->d2d(m_xt0.p(), m_tmp0.p(), (m_n + 1) * 8);
    solve_expansion_coeffs();
}

void AnmDriver::eval_xt(double a, double* out) const {
    // libsanm/anm.cpp:166-172; unary_polynomial::eval_tensor :115-126
    if (m_pade) {
        m_pade->eval_xt(a, out);
        return;
    }
    const size_t n1 = m_n + 1;
    const int N = m_nr_valid_coeffs - 1;
    std::vector<const double*> ptrs(N + 1);
    std::vector<double> coefs(N + 1);
    double ap = 1.0;
    for (int i = 0; i <= N; ++i) {
        ptrs[i] = m_xt_coeffs[i].p();
        coefs[i] = ap;
        ap *= a;
    }
    m_be->lincomb(n1, N + 1, ptrs.data(), coefs.data(), out);
}

double AnmDriver::eval(double a, double* x_host) const {
    eval_xt(a, m_tmp1.p());
    std::vector<double> h(m_n + 1);
    m_be->d2h(h.data(), m_tmp1.p(), h.size() * 8);
    std::copy(h.begin(), h.begin() + m_n, x_host);
    return h[m_n];
}

double AnmDriver::solve_a(double t) const {
    // libsanm/anm.cpp:174-191
    if (t == m_t_max) return m_t_max_a;
    if (m_pade) return m_pade->solve_a(t);
    sanm_check(t >= m_t_coeffs[0] && t < m_t_max, "solve_a: t=%g out of [%g, %g)", t, m_t_coeffs[0],
               m_t_max);
    double l, r;
    if (m_t_max_a > 0) {
        l = 0;
        r = m_t_max_a;
    } else {
        l = -m_t_max_a;
        r = 0;
    }
    return poly::solve_eqn(m_t_coeffs, l, r, t);
}

void AnmDriver::get_xt_coeff(int i, double* dst) const {
    sanm_check(i >= 0 && i < m_nr_valid_coeffs, "coefficient %d not available", i);
    m_be->d2h(dst, m_xt_coeffs[i].p(), (m_n + 1) * 8);
}

// ---------------------------------------------------- AnmSolverVecScale --
AnmSolverVecScale::AnmSolverVecScale(Backend* be, const Graph& g, int out_var,
                                     const SparseDesc& remap_inp, const SparseDesc& remap_out,
                                     const double* x0, int64_t n, double t0, const double* v,
                                     const HyperParam& hp, bool defer_solve, const ShardInfo& shard)
        : AnmDriver(be, g, out_var, remap_inp, remap_out, n, hp, shard) {
    // libsanm/anm.cpp:322-341
    sanm_check(remap_inp.in_size == n, "linear map expects %ld inputs, got x0 of %ld",
               (long)remap_inp.in_size, (long)n);
    sanm_check(remap_out.out_size == n, "currently we assume the system is a full-rank mapping");
    m_v = DVec{be, (size_t)n};
    if (v) be->h2d(m_v.p(), v, n * 8);
    if (!defer_solve) {
        init_xt0(x0, t0);
        solve_expansion_coeffs();
    }
}

void AnmSolverVecScale::check_t0v_match(const double* fx_dev) {
    // libsanm/anm.cpp:343-360
    double ex = m_be->t0v_excess(m_n, fx_dev, m_v.p(), get_t0(), m_hp.solution_check_tol);
    if (!(ex <= 0)) {
        sanm_throw(SANM_ERR_NUMERICAL, "f(x0)+t0*v is not zero: excess=%g iter=%zu", ex, m_iter);
    }
}

bool AnmSolverVecScale::on_fx0_computed(const double* fx_dev) {
    check_t0v_match(fx_dev);
    return true;
}

// --------------------------------------------------------- AnmEqnSolver --
AnmEqnSolver::AnmEqnSolver(Backend* be, const Graph& g, int out_var, const SparseDesc& remap_inp,
                           const SparseDesc& remap_out, const double* x0, const double* y,
                           int64_t n, const HyperParam& hp, const ShardInfo& shard)
        : AnmSolverVecScale(be, g, out_var, remap_inp, remap_out, x0, n, 0, nullptr, hp, true, shard),
          m_converge_rms{hp.converge_rms} {
    // libsanm/anm.cpp:446-462: f(x) - f(x0) + t*(y + f(x0)) = 0, t from 0
    SetupLaps dl("eqn solver");
    init_xt0(x0, 0);
    dl.lap("init_xt0");
    m_eqn_y = DVec{be, (size_t)n};
    be->h2d(m_eqn_y.p(), y, n * 8);
    dl.lap("y");
    const auto t0 = std::chrono::steady_clock::now();
    solve_expansion_coeffs();
    // (the first expansion belongs to the constructor like in the reference, anm.cpp:446-462: queued, not waited for)
    m_setup.emplace_back("first_expansion", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    if (std::getenv("SANM_DEBUG_SETUP"))
        std::fprintf(stderr, "[setup] constructor, first member to last statement: %.4f\n",
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - m_ctor_begin).count());
}

AnmEqnSolver& AnmEqnSolver::next_iter() {
    // libsanm/anm.cpp:464-478
    if (m_converged) return *this;
    g_trace.mark("next_iter");
    double a = get_t_upper() >= 1 ? solve_a(1) : get_t_max_a();
    g_trace.mark("restart parameter");
    eval_xt(a, m_tmp0.p());
    m_be->d2d(m_xt0.p(), m_tmp0.p(), (m_n + 1) * 8);
    g_trace.mark("restart point queued");
    m_be->zero(m_xt0.p() + m_n, 8);  // set t0 to 0 (queued, like everything before the first check of the step)
    m_t0_host = 0;
    m_t0_known = true;
    solve_expansion_coeffs();
    return *this;
}

bool AnmEqnSolver::on_fx0_computed(const double* fx_dev) {
    // libsanm/anm.cpp:480-491
    if (m_converged) return false;
    m_be->axpby(m_n, 1.0, fx_dev, 1.0, m_eqn_y.p(), m_v.p());
    g_trace.mark("order 0 queued");
    m_residual_rms = std::sqrt(m_be->dot(m_n, m_v.p(), m_v.p()) / double(m_n));
    g_trace.mark("residual returned");
    if (m_residual_rms < m_converge_rms) {
        m_converged = true;
        return false;
    }
    return true;
}

int AnmEqnSolver::run_steps(int count, const double* x0) {
    // `count` completed expansions without going back to the caller in between: next_iter while the solve has
    // not converged, a new solve from x0 when it has (the bench's step loop; a caller that only wants the
    // solution loops on next_iter itself)
    int solves = 0;
    while (count > 0) {
        const size_t before = m_iter;
        if (!m_converged) next_iter();
        if (m_iter == before) {  // converged (next_iter only evaluated f(x0)) or was already: start over
            sanm_check(x0, "run_steps: the solve has converged and no restart point was given");
            ++solves;
            restart(x0);
        }
        count -= (int)(m_iter - before);
    }
    return solves;
}

void AnmEqnSolver::restart(const double* x0) {
    m_converged = false;
    m_residual_rms = 0;
    init_xt0(x0, 0);
    solve_expansion_coeffs();
}

void AnmEqnSolver::get_x(double* x_host) const { m_be->d2h(x_host, m_xt0.p(), m_n * 8); }

// ---------------------------------------------------- AnmImplicitSolver --
AnmImplicitSolver::AnmImplicitSolver(Backend* be, const Graph& g, int out_var,
                                     const SparseDesc& remap_inp, const SparseDesc& remap_out,
                                     const double* x0, int64_t n, double t0, const HyperParam& hp)
        : AnmDriver(be, g, out_var, remap_inp, remap_out, n, hp) {
    // libsanm/anm.cpp:494-508
    sanm_check(remap_inp.in_size == n + 1 && remap_out.out_size == n,
               "implicit solver needs remap_inp with n+1 inputs and remap_out with n outputs");
    m_fx0_first = DVec{be, (size_t)n};
    init_xt0(x0, t0);
    solve_expansion_coeffs();
}

const double* AnmImplicitSolver::get_grad_t() { return m_grad_t_buf.p(); }

bool AnmImplicitSolver::on_fx0_computed(const double* fx_dev) {
    // libsanm/anm.cpp:510-518
    if (!m_has_fx0) {
        m_be->d2d(m_fx0_first.p(), fx_dev, m_n * 8);
        m_has_fx0 = true;
    } else {
        double ex = m_be->allclose_excess(m_n, m_fx0_first.p(), fx_dev, m_hp.solution_check_tol);
        sanm_check(ex < 0, "check f(x0, t0)=f(x, t): excess %g", ex);
    }
    return true;
}

void AnmImplicitSolver::get_fx0(double* dst) const { m_be->d2h(dst, m_fx0_first.p(), m_n * 8); }

}  // namespace sanm_hip
