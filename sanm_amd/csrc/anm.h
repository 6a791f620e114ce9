// ANM continuation drivers on device-resident state.
//
// Mirrors libsanm/anm.h:96-305 (ANMDriverHelper, ANMSolverVecScale,
// ANMEqnSolver, ANMImplicitSolver) with the same member names and the same
// arithmetic; what differs is where the data lives.  The reference drives
// host tensors through ParallelTaylorCoeffProp worker threads and PARDISO;
// here every vector of the per-order loop (x_k, b_k, the Jacobian CSR, the
// Taylor state) stays in HBM and only scalars (t_k, norms, convergence flags)
// come back to the host.
#pragma once
#include <chrono>
#include <map>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "multifrontal.h"
#include "backend.h"
#include "graph.h"
#include "sparse.h"
#include "vecprog_host.h"

namespace sanm_hip {

struct HyperParam {  // libsanm/anm.h:100-114, :247-251
    int use_pade = 0;
    int sanity_check = 1;
    int order = 8;
    double maxr = 1e-6;
    double solution_check_tol = 1e-4;
    double xcoeff_l2_penalty = 0;
    double converge_rms = 1e-5;
    // device linear solver (no counterpart in the reference, which uses a
    // direct LU: libsanm/sparse_solver.cpp:107-127)
    double solver_rtol = 1e-12;
    int solver_maxit = 100000;
    int solver_kind = 1;  // 0: Jacobi-PCG, 1: multifrontal LU (direct)
    int profile = 0;      // synchronise + time every phase
    int solver_refine = 0;  // direct solver: refinement steps per solve (0: only after perturbed pivots)
};

//! tet-sharded execution over several ranks (one process per GPU).  The reference's
//! counterpart is ParallelTaylorCoeffProp (libsanm/symbolic.cpp:306-590): worker w
//! owns tets [w*T/nr, (w+1)*T/nr) and the partial results are gathered; here every
//! rank owns such a range and the assembled vectors (b_k, f(x0), Jacobian values) are
//! summed with one all-reduce each -- RCCL over xGMI, supplied by the caller.
struct ShardInfo {
    int rank = 0, world = 1;
    //! in-place sum over all ranks of `count` doubles at device pointer `buf`; 0 = ok.  nullptr: the backend's
    //! own collective (RCCL on the solver's stream, Backend::allreduce_sum; no host synchronisation)
    int (*allreduce)(void* user, double* buf, int64_t count) = nullptr;
    void* user = nullptr;
    //! set by the sharded constructors: the sharded code path runs also with world == 1 (tests)
    bool enabled = false;
    bool active() const { return enabled; }
};

class DVec {
public:
    DVec() = default;
    DVec(Backend* be, size_t n) : m_be{be}, m_n{n} { m_p = static_cast<double*>(be->alloc(n * 8)); }
    DVec(DVec&& o) noexcept : m_be{o.m_be}, m_p{o.m_p}, m_n{o.m_n} { o.m_p = nullptr; }
    DVec& operator=(DVec&& o) noexcept {
        if (this != &o) {
            reset();
            m_be = o.m_be; m_p = o.m_p; m_n = o.m_n;
            o.m_p = nullptr;
        }
        return *this;
    }
    DVec(const DVec&) = delete;
    ~DVec() { reset(); }
    void reset() {
        if (m_p) m_be->free(m_p);
        m_p = nullptr;
    }
    double* p() const { return m_p; }
    size_t size() const { return m_n; }
    bool empty() const { return !m_p; }

private:
    Backend* m_be = nullptr;
    double* m_p = nullptr;
    size_t m_n = 0;
};

//! libsanm/sparse_solver.h:17-87 on a fixed device CSR pattern
class LinearSolver {
public:
    virtual ~LinearSolver() = default;
    virtual void prepare() = 0;                       // after new values were assembled
    //! the same without waiting for the device: a failure (singular matrix) is reported by check_prepared(),
    //! to be called after the next synchronisation; `status`: a double the host can read then
    virtual void prepare_async(double* status) {
        (void)status;
        prepare();
    }
    //! true: the factorisation needed help (perturbed pivots) and the solver has switched to iterative
    //! refinement -- solutions obtained since prepare_async are to be computed again
    virtual bool check_prepared(const double* status) {
        (void)status;
        return false;
    }
    virtual void solve(const double* b, double* x) = 0;
    //! Where the solver wants its right-hand side, if the caller can put it there while producing it: entry i
    //! at rhs_work()[rhs_perm()[i]] (null: no such place).  solve_fused(nullptr, ...) then solves for that
    //! right-hand side; with dot_y it also leaves x . dot_y in *dot_out (device memory).
    virtual const int32_t* rhs_perm() const { return nullptr; }
    virtual double* rhs_work() const { return nullptr; }
    virtual void solve_fused(const double*, double*, const double*, double*) {
        sanm_throw(SANM_ERR_ASSERT, "solve_fused: not offered by this solver (rhs_perm() is null)");
    }
    int64_t nr_solve = 0, tot_iters = 0, last_iters = 0, nr_perturbed_pivots = 0;
    double last_relres = 0;
    // direct solver analysis (0 for iterative solvers)
    int64_t nnz_factors = 0, nr_front = 0, nr_level = 0, max_front = 0;
    double factor_flops = 0;
    // distributed direct solver (MfSchedule::Dist): what THIS rank factors -- its subtrees (own) and its fronts of the
    // top (top_own) --, the whole top, and the critical path of the factorisation (sum over the stages of the busiest
    // rank's flops)
    double factor_flops_own = 0, factor_flops_top = 0, factor_flops_top_own = 0, factor_flops_critical = 0;
    int64_t nr_subtree = 0, nr_subtree_own = 0, nr_dist_stage = 0, front_store_doubles = 0;
    int64_t dist_schur_doubles = 0, dist_inbox_doubles = 0;
};
//! sum over the ranks of `count` doubles at a device pointer, in place (the driver's all-reduce: RCCL on the solver's
//! stream or the C ABI's callback)
using Collective = std::function<void(double*, int64_t)>;
//! grouped point-to-point transfers / broadcasts of ranges of one device buffer (MfSchedule::Xfer; Backend::
//! comm_exchange on the backend's own communicator); empty: the communicator at hand offers the all-reduce only
using PointToPoint = std::function<void(double*, const MfSchedule::Xfer*, int)>;
//! a process-wide point-to-point callback that solvers under construction pick up (sanm_hip_test.h, sanm_test_set_p2p)
struct TestP2p {
    int (*fn)(void* user, double* base, const void* xfers, int n) = nullptr;
    void* user = nullptr;
};
TestP2p& test_p2p();
std::unique_ptr<LinearSolver> make_pcg_solver(Backend* be, const JacobianPattern& pat,
                                              const HyperParam& hp);
//! multifrontal LU (multifrontal.h); coords: (n,3) ordering hint or null
//! world > 1: the factorisation and the solves distributed by subtrees over the ranks (multifrontal.h), `coll` for
//! the exchanges
//! analysed: the analysis of pat's pattern done beforehand (Multifrontal with defer_device, finished); null: done here
std::unique_ptr<LinearSolver> make_direct_solver(Backend* be, const JacobianPattern& pat,
                                                 const HyperParam& hp, const double* coords, int rank = 0,
                                                 int world = 1, Collective coll = {}, PointToPoint p2p = {},
                                                 std::unique_ptr<Multifrontal> analysed = nullptr);

//! dense LU with partial pivoting: the small general systems of graphs on the vector interpreter
std::unique_ptr<LinearSolver> make_dense_solver(Backend* be, const JacobianPattern& pat);

//! libsanm/pade.h on device vectors
//! buffers of the Pade basis sweep that live as long as the driver, so that the sweep (a launch-bound
//! sequence of ~80 small kernels with the same arguments every continuation step) can be replayed as a graph
struct PadeWorkspace {
    Backend* be = nullptr;
    std::vector<DVec> orth;
    DVec acoef;
    DVec tmp_row;  // projections of the re-orthogonalisation pass (SANM_PADE_ORTH=cgs2)
    void* graph = nullptr;
    //! Gram-Schmidt steps 1 .. done of the current series are queued (the driver queues step i on the side
    //! queue as soon as x_i exists: PadeApproximation then finds the basis ready); -1: steps not in use
    int done = 0;
    bool done_anm_cond = false;
    //! the coefficient table on the host (pinned): the driver queues the copy behind the last step, before the
    //! synchronisation the end of the order loop needs anyway; PadeApproximation then reads it without another
    double* host_acoef = nullptr;
    bool host_valid = false;
    ~PadeWorkspace() {
        if (graph) be->graph_destroy(graph);
        if (host_acoef) be->free_host(host_acoef);
    }
    void ensure(Backend* be_, int nx, size_t len);
    //! one classical Gram-Schmidt step (pade.cpp:36-70) for xs[i], i = done + 1; the last step also completes
    //! the normalisation of its own vector
    void step(const std::vector<DVec>& xs, int i, bool anm_cond);
    //! the same in its three phases (1 projections, 2 update, 3 scaling): step(i) == phase(i, 1..3, false);
    //! with `defer` the backend may run the phase inside one of its next launches (Backend::defer_gs_phase)
    void phase(const std::vector<DVec>& xs, int i, int k, bool anm_cond, bool defer);
};

//! What decided the last Pade range estimate (pade.cpp:107-173): every discrete decision with the quantity it was
//! taken on, so that two implementations can be compared decision by decision (sanm_anm_pade_diag).
struct PadeDiag {
    int attempted = 0;    //!< use_pade && a_bound < stable_x_range (anm.cpp:143-152)
    int built = 0;        //!< the constructor produced a denominator (pade.cpp:18-20, :49-53)
    int roots_valid = 0;  //!< unary_polynomial::roots returned a value (pade.cpp:113-116)
    int accepted = 0;
    double start = 0, pole = 0, t_max_a = 0;
    std::vector<double> d;  //!< denominator coefficients m_d, low order first
    //! check(a) probes in the order the reference would make them (pade.cpp:129-165): a, margin, outcome, where
    //! margin = |pn_lo * denom_n / denom_lo - pn|^2 / (eps^2 |pn|^2); the probe passes iff margin <= 1
    struct Probe {
        double a, margin;
        int ok;
    };
    std::vector<Probe> probes;
};

class PadeApproximation {
public:
    const PadeDiag& diag() const { return m_diag; }
    //! t_coeffs: host copy of the last element of every xs[i]
    PadeApproximation(Backend* be, const std::vector<DVec>& xs, const std::vector<double>& t_coeffs,
                      bool anm_cond, PadeWorkspace* ws = nullptr);
    bool estimate_valid_range(double start, double eps, double limit);
    double get_t_max() const { return m_t_max; }
    double get_t_max_a() const { return m_t_max_a; }
    double solve_a(double t) const;
    void eval_xt(double a, double* out) const;
    double eval_t(double a) const;
    bool valid() const { return !m_d.empty(); }

private:
    Backend* m_be;
    const std::vector<DVec>& m_xs;
    size_t m_len;
    std::vector<double> m_d, m_d_lo, m_t_nume;
    double m_t0 = 0, m_t_max = 0, m_t_max_a = 0;
    PadeDiag m_diag;
    std::vector<double> m_probe_margin;  // of the last batch of probes
    void eval_nume(double a, const double* d, int n, double* out) const;
    void nume_coefs(double a, const double* d, int n, double* coefs) const;
};

class AnmDriver {  // ANMDriverHelper, libsanm/anm.h:96-207
public:
    AnmDriver(Backend* be, const Graph& g, int out_var, const SparseDesc& remap_inp,
              const SparseDesc& remap_out, int64_t nr_unknown, const HyperParam& hp,
              const ShardInfo& shard = {});
    virtual ~AnmDriver();

    void update_approx();
    double get_t_upper() const { return m_t_max; }
    double get_t_max_a() const { return m_t_max_a; }
    double solve_a(double t) const;
    //! x(a) (n doubles, host) and t(a)
    double eval(double a, double* x_host) const;
    size_t get_nr_iter() const { return m_iter; }
    int64_t nr_unknown() const { return m_n; }
    int order() const { return m_hp.order; }
    //! copy coefficient i of [x(a); t(a)] to the host (n+1 doubles)
    void get_xt_coeff(int i, double* dst) const;
    int nr_valid_xt_coeffs() const { return m_nr_valid_coeffs; }
    bool has_pade() const { return (bool)m_pade; }
    //! decisions of the last range estimate
    const PadeDiag& pade_diag() const { return m_pade_diag; }

    //! seconds per tag (HyperParam::profile != 0); with event timing this waits for the device first
    const std::map<std::string, double>& profile();
    const std::map<std::string, double>& profile_counts() const { return m_profile_cnt; }
    //! kernel launches queued inside each tag's brackets (nested tags count in their parents too)
    const std::map<std::string, double>& profile_launches() const { return m_profile_launches; }
    //! 0: off, 1: host clock around synchronised phases, 2: device events (no synchronisation)
    void set_profile_mode(int mode) { m_profile_mode = mode; }
    void clear_profile() {
        m_profile.clear();
        m_profile_cnt.clear();
        m_profile_launches.clear();
    }
    const LinearSolver& linear_solver() const { return *m_solver; }
    const JacobianPattern& pattern() const { return *m_pattern; }
    //! host-clock seconds of the constructor's phases, in order (what the reference's time_solve contains beside the
    //! continuation steps, fea/main.cpp:382, :418-425): "tet_order", "program", "jit" (+ "jit_compiled" /
    //! "jit_disk_hit" / "jit_memory_hit": 1 for the source the code object came from), "remap_tables", "pattern"
    //! (symbolic product remap_out J remap_in), "analysis" (ordering + symbolic factorisation of the direct solver,
    //! its device tables and the work vectors)
    const std::vector<std::pair<std::string, double>>& setup_profile() const { return m_setup; }
    //! the per-tet program of a (T,3,3) graph; graphs on the vector interpreter have none
    Program& program() {
        if (!m_prog) sanm_throw(SANM_ERR_UNSUPPORTED, "this solver runs its graph on the vector interpreter");
        return *m_prog;
    }
    bool on_vector_interpreter() const { return (bool)m_vprog; }
    int64_t batch() const;
    size_t arena_bytes() const;
    Backend* backend() const { return m_be; }
    const double* last_xt_coeff_dev(int i) const { return m_xt_coeffs[i].p(); }
    double* scratch_dev(int i) const { return i == 0 ? m_tmp0.p() : m_tmp1.p(); }
    //! Test hook: corrupt one entry during the next solve_expansion_coeffs so that the deferred checks can be seen
    //! to fire.  kind 1: x_order[index], 2: b_order[index] (before its solve), 3: Jacobian value [index] (after
    //! the assembly); the entry is multiplied by `value` if `scale`, else replaced by it.
    struct Injection {
        int kind = 0, order = 0;
        int64_t index = 0;
        double value = 0;
        bool scale = false;
    };
    void set_injection(const Injection& inj) { m_inject = inj; }
    //! per-order records of the last solve_expansion_coeffs (|b_k|, |x_k|, t_k)
    std::vector<double> trace_b_norm, trace_x_norm, trace_t;
    //! what the reference prints per expansion under SANM_VERBOSE (anm.cpp:200-203, :247-259, :295-309), built when
    //! profile mode 1 or SANM_VERBOSE is on
    const std::string& verbose_text() const { return m_verbose_text; }

protected:
    Backend* m_be;
    const HyperParam m_hp;
    const int64_t m_n;
    const double m_max_a_bound;
    const ShardInfo m_shard;
    int m_profile_mode = 0;
    Injection m_inject;
    void apply_injection(double* vec, int64_t len);
    void allreduce(double* buf, int64_t count);
    void exchange_p2p(double* base, const MfSchedule::Xfer* x, int n);
    std::unique_ptr<Program> m_prog;
    // Graphs over vectors or matrices of other sizes than 3 x 3 (vecprog.h): the same order loop with the vector
    // interpreter as its pass engine -- remap_inp as a device gather in front of it, the Jacobian's blocks
    // (odim x idim per batch item) assembled through the same JacobianPattern.
    std::unique_ptr<VecProgram> m_vprog;
    std::unique_ptr<DeviceRows> m_vec_remap_in;
    DVec m_vec_xin;
    void run_pass(int mode, int order, const double* x);
    const double* out_value0() const;       // graph output as remap_out gathers it: order-0 value ...
    const double* out_bias() const;         // ... and order-k bias, [B][odim]
    const double* jacobian_blocks() const;  // [B][odim][idim]
    double* pow_flag_words() const;         // the two raise-only error words of the order-0 pass; nullptr: none
    std::string pow_exponent_list() const;
    void construct_on_vector_interpreter(const Graph& g, int out_var, const SparseDesc& remap_inp,
                                         const SparseDesc& remap_out);
    void construct_solver_and_vectors(const double* coords, std::unique_ptr<Multifrontal> analysed = nullptr);
    std::unique_ptr<DeviceRows> m_remap_out;
    std::unique_ptr<JacobianPattern> m_pattern;
    std::unique_ptr<LinearSolver> m_solver;

    DVec m_xt0;
    double m_t0_host = 0;
    bool m_t0_known = false;
    size_t m_iter = 0;
    std::vector<DVec> m_xt_coeffs;
    int m_nr_valid_coeffs = 0;
    std::vector<double> m_t_coeffs;
    double m_t_max = 0, m_t_max_a = 0;
    std::unique_ptr<PadeApproximation> m_pade;
    PadeDiag m_pade_diag;
    std::string m_verbose_text;
    std::vector<double> m_trace_xbi_norm;
    double m_trace_gt = 0, m_trace_xgt = 0, m_trace_jacob = 0;
    DVec m_fx0, m_bi, m_xbi, m_xgt, m_grad_t_buf, m_tmp0, m_tmp1;
    std::vector<DVec> m_bi_all;  // b_i of every order, kept for the checks after the order loop (sanity_check)
    PadeWorkspace m_pade_ws;
    DVec m_dev_scalars;                // per order: xb_i . x_1 (consumed on the device)
    DVec m_order1_sc;                  // {t_1, xgt . x_1, |xgt|^2}: order 1's scalars when they stay on the device
    bool m_force_order1_host = false;  // an expansion taken again after perturbed pivots
    double* m_host_scalars = nullptr;  // pinned, per order: t_i, sanity excess, sanity x-dot
    std::map<std::string, double> m_profile, m_profile_cnt, m_profile_launches;
    std::vector<std::pair<std::string, double>> m_setup;
    std::chrono::steady_clock::time_point m_ctor_begin = std::chrono::steady_clock::now();

    void init_xt0(const double* x_host, double t);
    void solve_expansion_coeffs();
    void estimate_valid_range();
    void eval_xt(double a, double* out_dev) const;
    double get_t0() const { return m_t_coeffs[0]; }

    //! device pointer of dF/dt (n doubles)
    virtual const double* get_grad_t() = 0;
    //! called with f(x0) on the device; false stops the expansion
    virtual bool on_fx0_computed(const double* fx_dev) = 0;

    class ScopedTimer;
};

class AnmSolverVecScale : public AnmDriver {  // libsanm/anm.h:209-243
public:
    AnmSolverVecScale(Backend* be, const Graph& g, int out_var, const SparseDesc& remap_inp,
                      const SparseDesc& remap_out, const double* x0, int64_t n, double t0,
                      const double* v, const HyperParam& hp, bool defer_solve = false,
                      const ShardInfo& shard = {});

protected:
    DVec m_v;
    const double* get_grad_t() override { return m_v.p(); }
    bool on_fx0_computed(const double* fx_dev) override;
    void check_t0v_match(const double* fx_dev);
};

class AnmEqnSolver final : public AnmSolverVecScale {  // libsanm/anm.h:245-283
public:
    AnmEqnSolver(Backend* be, const Graph& g, int out_var, const SparseDesc& remap_inp,
                 const SparseDesc& remap_out, const double* x0, const double* y, int64_t n,
                 const HyperParam& hp, const ShardInfo& shard = {});
    double residual_rms() const { return m_residual_rms; }
    bool converged() const { return m_converged; }
    AnmEqnSolver& next_iter();
    //! start a new solve from x0 on the same model (a new ANMEqnSolver in the
    //! reference; here the device program, CSR pattern and solver analysis are kept)
    void restart(const double* x0);
    //! exactly `count` more completed expansions (restarting from x0 whenever the solve converges); returns the
    //! number of restarts
    int run_steps(int count, const double* x0);
    void get_x(double* x_host) const;

private:
    const double m_converge_rms;
    bool m_converged = false;
    DVec m_eqn_y;
    double m_residual_rms = 0;
    bool on_fx0_computed(const double* fx_dev) override;
};

class AnmImplicitSolver final : public AnmDriver {  // libsanm/anm.h:285-305
public:
    AnmImplicitSolver(Backend* be, const Graph& g, int out_var, const SparseDesc& remap_inp,
                      const SparseDesc& remap_out, const double* x0, int64_t n, double t0,
                      const HyperParam& hp);
    void get_fx0(double* dst) const;

protected:
    DVec m_fx0_first, m_grad_t;
    bool m_has_fx0 = false;
    const double* get_grad_t() override;
    bool on_fx0_computed(const double* fx_dev) override;
};

}  // namespace sanm_hip
