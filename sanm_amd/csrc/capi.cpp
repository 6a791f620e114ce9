// extern "C" entry points declared in include/sanm_hip.h.
#include "../../include/sanm_hip.h"
#include "../../include/sanm_hip_test.h"
#include "vecprog_host.h"

#include <cstring>
#include <memory>
#include <string>

#include "anm.h"
#include "tet_ops.h"
#include "fea.h"
#include "graph.h"
#include "multifrontal.h"
#include "poly.h"
#include "sparse.h"

using namespace sanm_hip;

struct sanm_graph {
    Graph g;
};
struct sanm_sparse_desc {
    SparseDesc d;
};
struct sanm_fea_model {
    ElasticForceModel m;
    sanm_graph graph_view;  // unused; graph accessed through cast below
    sanm_sparse_desc inp, out;
};

namespace {
thread_local std::string g_last_error;
std::unique_ptr<Backend> g_backend;

Backend* backend() {
    if (!g_backend) sanm_throw(SANM_ERR_ASSERT, "sanm_hip_init() has not been called");
    return g_backend.get();
}

template <class F>
int guard(F&& f) {
    try {
        f();
        return SANM_HIP_OK;
    } catch (const SanmError& e) {
        g_last_error = e.msg;
        return e.code;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return SANM_HIP_ERR_UNKNOWN;
    } catch (...) {
        g_last_error = "unknown exception";
        return SANM_HIP_ERR_UNKNOWN;
    }
}

HyperParam to_hp(const sanm_hyper_param* h) {
    HyperParam r;
    r.use_pade = h->use_pade;
    r.sanity_check = h->sanity_check;
    r.order = h->order;
    r.maxr = h->maxr;
    r.solution_check_tol = h->solution_check_tol;
    r.xcoeff_l2_penalty = h->xcoeff_l2_penalty;
    r.converge_rms = h->converge_rms;
    r.solver_rtol = h->solver_rtol;
    r.solver_maxit = h->solver_maxit;
    r.solver_kind = h->solver_kind;
    r.profile = h->profile;
    r.solver_refine = h->solver_refine;
    return r;
}
}  // namespace

// TaylorCoeffProp on the device (libsanm/symbolic.cpp:142-304)
struct sanm_taylor_prop {
    std::unique_ptr<Program> prog;
    // graphs over vectors (Slice / Concat, sizes other than 1 / 3 / 9): the vector interpreter (vecprog.h) and the
    // remap_inp rows it is fed through
    std::unique_ptr<VecProgram> vec;
    SparseDesc vec_remap;
    DVec xin;
    DVec x;
    int order = 0;
    bool xi_known = false, jacobian_done = false;
    int64_t n_in = 0;

    void ensure_jacobian() {
        if (jacobian_done) return;
        sanm_check(order == 0, "jacobian must be taken at order 0");
        if (vec) backend()->run_vec_pass(vec->dev(), PASS_GRAD, 0, nullptr);
        else backend()->run_pass(prog->dev(), PASS_GRAD, 0, nullptr);
        jacobian_done = true;
    }
    int max_order() const { return vec ? vec->max_order() : prog->max_order(); }
};

struct sanm_anm_solver {
    std::unique_ptr<AnmDriver> drv;
    AnmEqnSolver* eqn = nullptr;
    std::vector<std::string> tag_storage;
};

extern "C" {

static int g_device = -1;
int sanm_hip_init(int device) {
    return guard([&] {
        if (!g_backend) {
            g_backend.reset(make_backend(device));
            g_device = device;
        } else if (device != g_device) {
            sanm_throw(SANM_ERR_ASSERT, "sanm_hip_init(%d): this process is bound to device %d (one process per GPU)",
                       device, g_device);
        }
    });
}
const char* sanm_hip_last_error(void) { return g_last_error.c_str(); }
const char* sanm_hip_backend_name(void) { return g_backend ? g_backend->name() : "uninitialised"; }
int sanm_hip_abi_version(void) { return SANM_HIP_ABI_VERSION; }

// ---- graph ---------------------------------------------------------------
int sanm_graph_create(sanm_graph** g) {
    return guard([&] { *g = new sanm_graph; });
}
void sanm_graph_destroy(sanm_graph* g) { delete g; }
int sanm_graph_placeholder(sanm_graph* g, int* var) {
    return guard([&] { *var = g->g.placeholder(); });
}
int sanm_graph_placeholder_vector(sanm_graph* g, int size, int* var) {
    return guard([&] { *var = g->g.placeholder_vector(size); });
}
int sanm_graph_placeholder_matrix(sanm_graph* g, int rows, int cols, int* var) {
    return guard([&] { *var = g->g.placeholder_matrix(rows, cols); });
}
int sanm_graph_constant_matrix(sanm_graph* g, const double* val, int64_t batch, int rows, int cols, int* var) {
    return guard([&] { *var = g->g.constant_matrix(val, batch, rows, cols); });
}
int sanm_graph_slice(sanm_graph* g, int x, int axis, int has_begin, int begin, int has_end, int end, int stride,
                     int* var) {
    return guard([&] { *var = g->g.slice(x, axis, has_begin, begin, has_end, end, stride); });
}
int sanm_graph_concat(sanm_graph* g, int n, const int* vars, int axis, int* var) {
    return guard([&] { *var = g->g.concat(n, vars, axis); });
}
int sanm_graph_constant(sanm_graph* g, const double* val, int64_t batch, int size, int* var) {
    return guard([&] { *var = g->g.constant(val, batch, size); });
}
int sanm_graph_linear_combine(sanm_graph* g, int n, const double* coeffs, const int* vars,
                              double bias, int* var) {
    return guard([&] { *var = g->g.linear_combine(n, coeffs, vars, bias); });
}
int sanm_graph_multiply(sanm_graph* g, int a, int b, int* var) {
    return guard([&] { *var = g->g.multiply(a, b); });
}
int sanm_graph_pow(sanm_graph* g, int x, double e, int* var) {
    return guard([&] { *var = g->g.pow(x, e); });
}
int sanm_graph_log(sanm_graph* g, int x, int* var) {
    return guard([&] { *var = g->g.log(x); });
}
int sanm_graph_reduce_sum(sanm_graph* g, int x, int axis, int* var) {
    return guard([&] { *var = g->g.reduce_sum(x, axis); });
}
int sanm_graph_batched_matmul(sanm_graph* g, int a, int b, int* var) {
    return guard([&] { *var = g->g.batched_matmul(a, b); });
}
int sanm_graph_batched_mat_inv_mul(sanm_graph* g, int x, int a, int is_left, int* var) {
    return guard([&] { *var = g->g.batched_mat_inv_mul(x, a, is_left != 0); });
}
int sanm_graph_batched_det(sanm_graph* g, int x, int* var) {
    return guard([&] { *var = g->g.batched_det(x); });
}
int sanm_graph_batched_transpose(sanm_graph* g, int x, int* var) {
    return guard([&] { *var = g->g.batched_transpose(x); });
}
int sanm_graph_batched_mul_eye(sanm_graph* g, int x, int dim, int* var) {
    return guard([&] { *var = g->g.batched_mul_eye(x, dim); });
}
int sanm_graph_batched_svd_w(sanm_graph* g, int x, int require_rotation, int usw[3]) {
    return guard([&] { g->g.batched_svd_w(x, require_rotation != 0, usw); });
}

// ---- sparse desc -----------------------------------------------------------
int sanm_sparse_desc_create(int64_t out_size, int64_t in_size, const uint64_t* rowptr,
                            const uint64_t* idx, const double* coeff, sanm_sparse_desc** d) {
    return guard([&] {
        auto p = std::make_unique<sanm_sparse_desc>();
        p->d = SparseDesc(out_size, in_size, rowptr, idx, coeff);
        *d = p.release();
    });
}
void sanm_sparse_desc_destroy(sanm_sparse_desc* d) { delete d; }
int sanm_sparse_desc_get(const sanm_sparse_desc* d, int64_t* out_size, int64_t* in_size,
                         int64_t* nnz, uint64_t* rowptr, uint64_t* idx, double* coeff) {
    return guard([&] {
        if (out_size) *out_size = d->d.out_size;
        if (in_size) *in_size = d->d.in_size;
        if (nnz) *nnz = d->d.idx.size();
        if (rowptr) std::memcpy(rowptr, d->d.rowptr.data(), d->d.rowptr.size() * 8);
        if (idx) std::memcpy(idx, d->d.idx.data(), d->d.idx.size() * 8);
        if (coeff) std::memcpy(coeff, d->d.coef.data(), d->d.coef.size() * 8);
    });
}

int sanm_sparse_desc_set_out_coords(sanm_sparse_desc* d, const double* coords) {
    return guard([&] {
        if (coords) d->d.out_coords.assign(coords, coords + d->d.out_size * 3);
        else d->d.out_coords.clear();
    });
}

// ---- direct solver (SparseSolver, libsanm/sparse_solver.h:17-87) ----------
struct sanm_direct_solver {
    std::vector<uint32_t> rowptr, col;
    std::unique_ptr<Multifrontal> mf;
    CsrDev csr{};
    DVec val, b, x;
};
int sanm_direct_solver_create(int64_t n, const uint32_t* rowptr, const uint32_t* col,
                              const double* coords, sanm_direct_solver** out) {
    return guard([&] {
        Backend* be = backend();
        auto s = std::make_unique<sanm_direct_solver>();
        s->rowptr.assign(rowptr, rowptr + n + 1);
        s->col.assign(col, col + rowptr[n]);
        // (SANM_MF_PLAN_WORLD=G: analysis as rank 0 of G -- what scripts/dist_plan.py reads back through
        // sanm_direct_solver_dist_plan; factor / solve of such a handle are not offered)
        const char* plan = std::getenv("SANM_MF_PLAN_WORLD");
        s->mf = std::make_unique<Multifrontal>(be, n, s->rowptr, s->col, coords, 0, plan ? std::max(std::atoi(plan), 1) : 1);
        void* drp = be->alloc((n + 1) * 4);
        void* dcol = be->alloc(std::max<size_t>(s->col.size(), 1) * 4);
        be->h2d(drp, s->rowptr.data(), (n + 1) * 4);
        be->h2d(dcol, s->col.data(), s->col.size() * 4);
        s->val = DVec{be, std::max<size_t>(s->col.size(), 1)};
        s->b = DVec{be, (size_t)n};
        s->x = DVec{be, (size_t)n};
        s->csr = {static_cast<uint32_t*>(drp), static_cast<uint32_t*>(dcol), s->val.p(), n,
                  (int64_t)s->col.size()};
        *out = s.release();
    });
}
int sanm_test_set_p2p(sanm_test_p2p_fn fn, void* user) {
    return guard([&] {
        test_p2p().fn = reinterpret_cast<int (*)(void*, double*, const void*, int)>(fn);
        test_p2p().user = user;
    });
}
int sanm_direct_solver_dist_plan(const sanm_direct_solver* s, int64_t cap, double* out, int64_t* n_out) {
    return guard([&] {
        const auto& D = s->mf->schedule().dist;
        sanm_check(out && n_out, "null output");
        const int G = D.enabled ? D.world : 1, S = D.enabled ? D.nr_stage : 1;
        std::vector<double> v{(double)G, (double)S, s->mf->factor_flops, D.flops_top, (double)D.nr_subtree,
                              (double)D.schur_doubles, (double)D.inbox_doubles, D.nnz_top, (double)s->mf->nnz_factors,
                              D.flops_critical, D.imbalance, 0.0};
        if (D.enabled) {
            v.insert(v.end(), D.stage_flops.begin(), D.stage_flops.end());
            v.insert(v.end(), D.stage_nnz.begin(), D.stage_nnz.end());
            for (int st = 0; st < S; ++st) {
                std::vector<double> recv(G, 0.0);
                for (const auto& x : D.schur[st].xfers) recv[x.dst] += (double)x.cnt;
                v.push_back((double)D.schur[st].doubles);
                v.push_back(*std::max_element(recv.begin(), recv.end()));
            }
            // the transfers themselves: {stage, src, dst, doubles, src_stage} each, behind their count
            size_t nx = 0;
            for (int st = 0; st < S; ++st) nx += D.schur[st].xfers.size();
            v.push_back((double)nx);
            for (int st = 0; st < S; ++st)
                for (const auto& x : D.schur[st].xfers) {
                    const double e[5] = {(double)st, (double)x.src, (double)x.dst, (double)x.cnt, (double)x.src_stage};
                    v.insert(v.end(), e, e + 5);
                }
        } else {
            v.push_back(s->mf->factor_flops);
            v.push_back((double)s->mf->nnz_factors);
            v.push_back(0.0);
            v.push_back(0.0);
            v.push_back(0.0);
        }
        *n_out = (int64_t)v.size();
        for (int64_t i = 0; i < (int64_t)v.size() && i < cap; ++i) out[i] = v[i];
    });
}
void sanm_direct_solver_destroy(sanm_direct_solver* s) {
    if (!s) return;
    if (g_backend) {
        g_backend->free(const_cast<uint32_t*>(s->csr.rowptr));
        g_backend->free(const_cast<uint32_t*>(s->csr.col));
    }
    delete s;
}
int sanm_direct_solver_factor(sanm_direct_solver* s, const double* val, int* nr_bad_pivot) {
    return guard([&] {
        Backend* be = backend();
        sanm_check(!s->mf->schedule().dist.enabled, "plan-only handle (SANM_MF_PLAN_WORLD): factor is not offered");
        be->h2d(s->val.p(), val, s->col.size() * 8);
        int bad = be->mf_factor(s->mf->dev(), s->mf->schedule(), s->csr);
        if (nr_bad_pivot) *nr_bad_pivot = bad;
    });
}
int sanm_direct_solver_solve(sanm_direct_solver* s, const double* b, double* x) {
    return guard([&] {
        Backend* be = backend();
        sanm_check(!s->mf->schedule().dist.enabled, "plan-only handle (SANM_MF_PLAN_WORLD): solve is not offered");
        be->h2d(s->b.p(), b, s->csr.n * 8);
        be->mf_solve(s->mf->dev(), s->mf->schedule(), s->b.p(), s->x.p());
        be->d2h(x, s->x.p(), s->csr.n * 8);
    });
}
int sanm_direct_solver_apply(sanm_direct_solver* s, const double* x, double* y) {
    return guard([&] {
        Backend* be = backend();
        be->h2d(s->b.p(), x, s->csr.n * 8);
        be->spmv(s->csr, s->b.p(), s->x.p());
        be->d2h(y, s->x.p(), s->csr.n * 8);
    });
}
int sanm_direct_solver_coeff_l2(sanm_direct_solver* s, double* l2) {
    return guard([&] { *l2 = std::sqrt(backend()->dot(s->csr.nnz, s->val.p(), s->val.p())); });
}
int sanm_direct_solver_stats(const sanm_direct_solver* s, int64_t* nnz_factors, double* flops,
                             int32_t* nr_front, int32_t* nr_level, int32_t* max_front,
                             int32_t* root_pivots, int32_t* nr_supervar) {
    return guard([&] {
        if (nnz_factors) *nnz_factors = s->mf->nnz_factors;
        if (flops) *flops = s->mf->factor_flops;
        if (nr_front) *nr_front = s->mf->nr_front;
        if (nr_level) *nr_level = s->mf->nr_level;
        if (max_front) *max_front = s->mf->max_front;
        if (root_pivots) *root_pivots = s->mf->root_pivots;
        if (nr_supervar) *nr_supervar = s->mf->nr_supervar;
    });
}

// ---- taylor ----------------------------------------------------------------
int sanm_taylor_create(const sanm_graph* g, int out_var, const sanm_sparse_desc* remap_inp,
                       int max_order, sanm_taylor_prop** prop) {
    return guard([&] {
        Backend* be = backend();
        if (graph_is_vector(g->g, out_var)) {
            // batched vectors: the placeholder's length comes from the graph, the batch from remap_inp
            // (the placeholder the OUTPUT depends on: a graph object may hold others that this output does not use)
            int idim = 0;
            {
                std::vector<char> need(g->g.vars.size(), 0);
                need[out_var] = 1;
                for (int oi = (int)g->g.ops.size() - 1; oi >= 0; --oi) {
                    const GraphOp& op = g->g.ops[oi];
                    bool used = false;
                    for (int o : op.out) used = used || need[o];
                    if (!used) continue;
                    for (int in : op.in) need[in] = 1;
                    if (op.type == OP_PLACEHOLDER && !op.out.empty()) idim = g->g.vars[op.out[0]].size;
                }
            }
            sanm_check(idim > 0 && remap_inp->d.out_size % idim == 0, "remap_inp must produce a (B,%d) tensor", idim);
            auto p = std::make_unique<sanm_taylor_prop>();
            p->vec = std::make_unique<VecProgram>(be, g->g, out_var, remap_inp->d.out_size / idim, max_order);
            sanm_check(p->vec->idim() == idim, "placeholder of %d elements, program input of %d", idim, p->vec->idim());
            p->vec_remap = remap_inp->d;
            p->n_in = remap_inp->d.in_size;
            p->xin = DVec{be, (size_t)remap_inp->d.out_size};
            *prop = p.release();
            return;
        }
        sanm_check(remap_inp->d.out_size % 9 == 0, "remap_inp must produce a (T,3,3) tensor");
        auto p = std::make_unique<sanm_taylor_prop>();
        p->prog = std::make_unique<Program>(be, g->g, out_var, remap_inp->d.out_size / 9, max_order);
        p->prog->set_remap_in(remap_inp->d.in_size, remap_inp->d.rowptr.data(),
                              remap_inp->d.idx.data(), remap_inp->d.coef.data());
        p->n_in = remap_inp->d.in_size;
        p->x = DVec{be, (size_t)p->n_in};
        *prop = p.release();
    });
}
void sanm_taylor_destroy(sanm_taylor_prop* p) { delete p; }

int sanm_taylor_push_xi(sanm_taylor_prop* p, const double* x, double* y_k) {
    return guard([&] {
        Backend* be = backend();
        sanm_check(!p->xi_known, "push_xi called twice for one order");
        sanm_check(p->order <= p->max_order(), "order exceeds max_order");
        if (p->vec) {
            // SparseLinearDesc::apply (anm.cpp:55-75) of a handful of entries, then the pass on the device
            const SparseDesc& R = p->vec_remap;
            std::vector<double> xin(R.out_size, 0.0);
            for (int64_t i = 0; i < R.out_size; ++i)
                for (uint64_t q = R.rowptr[i]; q < R.rowptr[i + 1]; ++q) xin[i] += R.coef[q] * x[R.idx[q]];
            be->h2d(p->xin.p(), xin.data(), xin.size() * 8);
            be->run_vec_pass(p->vec->dev(), p->order == 0 ? PASS_EVAL0 : PASS_COEFF, p->order, p->xin.p());
            be->sync();
            if (p->order == 0) {
                double fl[2];
                p->vec->take_flags(fl);
                if (fl[0] != 0) sanm_throw(SANM_ERR_NUMERICAL, "0^p when p is not integer");
                if (fl[1] != 0)
                    sanm_throw(SANM_ERR_UNSUPPORTED, "integer power (other than the square) of a series through zero on "
                                                     "the vector interpreter");
            }
            p->xi_known = true;
            if (y_k) p->vec->download_out(p->order, y_k);
            return;
        }
        be->h2d(p->x.p(), x, p->n_in * 8);
        be->run_pass(p->prog->dev(), p->order == 0 ? PASS_EVAL0 : PASS_COEFF, p->order, p->x.p());
        be->sync();
        if (p->order == 0 && !p->prog->pow_flags().empty()) {
            // 0^p (analytic_unary.cpp:112-131): flagged by the order-0 pass
            double fl[2] = {0, 0};
            const double zero[2] = {0, 0};
            double* dev = p->prog->arena_dev() + p->prog->pow_flags()[0].off;
            be->d2h(fl, dev, 16);
            if (fl[0] != 0 || fl[1] != 0) {
                be->h2d(dev, zero, 16);
                if (fl[0] == 0)
                    sanm_throw(SANM_ERR_UNSUPPORTED, "integer power of a series through zero beyond order %d",
                               POW_INT_MAX_ORDER);
                sanm_throw(SANM_ERR_NUMERICAL, "0^p when p is not integer");
            }
        }
        p->xi_known = true;
        if (y_k) {
            int T = p->prog->T();
            (void)T;
            // the output is a local var; find its graph id through download of local index
            const ProgramDev d = p->prog->dev();
            const VarDesc& vd = p->prog->vars()[d.out_var];
            const int sz = vd.size;
            std::vector<double> soa((size_t)sz * d.Tpad);
            be->d2h(soa.data(), d.arena + vd.coef + (int64_t)p->order * sz * d.Tpad, soa.size() * 8);
            for (int64_t e = 0; e < d.T; ++e)
                for (int c = 0; c < sz; ++c) y_k[e * sz + c] = soa[c * d.Tpad + e];
        }
    });
}

int sanm_taylor_compute_next_order_bias(sanm_taylor_prop* p, double* bias) {
    return guard([&] {
        Backend* be = backend();
        p->ensure_jacobian();
        sanm_check(p->xi_known, "push_xi must precede compute_next_order_bias");
        sanm_check(p->order < p->max_order(), "order exceeds max_order");
        ++p->order;
        p->xi_known = false;
        if (p->vec) {
            be->run_vec_pass(p->vec->dev(), PASS_BIAS, p->order, nullptr);
            be->sync();
            if (bias) p->vec->download_out(-1, bias);
            return;
        }
        be->run_pass(p->prog->dev(), PASS_BIAS, p->order, nullptr);
        be->sync();
        if (bias) {
            const ProgramDev d = p->prog->dev();
            const VarDesc& vd = p->prog->vars()[d.out_var];
            const int sz = vd.size;
            std::vector<double> soa((size_t)sz * d.Tpad);
            be->d2h(soa.data(), d.arena + vd.bias, soa.size() * 8);
            for (int64_t e = 0; e < d.T; ++e)
                for (int c = 0; c < sz; ++c) bias[e * sz + c] = soa[c * d.Tpad + e];
        }
    });
}

int sanm_taylor_output_size(const sanm_taylor_prop* p, int* size) {
    return guard([&] { *size = p->vec ? p->vec->odim() : p->prog->dev().odim; });
}

int sanm_taylor_get_jacobian(sanm_taylor_prop* p, double* jac) {
    return guard([&] {
        sanm_check(p->jacobian_done || p->order == 0, "jacobian must be taken at order 0");
        sanm_check(p->xi_known || p->jacobian_done, "push_xi must precede get_jacobian");
        p->ensure_jacobian();
        backend()->sync();
        if (p->vec) p->vec->download_jacobian(jac);
        else p->prog->download_jacobian(jac);
    });
}

int sanm_taylor_get_var(sanm_taylor_prop* p, int var, int order, double* dst) {
    return guard([&] {
        if (p->vec) p->vec->download_var(var, order, dst);
        else p->prog->download_var(var, order, dst);
    });
}

int sanm_taylor_reset(sanm_taylor_prop* p) {
    return guard([&] {
        p->order = 0;
        p->xi_known = false;
        p->jacobian_done = false;
    });
}

// ---- ANM -------------------------------------------------------------------
void sanm_hyper_param_default(sanm_hyper_param* hp, int eqn_solver) {
    HyperParam d;
    hp->use_pade = d.use_pade;
    hp->sanity_check = d.sanity_check;
    hp->order = d.order;
    hp->maxr = d.maxr;
    hp->solution_check_tol = d.solution_check_tol;
    hp->xcoeff_l2_penalty = d.xcoeff_l2_penalty;
    hp->converge_rms = d.converge_rms;
    hp->solver_rtol = d.solver_rtol;
    hp->solver_maxit = d.solver_maxit;
    hp->solver_kind = d.solver_kind;
    hp->profile = d.profile;
    hp->solver_refine = d.solver_refine;
    (void)eqn_solver;
}

int sanm_anm_eqn_solver_create(const sanm_graph* g, int out_var, const sanm_sparse_desc* remap_inp,
                               const sanm_sparse_desc* remap_out, const double* x0, const double* y,
                               int64_t n, const sanm_hyper_param* hp, sanm_anm_solver** s) {
    return guard([&] {
        auto p = std::make_unique<sanm_anm_solver>();
        auto* e = new AnmEqnSolver(backend(), g->g, out_var, remap_inp->d, remap_out->d, x0, y, n,
                                   to_hp(hp));
        p->drv.reset(e);
        p->eqn = e;
        *s = p.release();
    });
}
int sanm_anm_eqn_solver_create_sharded(const sanm_graph* g, int out_var,
                                       const sanm_sparse_desc* remap_inp,
                                       const sanm_sparse_desc* remap_out, const double* x0,
                                       const double* y, int64_t n, const sanm_hyper_param* hp, int rank,
                                       int world, sanm_allreduce_fn allreduce, void* user,
                                       sanm_anm_solver** s) {
    return guard([&] {
        auto p = std::make_unique<sanm_anm_solver>();
        ShardInfo sh;
        sh.rank = rank;
        sh.world = world;
        sh.allreduce = allreduce;
        sh.user = user;
        sh.enabled = true;
        auto* e = new AnmEqnSolver(backend(), g->g, out_var, remap_inp->d, remap_out->d, x0, y, n,
                                   to_hp(hp), sh);
        p->drv.reset(e);
        p->eqn = e;
        *s = p.release();
    });
}
int sanm_hip_comm_available(void) {
    try {
        return backend()->comm_available() ? 1 : 0;
    } catch (...) {
        return 0;
    }
}
int sanm_hip_comm_unique_id(void* id, size_t cap) {
    return guard([&] {
        sanm_check(cap >= 128, "the identifier needs 128 bytes");
        backend()->comm_unique_id(id);
    });
}
int sanm_hip_comm_init(int rank, int world, const void* id, size_t id_bytes) {
    return guard([&] {
        sanm_check(id_bytes >= 128 && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
        backend()->comm_init(rank, world, id);
    });
}
int sanm_hip_comm_query(int* world, int* rank) {
    return guard([&] {
        sanm_check(world && rank, "null output");
        *world = *rank = 0;
        if (g_backend) g_backend->comm_query(world, rank);
    });
}
int sanm_hip_comm_destroy(void) {
    return guard([&] {
        if (g_backend) g_backend->comm_destroy();
    });
}
int sanm_anm_vecscale_solver_create(const sanm_graph* g, int out_var,
                                    const sanm_sparse_desc* remap_inp,
                                    const sanm_sparse_desc* remap_out, const double* x0, double t0,
                                    const double* v, int64_t n, const sanm_hyper_param* hp,
                                    sanm_anm_solver** s) {
    return guard([&] {
        auto p = std::make_unique<sanm_anm_solver>();
        p->drv = std::make_unique<AnmSolverVecScale>(backend(), g->g, out_var, remap_inp->d,
                                                     remap_out->d, x0, n, t0, v, to_hp(hp));
        *s = p.release();
    });
}
int sanm_anm_implicit_solver_create(const sanm_graph* g, int out_var,
                                    const sanm_sparse_desc* remap_inp,
                                    const sanm_sparse_desc* remap_out, const double* x0, double t0,
                                    int64_t n, const sanm_hyper_param* hp, sanm_anm_solver** s) {
    return guard([&] {
        auto p = std::make_unique<sanm_anm_solver>();
        p->drv = std::make_unique<AnmImplicitSolver>(backend(), g->g, out_var, remap_inp->d,
                                                     remap_out->d, x0, n, t0, to_hp(hp));
        *s = p.release();
    });
}
void sanm_anm_solver_destroy(sanm_anm_solver* s) { delete s; }

int sanm_anm_next_iter(sanm_anm_solver* s) {
    return guard([&] {
        sanm_check(s->eqn, "next_iter is only defined for ANMEqnSolver");
        s->eqn->next_iter();
    });
}
int sanm_anm_restart(sanm_anm_solver* s, const double* x0) {
    return guard([&] {
        sanm_check(s->eqn, "restart is only defined for ANMEqnSolver");
        s->eqn->restart(x0);
    });
}
// The HIP source of the pass kernels of an fea model's graph at `order` -- it depends on the structure of the graph
// and on the order only (graph.cpp: Program::spec_source), so it is produced here from a one-cell mesh, without a
// device: sanm_amd/build.py compiles these sources ahead of time and embeds the code objects in the library.
int64_t sanm_fea_spec_source(int energy_model, int inverse, int order, char* buf, int64_t cap) {
    int64_t len = -1;
    guard([&] {
        // one cell of TetrahedralMesh::make_cuboid (fea/tetrahedral_mesh.cpp:93-204)
        const double V[24] = {0, 0, 0, 0, 0, 1, 0, 1, 0, 0, 1, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0, 1, 1, 1};
        // (vertex (x, y, z) at index (x * 2 + y) * 2 + z; the cell's corners h0..h7 in make_cuboid's order)
        const int32_t h[8] = {0, 4, 6, 2, 1, 5, 7, 3};
        const int pat[5][4] = {{0, 2, 1, 5}, {0, 4, 7, 5}, {0, 2, 5, 7}, {2, 6, 5, 7}, {0, 7, 3, 2}};
        int32_t tets[20];
        for (int t = 0; t < 5; ++t)
            for (int q = 0; q < 4; ++q) tets[t * 4 + q] = h[pat[t][q]];
        uint8_t fixed[24] = {1, 1, 1};
        ElasticForceModel m;
        const Material mat = Material::from_young_poisson(1.0, 0.3);
        if (inverse) make_inverse(m, 8, V, 5, tets, fixed, (EnergyModel)energy_model, mat);
        else make_forward(m, 8, V, 5, tets, fixed, (EnergyModel)energy_model, mat, nullptr, nullptr);
        Program prog(nullptr, m.graph, m.y, 5, order, 0, 5, /*full_history=*/false);
        const std::string src = prog.spec_source();
        len = src.size();
        if (buf && cap > 0) {
            const int64_t n = std::min<int64_t>(len, cap - 1);
            std::memcpy(buf, src.data(), n);
            buf[n] = 0;
        }
    });
    return len;
}
int64_t sanm_anm_spec_source(sanm_anm_solver* s, char* buf, int64_t cap) {
    int64_t len = -1;
    guard([&] {
        const std::string src = s->drv->program().spec_source();
        len = src.size();
        if (buf && cap > 0) {
            const int64_t n = std::min<int64_t>(len, cap - 1);
            std::memcpy(buf, src.data(), n);
            buf[n] = 0;
        }
    });
    return len;
}
int sanm_anm_time_kernel(sanm_anm_solver* s, int kernel, int reps, int mode, int order,
                         double* avg_ms) {
    return guard([&] {
        auto& d = *s->drv;
        ProgramDev P = d.program().dev();
        CsrDev A = d.pattern().csr();
        sanm_check(reps > 0 && kernel >= 0 && kernel <= 2, "bad kernel/reps");
        const double* x = d.last_xt_coeff_dev(kernel == 0 && mode == PASS_EVAL0 ? 0 : 1);
        *avg_ms = d.backend()->time_kernel(kernel, reps, &P, mode, order, &A, x, d.scratch_dev(0));
    });
}
int sanm_anm_pass_timing(sanm_anm_solver* s, int enable, double* total_ms, int64_t* count) {
    return guard([&] {
        Backend* be = s->drv->backend();
        if (total_ms && count) be->pass_timing(total_ms, count);
        be->enable_pass_timing(enable != 0);
    });
}
int sanm_anm_run_steps(sanm_anm_solver* s, int count, const double* x0, int* nr_restart) {
    return guard([&] {
        sanm_check(s->eqn, "run_steps is only defined for ANMEqnSolver");
        const int r = s->eqn->run_steps(count, x0);
        if (nr_restart) *nr_restart = r;
    });
}
int sanm_anm_update_approx(sanm_anm_solver* s) {
    return guard([&] { s->drv->update_approx(); });
}
int sanm_anm_converged(const sanm_anm_solver* s, int* flag) {
    return guard([&] {
        sanm_check(s->eqn, "converged is only defined for ANMEqnSolver");
        *flag = s->eqn->converged();
    });
}
int sanm_anm_residual_rms(const sanm_anm_solver* s, double* r) {
    return guard([&] {
        sanm_check(s->eqn, "residual_rms is only defined for ANMEqnSolver");
        *r = s->eqn->residual_rms();
    });
}
int sanm_anm_get_x(const sanm_anm_solver* s, double* x) {
    return guard([&] {
        sanm_check(s->eqn, "get_x is only defined for ANMEqnSolver");
        s->eqn->get_x(x);
    });
}
int sanm_anm_get_t_upper(const sanm_anm_solver* s, double* t) {
    return guard([&] { *t = s->drv->get_t_upper(); });
}
int sanm_anm_get_t_max_a(const sanm_anm_solver* s, double* a) {
    return guard([&] { *a = s->drv->get_t_max_a(); });
}
int sanm_anm_solve_a(const sanm_anm_solver* s, double t, double* a) {
    return guard([&] { *a = s->drv->solve_a(t); });
}
int sanm_anm_eval(const sanm_anm_solver* s, double a, double* x, double* t) {
    return guard([&] { *t = s->drv->eval(a, x); });
}
int sanm_anm_nr_iter(const sanm_anm_solver* s, int64_t* iter) {
    return guard([&] { *iter = s->drv->get_nr_iter(); });
}
int sanm_anm_nr_xt_coeffs(const sanm_anm_solver* s, int* nr) {
    return guard([&] { *nr = s->drv->nr_valid_xt_coeffs(); });
}
int sanm_anm_xt_coeff(const sanm_anm_solver* s, int i, double* xt) {
    return guard([&] { s->drv->get_xt_coeff(i, xt); });
}
int sanm_anm_has_pade(const sanm_anm_solver* s, int* flag) {
    return guard([&] { *flag = s->drv->has_pade(); });
}
int sanm_anm_get_stats_sized(const sanm_anm_solver* s, void* st, size_t st_bytes) {
    // (a caller built against an older header has a shorter record: it gets the prefix it knows)
    sanm_anm_stats full;
    std::memset(&full, 0, sizeof full);
    const int rc = sanm_anm_get_stats(s, &full);
    if (rc == 0 && st) std::memcpy(st, &full, std::min(st_bytes, sizeof full));
    return rc;
}
int sanm_anm_setup_profile(const sanm_anm_solver* s, int max_tags, const char** names, double* seconds) {
    const auto& sp = s->drv->setup_profile();
    const int k = (int)sp.size();
    for (int i = 0; i < k && i < max_tags; ++i) {
        if (names) names[i] = sp[i].first.c_str();
        if (seconds) seconds[i] = sp[i].second;
    }
    return k;
}
int sanm_anm_get_stats(const sanm_anm_solver* s, sanm_anm_stats* st) {
    return guard([&] {
        auto& d = *s->drv;
        st->nr_unknown = d.nr_unknown();
        st->nr_tet = d.batch();
        st->jacobian_nnz = d.pattern().nnz();
        st->assembly_contribs = d.pattern().nr_contrib();
        st->nr_linear_solve = d.linear_solver().nr_solve;
        st->linear_iters_total = d.linear_solver().tot_iters;
        st->linear_iters_last = d.linear_solver().last_iters;
        st->linear_relres_last = d.linear_solver().last_relres;
        st->arena_bytes = d.arena_bytes();
        st->factor_nnz = d.linear_solver().nnz_factors;
        st->factor_flops = d.linear_solver().factor_flops;
        st->nr_front = d.linear_solver().nr_front;
        st->nr_level = d.linear_solver().nr_level;
        st->max_front = d.linear_solver().max_front;
        st->factor_flops_own = d.linear_solver().factor_flops_own;
        st->factor_flops_top = d.linear_solver().factor_flops_top;
        st->nr_subtree = d.linear_solver().nr_subtree;
        st->nr_subtree_own = d.linear_solver().nr_subtree_own;
        st->dist_schur_doubles = d.linear_solver().dist_schur_doubles;
        st->dist_inbox_doubles = d.linear_solver().dist_inbox_doubles;
        st->factor_flops_top_own = d.linear_solver().factor_flops_top_own;
        st->factor_flops_critical = d.linear_solver().factor_flops_critical;
        st->nr_dist_stage = d.linear_solver().nr_dist_stage;
        st->front_store_doubles = d.linear_solver().front_store_doubles;
    });
}
int sanm_anm_debug_inject(sanm_anm_solver* s, int kind, int order, int64_t index, double value, int scale) {
    return guard([&] {
        sanm_check(kind >= 0 && kind <= 3, "injection kind %d", kind);
        AnmDriver::Injection inj;
        inj.kind = kind;
        inj.order = order;
        inj.index = index;
        inj.value = value;
        inj.scale = scale != 0;
        s->drv->set_injection(inj);
    });
}
int sanm_anm_set_profile(sanm_anm_solver* s, int mode, int clear) {
    return guard([&] {
        sanm_check(mode >= 0 && mode <= 2, "profile mode %d", mode);
        if (clear) {
            (void)s->drv->profile();  // drains pending event brackets
            s->drv->clear_profile();
        }
        s->drv->set_profile_mode(mode);
    });
}
int sanm_anm_profile_counts(const sanm_anm_solver* s, int max_tags, double* counts) {
    // (profile() reads device events: errors must not cross the C boundary; negative = error code)
    int k = 0;
    const int rc = guard([&] {
        for (auto& kv : s->drv->profile_counts()) {
            if (k < max_tags && counts) counts[k] = kv.second;
            ++k;
        }
    });
    return rc == 0 ? k : -std::abs(rc);
}
int sanm_anm_profile_launches(const sanm_anm_solver* s, int max_tags, double* launches) {
    int k = 0;
    const int rc = guard([&] {
        const auto& L = s->drv->profile_launches();
        for (auto& kv : s->drv->profile()) {  // same order as sanm_anm_profile
            auto it = L.find(kv.first);
            if (k < max_tags && launches) launches[k] = it == L.end() ? 0.0 : it->second;
            ++k;
        }
    });
    return rc == 0 ? k : -std::abs(rc);
}
int sanm_anm_profile(const sanm_anm_solver* s, int max_tags, const char** names, double* seconds) {
    auto* ms = const_cast<sanm_anm_solver*>(s);
    int k = 0;
    const int rc = guard([&] {
        ms->tag_storage.clear();
        for (auto& kv : ms->drv->profile()) {
            ms->tag_storage.push_back(kv.first);
            if (k < max_tags && seconds) seconds[k] = kv.second;
            ++k;
        }
        if (names)
            for (int i = 0; i < k && i < max_tags; ++i) names[i] = ms->tag_storage[i].c_str();
    });
    return rc == 0 ? k : -std::abs(rc);
}
int sanm_anm_trace(const sanm_anm_solver* s, int max_n, double* b_norm, double* x_norm, double* t) {
    int k = s->drv->trace_t.size();
    for (int i = 0; i < k && i < max_n; ++i) {
        if (b_norm) b_norm[i] = s->drv->trace_b_norm[i];
        if (x_norm) x_norm[i] = s->drv->trace_x_norm[i];
        if (t) t[i] = s->drv->trace_t[i];
    }
    return k;
}
int64_t sanm_anm_verbose_text(const sanm_anm_solver* s, char* buf, int64_t cap) {
    const std::string& t = s->drv->verbose_text();
    if (buf && cap > 0) {
        const size_t k = std::min<size_t>(t.size(), (size_t)cap - 1);
        std::memcpy(buf, t.data(), k);
        buf[k] = 0;
    }
    return (int64_t)t.size();
}
int sanm_anm_pade_diag(const sanm_anm_solver* s, double head[8], double* d, int d_cap, int* nd, double* probes,
                       int probe_cap) {
    return guard([&] {
        const PadeDiag& g = s->drv->pade_diag();
        const double h[8] = {(double)g.attempted, (double)g.built, (double)g.roots_valid, (double)g.accepted,
                             g.start,             g.pole,          g.t_max_a,             (double)g.probes.size()};
        for (int i = 0; i < 8; ++i) head[i] = h[i];
        if (nd) *nd = g.d.size();
        for (size_t i = 0; d && i < g.d.size() && (int)i < d_cap; ++i) d[i] = g.d[i];
        for (size_t i = 0; probes && i < g.probes.size() && (int)i < probe_cap; ++i) {
            probes[3 * i] = g.probes[i].a;
            probes[3 * i + 1] = g.probes[i].margin;
            probes[3 * i + 2] = g.probes[i].ok;
        }
    });
}
int sanm_anm_jacobian_csr(const sanm_anm_solver* s, int64_t* n, int64_t* nnz, uint32_t* rowptr,
                          uint32_t* col, double* val) {
    return guard([&] {
        const JacobianPattern& p = s->drv->pattern();
        if (n) *n = p.n();
        if (nnz) *nnz = p.nnz();
        if (rowptr) std::memcpy(rowptr, p.h_rowptr().data(), (p.n() + 1) * 4);
        if (col) std::memcpy(col, p.h_col().data(), p.nnz() * 4);
        if (val) backend()->d2h(val, p.csr().val, p.nnz() * 8);
    });
}

// ---- fea -------------------------------------------------------------------
int sanm_fea_model_create(int64_t nv, const double* vertices, int64_t nr_tet, const int32_t* tets,
                          const uint8_t* fixed_mask, int energy_model, double young, double poisson,
                          int inverse, const double* init_vtx_coord, const double* vtx_delta,
                          sanm_fea_model** m) {
    return guard([&] {
        auto p = std::make_unique<sanm_fea_model>();
        Material mat = Material::from_young_poisson(young, poisson);
        if (inverse) {
            sanm_check(!init_vtx_coord && !vtx_delta, "inverse model takes no init coord / delta");
            make_inverse(p->m, nv, vertices, nr_tet, tets, fixed_mask, (EnergyModel)energy_model, mat);
        } else {
            make_forward(p->m, nv, vertices, nr_tet, tets, fixed_mask, (EnergyModel)energy_model, mat,
                         init_vtx_coord, vtx_delta);
        }
        p->graph_view.g = p->m.graph;
        p->inp.d = p->m.lt_inp;
        p->out.d = p->m.lt_out;
        // ordering hint for the direct solver: position of the vertex of every unknown
        p->out.d.out_coords.resize(p->m.n * 3);
        const double* pos = init_vtx_coord ? init_vtx_coord : vertices;
        for (int64_t i = 0; i < p->m.n; ++i)
            for (int d = 0; d < 3; ++d)
                p->out.d.out_coords[i * 3 + d] = pos[(int64_t)p->m.vertex_loc[i].first * 3 + d];
        *m = p.release();
    });
}
void sanm_fea_model_destroy(sanm_fea_model* m) { delete m; }
int sanm_fea_model_nr_unknown(const sanm_fea_model* m, int64_t* n) {
    return guard([&] { *n = m->m.n; });
}
const sanm_graph* sanm_fea_model_graph(const sanm_fea_model* m) { return &m->graph_view; }
int sanm_fea_model_output_var(const sanm_fea_model* m) { return m->m.y; }
int sanm_fea_model_F_var(const sanm_fea_model* m) { return m->m.F; }
const sanm_sparse_desc* sanm_fea_model_remap_inp(const sanm_fea_model* m) { return &m->inp; }
const sanm_sparse_desc* sanm_fea_model_remap_out(const sanm_fea_model* m) { return &m->out; }
int sanm_fea_model_x0(const sanm_fea_model* m, double* x0) {
    return guard([&] { std::memcpy(x0, m->m.x0.data(), m->m.n * 8); });
}
int sanm_fea_model_copy_vtx_values(const sanm_fea_model* m, const double* vtx_values, double* out) {
    return guard([&] {
        for (int64_t i = 0; i < m->m.n; ++i) {
            auto [v, c] = m->m.vertex_loc[i];
            out[i] = vtx_values[(int64_t)v * 3 + c];
        }
    });
}
int sanm_fea_model_scatter(const sanm_fea_model* m, const double* x, double* vertices) {
    return guard([&] {
        for (int64_t i = 0; i < m->m.n; ++i) {
            auto [v, c] = m->m.vertex_loc[i];
            vertices[(int64_t)v * 3 + c] = x[i];
        }
    });
}
int sanm_fea_gravity_load(int64_t nv, const double* vertices, int64_t nr_tet, const int32_t* tets,
                          double density, const double g[3], double* f_load) {
    return guard([&] {
        std::vector<double> f;
        gravity_load(nv, vertices, nr_tet, tets, density, g, f);
        std::memcpy(f_load, f.data(), f.size() * 8);
    });
}
int sanm_fea_boundary_by_threshold(int64_t nv, const double* vertices, const uint8_t* is_surface,
                                   const double proj_dir[3], double thresh,
                                   const double* filter_dir, double filter_min, double filter_max,
                                   uint8_t* fixed_mask) {
    return guard([&] {
        std::vector<uint8_t> f;
        boundary_by_threshold(nv, vertices, is_surface, proj_dir, thresh, filter_dir, filter_min,
                              filter_max, f);
        std::memcpy(fixed_mask, f.data(), f.size());
    });
}

// ---- stand-alone Pade approximation ----------------------------------------
struct sanm_pade {
    std::vector<DVec> xs;
    std::vector<double> t_coeffs;
    std::unique_ptr<PadeApproximation> pade;
    int64_t len = 0;
};
int sanm_pade_create(int nr_coeff, int64_t len, const double* xs, int anm_cond, sanm_pade** out) {
    return guard([&] {
        sanm_check(nr_coeff >= 3 && len >= 2, "pade: %d coefficients of %ld entries", nr_coeff, (long)len);
        Backend* be = backend();
        auto p = std::make_unique<sanm_pade>();
        p->len = len;
        p->xs.resize(nr_coeff);
        for (int i = 0; i < nr_coeff; ++i) {
            p->xs[i] = DVec{be, (size_t)len};
            be->h2d(p->xs[i].p(), xs + (int64_t)i * len, len * 8);
            p->t_coeffs.push_back(xs[(int64_t)i * len + len - 1]);
        }
        p->pade = std::make_unique<PadeApproximation>(be, p->xs, p->t_coeffs, anm_cond != 0);
        *out = p.release();
    });
}
void sanm_pade_destroy(sanm_pade* p) { delete p; }
int sanm_pade_estimate_valid_range(sanm_pade* p, double start, double eps, double limit, int* ok) {
    return guard([&] { *ok = p->pade->estimate_valid_range(start, eps, limit) ? 1 : 0; });
}
int sanm_pade_get_t_max(const sanm_pade* p, double* t_max, double* t_max_a) {
    return guard([&] {
        *t_max = p->pade->get_t_max();
        *t_max_a = p->pade->get_t_max_a();
    });
}
int sanm_pade_solve_a(const sanm_pade* p, double t, double* a) {
    return guard([&] { *a = p->pade->solve_a(t); });
}
int sanm_pade_eval_xt(const sanm_pade* p, double a, double* xt) {
    return guard([&] {
        Backend* be = backend();
        DVec out{be, (size_t)p->len};
        p->pade->eval_xt(a, out.p());
        be->d2h(xt, out.p(), p->len * 8);
    });
}

// ---- host scalar helpers -----------------------------------------------------
int sanm_poly_solve_eqn(const double* f, int n, double xmin, double xmax, double b, double eps,
                        double* x) {
    return guard([&] { *x = poly::solve_eqn(std::vector<double>(f, f + n), xmin, xmax, b, eps); });
}
int sanm_poly_real_roots(const double* f, int n, double* roots, int* nr_roots) {
    return guard([&] {
        std::vector<double> r;
        if (!poly::real_roots(std::vector<double>(f, f + n), r)) {
            *nr_roots = -1;
            return;
        }
        *nr_roots = r.size();
        for (size_t i = 0; i < r.size(); ++i) roots[i] = r[i];
    });
}

int sanm_poly_roots(const double* f, int n, int only_real, int max_iter, double tol, double* re, double* im,
                    int* nr_roots) {
    return guard([&] {
        std::vector<std::complex<double>> z;
        if (!poly::roots(std::vector<double>(f, f + n), only_real != 0, z, max_iter, tol)) {
            *nr_roots = -1;
            return;
        }
        *nr_roots = z.size();
        for (size_t i = 0; i < z.size(); ++i) {
            re[i] = z[i].real();
            im[i] = z[i].imag();
        }
    });
}

}  // extern "C"
