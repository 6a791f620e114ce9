/*
 * sanm_hip.h -- C ABI of the MI355X-native ANM hot path (libsanm_hip.so).
 *
 * Every entry point replaces a piece of the C++ interface the reference's ANM
 * inner loop sits behind (jia-kai/SANM; paths relative to the reference
 * root).  The reference has no C ABI of its own: INTEGRATION.md shows the
 * C++ adapter a maintainer would put in front of these calls.
 *
 * Conventions
 *   - every function returns 0 on success, otherwise an error code; the text
 *     of the last error of the calling thread is sanm_hip_last_error().
 *     Codes map to the reference's exception types (libsanm/utils.h:34-50).
 *   - all pointers are HOST pointers owned by the caller; tensors are
 *     row-major fp64, batch first: (T,3,3) matrices are T*9 doubles.
 *   - handles are opaque; device memory is owned by the handle.
 *   - one host thread per context; no callbacks into the caller.
 *   - there is no CPU fallback: sanm_hip_init() fails if no HIP device exists.
 */
#ifndef SANM_HIP_H
#define SANM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SANM_HIP_OK 0
#define SANM_HIP_ERR_ASSERT 1      /* SANMAssertionError, libsanm/utils.cpp:55-68 */
#define SANM_HIP_ERR_NUMERICAL 2   /* SANMNumericalError, libsanm/anm.cpp:354 */
#define SANM_HIP_ERR_HIP 3         /* HIP runtime failure / no device */
#define SANM_HIP_ERR_UNSUPPORTED 4 /* valid in the reference, not on the device path */
#define SANM_HIP_ERR_UNKNOWN 5

typedef struct sanm_graph sanm_graph;               /* ComputingGraph, libsanm/symbolic.h:283-293 */
typedef struct sanm_sparse_desc sanm_sparse_desc;   /* SparseLinearDescCompressed, libsanm/anm.h:76-85 */
typedef struct sanm_taylor_prop sanm_taylor_prop;   /* TaylorCoeffProp, libsanm/symbolic.h:337-383 */
typedef struct sanm_anm_solver sanm_anm_solver;     /* ANM*Solver, libsanm/anm.h:209-305 */
typedef struct sanm_fea_model sanm_fea_model;       /* ElasticForceModel, fea/mesh.h:149-226 */

/* ---- context ----------------------------------------------------------- */
/* select the device (one process per GPU); replaces sanm::set_num_threads
 * (libsanm/tensor.cpp:86-94) as the place where parallel resources are bound */
int sanm_hip_init(int device);
const char* sanm_hip_last_error(void);
const char* sanm_hip_backend_name(void);
/* Version of this interface: bumped whenever a record a caller allocates (sanm_hyper_param, sanm_anm_stats) grows or
 * an entry point changes meaning.  A consumer compiled against this header checks
 * sanm_hip_abi_version() == SANM_HIP_ABI_VERSION once after loading the library (adapter/anm_hip.h does), or uses
 * the *_sized calls, which write no more than the caller's own record holds. */
#define SANM_HIP_ABI_VERSION 7
int sanm_hip_abi_version(void);

/* ---- operator API: libsanm/oprs.h:14-103, oprs.cpp:16-102 --------------- */
/* variables are int ids local to the graph (VarNode*, symbolic.h:222-251) */
int sanm_graph_create(sanm_graph** g);
void sanm_graph_destroy(sanm_graph* g);
int sanm_graph_placeholder(sanm_graph* g, int* var);                       /* oprs.h:80 */
/* sanm_graph_constant: val is (batch, size), batch = the graph's batch or 1 (broadcast); nine values are a (T,3,3)
 * matrix, any other size a (batch, size) vector (sanm_graph_constant_matrix declares other matrix shapes) */
/* ---- graphs over vectors: Slice / Concat (libsanm/oprs/misc.cpp:104-331, oprs.h:60) ---------------------------
 * A placeholder declared with sanm_graph_placeholder_vector is a (batch, size) tensor; graphs over it may use the
 * elementwise operators (linear_combine, multiply, pow, log, reduce_sum axis -1), constants of any length, and
 *   sanm_graph_slice:  x[:, begin:end]  (has_begin / has_end = 0 stand for the reference's None; negative values
 *                      count from the end; axis must be 1 and stride 1 like the reference's implementation)
 *   sanm_graph_concat: concatenation along axis 1.
 * They run on a vector interpreter of their own on the device (one workgroup per batch item) and are served by the
 * operator-level API (sanm_taylor_*: push_xi, compute_next_order_bias, get_jacobian -> (B, odim, idim)); the ANM
 * drivers run them as well (a dense LU with partial pivoting solves their small general systems).  Vectors of up
 * to 256 elements.
 *
 * Matrices of other sizes than 3 x 3 (libsanm/tensor_linalg.cpp:107-210 dynamic sizes; tests/symbolic.cpp:179-424 run
 * the operators at 4 x 4, 4 x 6, 5 x 5, 7 x 7): sanm_graph_placeholder_matrix declares a (batch, rows, cols) input,
 * sanm_graph_constant_matrix a constant of that rank.  batched_matmul (any conforming shapes), batched_transpose,
 * batched_mat_inv_mul / batched_det (square, up to 8 x 8; the determinant's series by the expansion up to 4 x 4 and by
 * the DFT of the polynomial matrix above, tensor_polymat.cpp:30-136, :325-379), batched_mul_eye(dim) and
 * batched_svd_w (the polar recurrences when only W is read, the full ones otherwise: tensor_svd.cpp:275-475) take them;
 * a graph with any shape other than (T,3,3) / (T,1) runs on the same interpreter as the vector graphs. */
int sanm_graph_placeholder_vector(sanm_graph* g, int size, int* var);
int sanm_graph_placeholder_matrix(sanm_graph* g, int rows, int cols, int* var);
int sanm_graph_constant_matrix(sanm_graph* g, const double* val, int64_t batch, int rows, int cols, int* var);
int sanm_graph_slice(sanm_graph* g, int x, int axis, int has_begin, int begin, int has_end, int end, int stride,
                     int* var);
int sanm_graph_concat(sanm_graph* g, int n, const int* vars, int axis, int* var);
int sanm_graph_constant(sanm_graph* g, const double* val, int64_t batch, int size, int* var); /* oprs.h:88 */
int sanm_graph_linear_combine(sanm_graph* g, int n, const double* coeffs, const int* vars,
                              double bias, int* var);                     /* oprs.h:76-77 */
int sanm_graph_multiply(sanm_graph* g, int a, int b, int* var);            /* SymbolVar::operator* */
int sanm_graph_pow(sanm_graph* g, int x, double exponent, int* var);       /* SymbolVar::pow */
int sanm_graph_log(sanm_graph* g, int x, int* var);                        /* SymbolVar::log */
int sanm_graph_reduce_sum(sanm_graph* g, int x, int axis, int* var);       /* SymbolVar::reduce_sum */
int sanm_graph_batched_matmul(sanm_graph* g, int a, int b, int* var);      /* SymbolVar::batched_matmul */
/* a < 0 means "a is the identity" (oprs.h:66-71) */
int sanm_graph_batched_mat_inv_mul(sanm_graph* g, int x, int a, int is_left, int* var);
int sanm_graph_batched_det(sanm_graph* g, int x, int* var);                /* SymbolVar::batched_det */
int sanm_graph_batched_transpose(sanm_graph* g, int x, int* var);          /* SymbolVar::batched_transpose */
int sanm_graph_batched_mul_eye(sanm_graph* g, int x, int dim, int* var);   /* SymbolVar::batched_mul_eye */
/* usw[3] = ids of U, S, W (SymbolVar::batched_svd_w, oprs.h:55) */
int sanm_graph_batched_svd_w(sanm_graph* g, int x, int require_rotation, int usw[3]);

/* ---- sparse remaps: libsanm/anm.h:24-85 --------------------------------- */
/* CSR by output element: output i = sum_{p in [rowptr[i],rowptr[i+1])} coeff[p]*x[idx[p]] */
int sanm_sparse_desc_create(int64_t out_size, int64_t in_size, const uint64_t* rowptr,
                            const uint64_t* idx, const double* coeff, sanm_sparse_desc** d);
void sanm_sparse_desc_destroy(sanm_sparse_desc* d);
/* optional ordering hint: spatial position (out_size,3) of every output element
 * of a remap_out; NULL clears it.  The fea model builder sets it by itself. */
int sanm_sparse_desc_set_out_coords(sanm_sparse_desc* d, const double* coords);

/* ---- sparse direct solver: SparseSolver, libsanm/sparse_solver.h:17-87 ---- */
/* ctor + make_builder: the pattern (CSR, uint32) is analysed once;
 * coords (n,3) is an optional nested-dissection hint */
typedef struct sanm_direct_solver sanm_direct_solver;
int sanm_direct_solver_create(int64_t n, const uint32_t* rowptr, const uint32_t* col,
                              const double* coords, sanm_direct_solver** s);
void sanm_direct_solver_destroy(sanm_direct_solver* s);
/* prepare (sparse_solver.cpp:327-421): numeric LU of the values (nnz doubles) */
int sanm_direct_solver_factor(sanm_direct_solver* s, const double* val, int* nr_bad_pivot);
/* solve (sparse_solver.cpp:154-180) */
int sanm_direct_solver_solve(sanm_direct_solver* s, const double* b, double* x);
/* apply (sparse_solver.cpp:202-215): y = A x with the values of the last factor call */
int sanm_direct_solver_apply(sanm_direct_solver* s, const double* x, double* y);
/* coeff_l2 (sparse_solver.cpp:217-223): Frobenius norm of the values of the last factor call */
int sanm_direct_solver_coeff_l2(sanm_direct_solver* s, double* l2);
int sanm_direct_solver_stats(const sanm_direct_solver* s, int64_t* nnz_factors, double* flops,
                             int32_t* nr_front, int32_t* nr_level, int32_t* max_front,
                             int32_t* root_pivots, int32_t* nr_supervar);

/* ---- TaylorCoeffProp on the device: libsanm/symbolic.h:337-383 ---------- */
/* remap_inp maps the flat input vector to the (T,3,3) placeholder
 * (SparseLinearDesc::apply is fused into the pass, libsanm/anm.cpp:55-75) */
int sanm_taylor_create(const sanm_graph* g, int out_var, const sanm_sparse_desc* remap_inp,
                       int max_order, sanm_taylor_prop** prop);
void sanm_taylor_destroy(sanm_taylor_prop* p);
/* per-tet size of the output variable: 9 (a 3x3 matrix, what the ANM drivers need) or 3 (the singular values of
 * batched_svd_w); the (T,3,3) / (T,9,9) shapes below read (T,size) / (T,size,9) */
int sanm_taylor_output_size(const sanm_taylor_prop* p, int* size);
/* push_xi (symbolic.cpp:162-204): x has remap_inp.in_size doubles; y_k (T,3,3) may be NULL */
int sanm_taylor_push_xi(sanm_taylor_prop* p, const double* x, double* y_k);
/* compute_next_order_bias (symbolic.cpp:249-289): bias (T,3,3) */
int sanm_taylor_compute_next_order_bias(sanm_taylor_prop* p, double* bias);
/* get_jacobian (symbolic.cpp:297-303): (T, 9, 9) = d out / d placeholder */
int sanm_taylor_get_jacobian(sanm_taylor_prop* p, double* jac);
/* coefficient `order` of any graph variable, (T, size); order < 0: current bias */
int sanm_taylor_get_var(sanm_taylor_prop* p, int var, int order, double* dst);
/* start over at order 0 (a new TaylorCoeffProp in the reference, anm.cpp:205) */
int sanm_taylor_reset(sanm_taylor_prop* p);

/* ---- ANM solvers: libsanm/anm.h:96-305 ---------------------------------- */
typedef struct sanm_hyper_param { /* ANMDriverHelper::HyperParam, anm.h:100-114, :247-251 */
    int use_pade;       /* any order, as in the reference (pade.cpp:13-30): the device's Gram-Schmidt / probe kernels take
                           24 series vectors per launch and run longer series in chunks (orders beyond 25: same arithmetic
                           order as one pass, more launches; rounds 1-3 refused them) */
    int sanity_check;
    int order;
    double maxr;
    double solution_check_tol;
    double xcoeff_l2_penalty;
    double converge_rms;
    double solver_rtol; /* device linear solver: relative residual target */
    int solver_maxit;
    int solver_kind;    /* 0 = Jacobi-PCG, 1 = multifrontal LU (default) */
    int profile;        /* 1: sync + host clock around each phase (ScopedProfiler tags, utils.h:225-249);
                           2: device events around each phase, no synchronisation (bench.py) */
    int solver_refine;  /* direct solver: steps of iterative refinement per solve (residual in double-double).
                           0 = only after a factorisation that had to perturb pivots (2 steps then), which is
                           what PARDISO does with the reference's settings (sparse_solver.cpp:107-127) */
} sanm_hyper_param;
void sanm_hyper_param_default(sanm_hyper_param* hp, int eqn_solver);

/* ANMEqnSolver: solve f(x) + y = 0 (anm.cpp:446-491) */
int sanm_anm_eqn_solver_create(const sanm_graph* g, int out_var, const sanm_sparse_desc* remap_inp,
                               const sanm_sparse_desc* remap_out, const double* x0, const double* y,
                               int64_t n, const sanm_hyper_param* hp, sanm_anm_solver** s);
/* Tet-sharded ANMEqnSolver, one process per GPU.  Replaces ParallelTaylorCoeffProp's
 * worker sharding (libsanm/symbolic.cpp:306-590): rank r owns tets
 * [r*T/world, (r+1)*T/world) for the Taylor passes and the assembly; the nodal
 * vectors f(x0), b_k (n doubles, once per order) and the Jacobian values (once per
 * step) are summed across ranks through `allreduce`, which must perform an in-place
 * sum of `count` doubles at the DEVICE pointer `buf` on all ranks and return 0
 * (ncclAllReduce on RCCL; the linear solve is replicated).  Every rank passes the
 * same graph, remaps, x0 and y.
 * With allreduce == NULL the library's own communicator is used (sanm_hip_comm_init below): ncclAllReduce is
 * queued on the solver's stream and the order loop never waits for the host.  rank/world must then equal the
 * communicator's.  world == 1 is allowed (the sharded code path on one rank). */
typedef int (*sanm_allreduce_fn)(void* user, double* buf, int64_t count);
/* RCCL communicator of this process's device context (one process per GPU; RCCL is loaded with dlopen on first
 * use).  Rank 0 calls sanm_hip_comm_unique_id (128 bytes = ncclUniqueId) and hands the bytes to every rank by any
 * out-of-band means; then every rank calls sanm_hip_comm_init, a collective.  Replaces the worker-thread pool of
 * ParallelTaylorCoeffProp (libsanm/symbolic.cpp:306-590) as the means by which shards exchange results. */
/* 1 when RCCL can be loaded by this process, 0 otherwise; no collective, no error state.  Lets the ranks AGREE on
 * the collective path before any of them enters sanm_hip_comm_init (sanm_amd/dist.py). */
int sanm_hip_comm_available(void);
int sanm_hip_comm_unique_id(void* id, size_t cap);
int sanm_hip_comm_init(int rank, int world, const void* id, size_t id_bytes);
int sanm_hip_comm_destroy(void);
/* Size and rank of the live communicator as RCCL itself reports them (ncclCommCount / ncclCommUserRank); 0, 0 when
 * the process holds none.  bench.py prints the size as "rccl_ranks": proof of how many ranks the collectives span. */
int sanm_hip_comm_query(int* world, int* rank);
int sanm_anm_eqn_solver_create_sharded(const sanm_graph* g, int out_var,
                                       const sanm_sparse_desc* remap_inp,
                                       const sanm_sparse_desc* remap_out, const double* x0,
                                       const double* y, int64_t n, const sanm_hyper_param* hp, int rank,
                                       int world, sanm_allreduce_fn allreduce, void* user,
                                       sanm_anm_solver** s);
/* ANMSolverVecScale: f(x) + t*v = 0 (anm.cpp:322-341) */
int sanm_anm_vecscale_solver_create(const sanm_graph* g, int out_var,
                                    const sanm_sparse_desc* remap_inp,
                                    const sanm_sparse_desc* remap_out, const double* x0, double t0,
                                    const double* v, int64_t n, const sanm_hyper_param* hp,
                                    sanm_anm_solver** s);
/* ANMImplicitSolver: F(x,t) = F(x0,t0) (anm.cpp:494-508); remap_inp has n+1 inputs */
int sanm_anm_implicit_solver_create(const sanm_graph* g, int out_var,
                                    const sanm_sparse_desc* remap_inp,
                                    const sanm_sparse_desc* remap_out, const double* x0, double t0,
                                    int64_t n, const sanm_hyper_param* hp, sanm_anm_solver** s);
void sanm_anm_solver_destroy(sanm_anm_solver* s);

int sanm_anm_next_iter(sanm_anm_solver* s);                      /* ANMEqnSolver::next_iter */
int sanm_anm_update_approx(sanm_anm_solver* s);                  /* ANMDriverHelper::update_approx */
/* begin a new solve from x0 on the same model: what constructing a new
 * ANMEqnSolver does in the reference (fea/main.cpp:418), minus rebuilding the
 * device program / CSR pattern, which depend on the mesh only */
int sanm_anm_restart(sanm_anm_solver* s, const double* x0);
/* exactly `count` more completed continuation steps without returning to the caller in between: next_iter while
 * the solve has not converged, sanm_anm_restart(x0) when it has (what a driver loop around the two calls does,
 * minus the caller's own time between them; bench.py's timed region) */
int sanm_anm_run_steps(sanm_anm_solver* s, int count, const double* x0, int* nr_restart);
/* measurement hook for bench.py: average duration (ms) of `reps` back-to-back
 * launches of one kernel on the solver's own data, HIP events on its stream.
 * kernel 0: Taylor pass (mode 0..3 = eval0/grad/bias/coeff at `order`),
 * kernel 1: CSR SpMV, kernel 2: PCG SpMV+dot.  The solver state is clobbered:
 * call sanm_anm_restart() afterwards. */
int sanm_anm_time_kernel(sanm_anm_solver* s, int kernel, int reps, int mode, int order,
                         double* avg_ms);
/* The HIP source of the pass kernels specialised for the solver's compiled graph (what the library compiles at run
 * time for batches of SANM_JIT_MIN_T tets or more).  Returns the length of the source;
 * copies at most cap-1 characters and a terminator into buf if it is not NULL. */
int64_t sanm_anm_spec_source(sanm_anm_solver* s, char* buf, int64_t cap);
/* measurement hook for bench.py: when enabled, every launch of the Taylor pass
 * kernel (the graph interpreter) is bracketed by HIP events on the solver's
 * stream.  Each call first returns the summed duration / launch count gathered
 * since the previous call (pass NULL to skip), then sets the enable flag. */
int sanm_anm_pass_timing(sanm_anm_solver* s, int enable, double* total_ms, int64_t* count);
int sanm_anm_converged(const sanm_anm_solver* s, int* flag);     /* ANMEqnSolver::converged */
int sanm_anm_residual_rms(const sanm_anm_solver* s, double* r);  /* ANMEqnSolver::residual_rms */
int sanm_anm_get_x(const sanm_anm_solver* s, double* x);         /* ANMEqnSolver::get_x (n) */
int sanm_anm_get_t_upper(const sanm_anm_solver* s, double* t);   /* get_t_upper */
int sanm_anm_get_t_max_a(const sanm_anm_solver* s, double* a);
int sanm_anm_solve_a(const sanm_anm_solver* s, double t, double* a); /* solve_a */
int sanm_anm_eval(const sanm_anm_solver* s, double a, double* x, double* t); /* eval */
int sanm_anm_nr_iter(const sanm_anm_solver* s, int64_t* iter);   /* get_nr_ieter (sic), anm.h:139 */
int sanm_anm_nr_xt_coeffs(const sanm_anm_solver* s, int* nr);
int sanm_anm_xt_coeff(const sanm_anm_solver* s, int i, double* xt); /* xt_coeffs()[i], n+1 doubles */
int sanm_anm_has_pade(const sanm_anm_solver* s, int* flag);
/* statistics of the run so far */
typedef struct sanm_anm_stats {
    int64_t nr_unknown, nr_tet, jacobian_nnz, assembly_contribs;
    int64_t nr_linear_solve, linear_iters_total, linear_iters_last;
    double linear_relres_last;
    double arena_bytes;
    /* direct solver analysis (0 with the iterative solver) */
    int64_t factor_nnz, nr_front, nr_level, max_front;
    double factor_flops;
    /* tet-sharded solver with the factorisation distributed by subtrees (sanm_anm_eqn_solver_create_sharded, world > 1,
     * systems from 50 GFLOP per factorisation or SANM_DIST_SOLVER=1): what THIS rank factors -- its own subtrees and the
     * replicated top of the elimination tree -- in the units of factor_flops; nr_subtree == 0: replicated solver */
    double factor_flops_own, factor_flops_top;
    int64_t nr_subtree, nr_subtree_own;
    /* ... and what its exchanges move, in doubles: the Schur complements of the cut roots once per factorisation, their
     * update rows once per solve (the third exchange, the solution, is nr_unknown doubles per solve) */
    int64_t dist_schur_doubles, dist_inbox_doubles;
    /* (ABI 6) the distribution over the whole tree: every front has one owner, the fronts above the subtrees -- the top,
     * factor_flops_top -- are mapped onto the ranks as well (factor_flops_top_own: this rank's) and run in nr_dist_stage
     * - 1 stages with exchanges between them; factor_flops_critical: sum over the stages of the busiest rank's flops,
     * i.e. the factorisation's critical path (factor_flops / factor_flops_critical = its speed-up if flops-bound) */
    double factor_flops_top_own, factor_flops_critical;
    int64_t nr_dist_stage;
    /* (ABI 7) doubles of this rank's front store: every front when the solver is replicated; with the distribution the fronts
     * the rank factors and the Schur blocks it receives (a shorter caller struct simply does not get it:
     * sanm_anm_get_stats_sized) */
    int64_t front_store_doubles;
} sanm_anm_stats;
int sanm_anm_get_stats(const sanm_anm_solver* s, sanm_anm_stats* st);
/* the same for a caller whose sanm_anm_stats may be older (shorter) than the library's: at most st_bytes are written */
int sanm_anm_get_stats_sized(const sanm_anm_solver* s, void* st, size_t st_bytes);
/* Host-clock seconds of the phases of the solver's construction, in order; returns the number of phases (names /
 * seconds may be NULL).  The reference's time_solve (fea/main.cpp:382, :418-425) runs from the solver's constructor
 * to convergence, so these are part of its metric: "tet_order", "program" (graph -> device program), "jit" (the pass
 * kernels of the graph: seconds; the next entry names their source -- "jit_compiled", "jit_disk_hit",
 * "jit_memory_hit" or "jit_none" -- with the value 1), "remap_tables", "pattern" (the symbolic product
 * remap_out J remap_in), "analysis" (ordering and symbolic factorisation of the direct solver + its device tables). */
int sanm_anm_setup_profile(const sanm_anm_solver* s, int max_tags, const char** names, double* seconds);
/* profile tags: returns the number of tags (or minus an error code: the call reads device events);
 * names/seconds may be NULL */
int sanm_anm_profile(const sanm_anm_solver* s, int max_tags, const char** names, double* seconds);
/* how many times each tag was entered, in the order of sanm_anm_profile (call that first) */
int sanm_anm_profile_counts(const sanm_anm_solver* s, int max_tags, double* counts);
/* kernel launches queued inside each tag (nested tags count in their parents too), same order */
int sanm_anm_profile_launches(const sanm_anm_solver* s, int max_tags, double* launches);
/* switch the phase profile of a live solver (modes as sanm_hyper_param.profile); clear != 0 drops
 * what was accumulated so far */
int sanm_anm_set_profile(sanm_anm_solver* s, int mode, int clear);
/* per-order trace of the last expansion (needs hp.profile): |b_k|, |x_k|, t_k; returns count */
int sanm_anm_trace(const sanm_anm_solver* s, int max_n, double* b_norm, double* x_norm, double* t);
/* What the reference prints per expansion under SANM_VERBOSE (anm.cpp:200-203, :247-259, :295-309), in its format:
 *   "=== ANM iter K:\ngt=.. xgt=.. jacob=.. 1:(bi=.. xbi=..) 2:(..) ...\nbound=.. t=..\nx(a): ..\nt(a): ..,\n"
 * of the last expansion (needs hp.profile = 1, or the environment variable SANM_VERBOSE, which also makes the
 * driver print it to stdout after every expansion like the reference).  Returns the length of the text; at most
 * cap - 1 characters and a terminating 0 are written to buf (buf may be NULL to query the length). */
int64_t sanm_anm_verbose_text(const sanm_anm_solver* s, char* buf, int64_t cap);
/* Decisions of the last Pade range estimate (PadeApproximation::estimate_valid_range, pade.cpp:107-173), for
 * decision-by-decision comparison with another implementation:
 *   head[0] attempted (use_pade && a_bound < stable range, anm.cpp:143-152)   head[1] denominator built
 *   head[2] roots() returned a value (None => rejected, pade.cpp:113-116)     head[3] accepted
 *   head[4] start (= a_bound)   head[5] pole   head[6] accepted range t_max_a   head[7] number of probes
 * d: denominator coefficients m_d, low order first (*nd of them; d_cap entries are written at most);
 * probes: triples (a, margin, ok) of every check(a) in the reference's order (pade.cpp:129-165), with
 * margin = |pn_lo * D_n / D_lo - pn|^2 / (eps^2 |pn|^2): a probe passes iff margin <= 1. */
int sanm_anm_pade_diag(const sanm_anm_solver* s, double head[8], double* d, int d_cap, int* nd, double* probes,
                       int probe_cap);
/* Jacobian CSR of the current step (SparseSolver contents, sparse_solver.cpp:327-421);
 * pass NULL pointers to query sizes only */
int sanm_anm_jacobian_csr(const sanm_anm_solver* s, int64_t* n, int64_t* nnz, uint32_t* rowptr,
                          uint32_t* col, double* val);

/* ---- fea model construction: fea/mesh_template.h, fea/material.cpp ------ */
#define SANM_ENERGY_NEOHOOKEAN_I 0
#define SANM_ENERGY_NEOHOOKEAN_C 1
#define SANM_ENERGY_ARAP 2
#define SANM_ENERGY_STVK_STRETCH 3
/* DeformableBody::make_forward / make_inverse (fea/mesh_template.h:174-219):
 * vertices (nv,3), tets (T,4) int32, fixed_mask (nv,3) uint8;
 * init_vtx_coord, vtx_delta: (nv,3) or NULL */
int sanm_fea_model_create(int64_t nv, const double* vertices, int64_t nr_tet, const int32_t* tets,
                          const uint8_t* fixed_mask, int energy_model, double young, double poisson,
                          int inverse, const double* init_vtx_coord, const double* vtx_delta,
                          sanm_fea_model** m);
void sanm_fea_model_destroy(sanm_fea_model* m);
int sanm_fea_model_nr_unknown(const sanm_fea_model* m, int64_t* n);
const sanm_graph* sanm_fea_model_graph(const sanm_fea_model* m);
int sanm_fea_model_output_var(const sanm_fea_model* m);
int sanm_fea_model_F_var(const sanm_fea_model* m);
const sanm_sparse_desc* sanm_fea_model_remap_inp(const sanm_fea_model* m);
const sanm_sparse_desc* sanm_fea_model_remap_out(const sanm_fea_model* m);
int sanm_fea_model_x0(const sanm_fea_model* m, double* x0);                 /* lt_inp->x0() */
/* MeshShapeMatTrans::copy_vtx_values (mesh_template.h:113-128): (nv,3) -> (n) */
int sanm_fea_model_copy_vtx_values(const sanm_fea_model* m, const double* vtx_values, double* out);
/* scatter unknowns back into a (nv,3) vertex array (replace_with_mask, fea/mesh.cpp:14-25) */
int sanm_fea_model_scatter(const sanm_fea_model* m, const double* x, double* vertices_inout);
/* sparse descriptor contents, for tests (pass NULL to query sizes) */
int sanm_sparse_desc_get(const sanm_sparse_desc* d, int64_t* out_size, int64_t* in_size,
                         int64_t* nnz, uint64_t* rowptr, uint64_t* idx, double* coeff);
/* nodal gravity load (fea/main.cpp:1025-1036): f_load (nv,3) */
int sanm_fea_gravity_load(int64_t nv, const double* vertices, int64_t nr_tet, const int32_t* tets,
                          double density, const double g[3], double* f_load);
/* setup_boundary_by_config (fea/main.cpp:921-982): filter_dir may be NULL */
int sanm_fea_boundary_by_threshold(int64_t nv, const double* vertices, const uint8_t* is_surface,
                                   const double proj_dir[3], double thresh,
                                   const double* filter_dir, double filter_min, double filter_max,
                                   uint8_t* fixed_mask);

/* ---- PadeApproximation on its own (libsanm/pade.h:21-62; what tests/pade.cpp:64-110 drives) --------------------
 * xs: (nr_coeff, len) row-major, the coefficients of a vector polynomial whose LAST entry is t (as the reference
 * assumes, pade.h:47); copied to the device.  anm_cond: xs[i] . xs[1] == (i == 1). */
typedef struct sanm_pade sanm_pade;
int sanm_pade_create(int nr_coeff, int64_t len, const double* xs, int anm_cond, sanm_pade** p);
void sanm_pade_destroy(sanm_pade* p);
/* estimate_valid_range(start, eps, limit): *ok = 0 where the reference returns false */
int sanm_pade_estimate_valid_range(sanm_pade* p, double start, double eps, double limit, int* ok);
int sanm_pade_get_t_max(const sanm_pade* p, double* t_max, double* t_max_a);
int sanm_pade_solve_a(const sanm_pade* p, double t, double* a);
int sanm_pade_eval_xt(const sanm_pade* p, double a, double* xt); /* len doubles: x(a) then t(a) */

/* ---- host scalar helpers (exposed for tests): libsanm/unary_polynomial.h - */
int sanm_poly_solve_eqn(const double* f, int n, double xmin, double xmax, double b, double eps,
                        double* x);
/* unary_polynomial::roots(f, only_real = true) as pade.cpp:113 calls it (ACM algorithm 30,
 * unary_polynomial.cpp:154-334).  roots must hold n-1 doubles, written in the order the algorithm finds them;
 * *nr_roots = -1 where the reference returns None */
int sanm_poly_real_roots(const double* f, int n, double* roots, int* nr_roots);
/* unary_polynomial::roots with all its arguments (unary_polynomial.h:50-52: max_iter 300, tol 1e-8).
 * re / im must hold n-1 doubles each; *nr_roots = -1 where the reference returns None */
int sanm_poly_roots(const double* f, int n, int only_real, int max_iter, double tol, double* re, double* im,
                    int* nr_roots);

#ifdef __cplusplus
}
#endif
#endif /* SANM_HIP_H */
