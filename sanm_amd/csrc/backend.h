// Device abstraction used by the ANM driver.
//
// The product implementation is HipBackend (backend_hip.hip): HIP memory,
// HIP kernels, one stream.  The only other implementation lives under
// tests/hostsim and exists so the host logic (graph compilation, assembly
// pattern, ANM driver, Pade) can be exercised in the GPU-less authoring
// container; it is never linked into libsanm_hip.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#include <map>
#include <string>

#include "mf_types.h"
#include "program.h"

namespace sanm_hip {

class LinearSolver;
class JacobianPattern;
struct HyperParam;

// rows of a sparse gather: dst[i] = sum_{p in [ptr[i], ptr[i+1])} coef[p] * src[idx[p]]
struct SparseRowsDev {
    const uint32_t* ptr;  // n+1
    const uint32_t* idx;
    const double* coef;
    int64_t nrows;
    // Rows in triples: when rows 3u, 3u+1, 3u+2 have the same coefficients and index lists that differ by the
    // constant offsets 0, 3, 6 (the nodal force f_c = sum_tets sum_j P[c][j] N_j: the normal does not depend on c),
    // one list per triple does: a third of the bytes.  bptr[u] .. bptr[u+1] into bidx / bcoef (the list of row 3u);
    // nullptr if the matrix does not have that structure.
    const uint32_t* bptr;
    const uint32_t* bidx;
    const double* bcoef;
};

struct CsrDev {
    const uint32_t* rowptr;  // n+1
    const uint32_t* col;
    double* val;
    int64_t n, nnz;
};

// Value assembly of the Jacobian A = remap_out . blockdiag(J_e) . remap_in (libsanm/anm.cpp:362-438, :520-608):
//   A[i, c] = sum over the triples (p, m, q) -- p an entry of row i of remap_out (output element e = b * odim + o of
//   batch item b, coefficient c_out), m < idim, q an entry of row b * idim + m of remap_in with column c -- of
//   drop((c_out * c_in) * J_b[o][m]),   drop(t) = |t| < 1e-9 ? 0 : t   (libsanm/sparse_solver.cpp:291-293),
// summed in the order (p, m, q).  The triples are enumerated AT ASSEMBLY TIME from the two remap tables (round 5):
// rounds 1-4 kept a gather list per non-zero -- 31 M (index, coefficient) pairs = 372 MB for armadillo_small, 3 GB for
// the 338 k-tet mesh, built on the host at set-up (the largest part of the solver's construction) and streamed from
// HBM at every assembly.  Column n of remap_in (the continuation parameter t of ANMImplicitSolver) goes to grad_t.
struct AssemblyDev {
    const uint32_t* ro_ptr;  // remap_out by unknown: n + 1
    const uint32_t* ro_idx;  // output element b * odim + o (global batch index b)
    const double* ro_coef;
    const uint32_t* ri_ptr;  // remap_in by placeholder element b * idim + m: T * idim + 1
    const uint32_t* ri_idx;  // unknown (or n: the t column)
    const double* ri_coef;
    const uint32_t* rowptr;  // CSR pattern of A (columns ascending in a row)
    const uint32_t* col;
    int64_t n;
    int32_t odim, idim;
    int64_t tet_begin, tet_end;  // batch items whose blocks `jac` holds ([b - tet_begin][odim][idim]); the others'
                                 // contributions are another rank's
    int32_t has_t;
    // Device back end: the triples of every non-zero as a gather list (index into jac, c_out * c_in) in enumeration
    // order, BUILT ON THE DEVICE from the tables above when the pattern is created (Backend::prepare_assembly: two
    // passes of one workgroup per row, ~25 ms for 235 k unknowns where the host loop of rounds 1-4 took 1.7 s).
    // Enumerating the triples at every assembly instead was measured (round 5): 14x slower than streaming the list.
    // nullptr on the host harness, which assembles from the tables (row_ops.h: assemble_row).
    const uint32_t* aptr = nullptr;   // nnz + 1
    const uint32_t* ajidx = nullptr;
    const double* acoef = nullptr;
    const uint32_t* tptr = nullptr;   // n + 1 (the t column's triples by row), with has_t
    const uint32_t* tjidx = nullptr;
    const double* tcoef = nullptr;
    int64_t nnz = 0;
    // Rows in triples (set by JacobianPattern when rows 3u, 3u+1, 3u+2 of remap_out have the same coefficients and
    // output elements that differ by 0 / 3 / 6 inside one batch item's block -- the three components of a vertex -- and
    // there is no t column): the three rows have the same columns, and the gather list of a non-zero of row 3u+c is
    // that of the same non-zero of row 3u with its Jacobian indices shifted by 3 c idim.  The device keeps the lists of
    // the rows 3u only -- a third of the bytes to build, to hold and to stream at every assembly -- and a slot table:
    // tslot_p[t] = position of the t-th non-zero of the rows 3u in the CSR arrays, tslot_len[t] = its row's length
    // (the same non-zero of rows 3u+1 / 3u+2 sits tslot_len / 2 tslot_len further on).
    int32_t triples = 0;
    const uint32_t* tslot_p = nullptr;
    const uint32_t* tslot_len = nullptr;
    int64_t ntslot = 0;
};

// One phase of a classical Gram-Schmidt step of the Pade basis (pade.cpp:36-70), as a value the backend may run at
// once or carry along with one of its next launches (Backend::defer_gs_phase):
//   kind 1: red_out[j] = x . vecs[j], j < nvec  (after completing the normalisation of vecs[nvec-1], see
//           multi_dot_async);   kind 2: out = x - sum_{j >= first} coefs[j] vecs[j], *red_out = out . out;
//   kind 3: out *= 1 / max(sqrt(*norm2), eps), *red_out = out . out
struct GsPhase {
    int kind = 0;
    size_t n = 0;
    const double* x = nullptr;
    int nvec = 0;
    static constexpr int kMaxVec = 24;  // = MAX_VEC of the reduction kernels (backend_hip.hip): vectors per launch;
                                        // a phase over more of them runs in chunks (orders beyond 25)
    std::vector<double*> vecs;
    const double* coefs = nullptr;
    int first = 0;
    double* out = nullptr;
    const double* norm2 = nullptr;  // kind 3: of `out`; kind 1: of vecs[nvec-1] (may be null)
    const double* nn2 = nullptr;    // kind 1: squared norm of vecs[nvec-1] after its first scaling
    double eps = 0;
    double* red_out = nullptr;      // device memory
};

//! the order loop's x_i = -t xg - xb, t = *num * scale (Backend::next_coeff_async), as arguments
struct NextCoeff {
    size_t n = 0;
    const double* num = nullptr;  // device scalar
    double scale = 0;
    const double* xg = nullptr;
    const double* xb = nullptr;
    double* out = nullptr;    // n + 1 entries: x_i, t
    double* t_out = nullptr;  // pinned host memory
    //! device pair {t_1, xg . x_1} of an order 1 that never came to the host (Backend::x1_async): when set, the scale
    //! is formed on the device, 1 / (sc[0] - sc[1]) -- the division the host would have made (anm.cpp:246-250)
    const double* sc = nullptr;
};

class Backend {
public:
    virtual ~Backend() = default;
    virtual const char* name() const = 0;

    virtual void* alloc(size_t bytes) = 0;
    virtual void free(void* p) = 0;
    //! Device memory asked for from ANY host thread while the owner thread works (the direct solver's analysis asks for
    //! its front store -- tens of GB whose mapping takes 0.1-1 s -- as soon as it knows the size, beside the rest of the
    //! constructor): nothing of the backend's own state is touched.  nullptr: not offered (or no memory: the owner then
    //! asks again through alloc and reports).  adopt() on the owner thread makes the block one of alloc()'s; a block
    //! that is never adopted goes back through free_detached().
    virtual void* alloc_detached(size_t) { return nullptr; }
    virtual void free_detached(void*) {}
    virtual void adopt(void*, size_t) {}
    virtual void h2d(void* dst, const void* src, size_t bytes) = 0;
    virtual void d2h(void* dst, const void* src, size_t bytes) = 0;
    //! queue a copy to pinned host memory (alloc_host) without waiting: valid after the next sync()
    virtual void d2h_async(void* dst_pinned, const void* src, size_t bytes) { d2h(dst_pinned, src, bytes); }
    virtual void d2d(void* dst, const void* src, size_t bytes) = 0;
    virtual void zero(void* dst, size_t bytes) = 0;
    virtual void sync() = 0;
    //! mark(): a point in the queue; wait_mark(): wait until everything queued BEFORE the last mark has run (what is
    //! queued behind it may still be running).  Backends without queues: the default (everything has run already).
    virtual void mark() {}
    virtual void wait_mark() { sync(); }
    //! Record the launches queued between begin and end instead of running them, for replay with
    //! graph_launch (a launch-bound sequence of small kernels whose arguments do not change from one
    //! continuation step to the next).  begin returns false if the backend has no graphs: the caller then
    //! simply runs the sequence.
    //! a buffer that keyed replayed launch chains (the direct solver's front store) is going away
    virtual void forget_chains(const void* key) { (void)key; }
    virtual bool graph_capture_begin() { return false; }
    virtual void* graph_capture_end() { return nullptr; }
    virtual void graph_launch(void*) {}
    virtual void graph_destroy(void*) {}
    //! run one Gram-Schmidt phase now (through the *_async primitives below)
    void run_gs_phase(const GsPhase& ph);
    //! The same, but the backend may hold the phase back and run it inside one of its next launches on the same
    //! queue (extra workgroups of a kernel that is launched anyway: the order loop is a chain of short launches
    //! that leave most of the chip idle, and a launch of its own costs the chain its full latency).  A held phase
    //! is run at the latest by the next defer_gs_phase, flush_deferred, sync or copy to the host.
    virtual void defer_gs_phase(const GsPhase& ph) { run_gs_phase(ph); }
    virtual void flush_deferred() {}
    //! A second in-order queue beside the main one, for work that depends on the main queue only at the moment it
    //! is forked off (the Pade basis grows by one vector per Taylor order while the main queue is busy with the
    //! next order's solve, a chain of short latency-bound launches that leaves most of the chip idle).
    //! side_fork(): everything queued so far on the main queue happens before what follows; launches go to the
    //! side queue until side_end().  side_join(): what was queued on the side queue happens before what the main
    //! queue gets from now on.  sync() waits for both.  Backends without queues run everything in order.
    virtual void side_fork() {}
    virtual void side_end() {}
    virtual void side_join() {}
    //! side_detach(): what the side queue holds is no longer waited for by sync(), by copies to the host or by
    //! side_join(), until side_wait() has waited for it on the host (work whose result is read much later and
    //! that should fill the gaps the main queue leaves in the meantime).
    virtual void side_detach() {}
    virtual void side_wait() {}
    //! host memory the device can write (pinned); results of the *_async reductions land here and are
    //! valid after the next sync()
    virtual double* alloc_host(size_t n_doubles) = 0;
    virtual void free_host(double* p) = 0;

    //! one whole graph pass over all tets (tet_ops.h: exec_program_tet)
    virtual void run_pass(const ProgramDev& P, int mode, int order, const double* xvec) = 0;
    //! Compile pass kernels for one program from generated source (graph.cpp: Program::spec_source) and return a
    //! handle for ProgramDev::spec_id; -1 if the backend does not compile at run time or the compilation failed
    //! (the generic interpreter kernels then run the program).
    virtual int specialize(const char* source) { (void)source; return -1; }
    virtual void release_specialized(int id) { (void)id; }
    //! where the code object of the last successful specialize() came from: 1 the process-wide cache, 2 the on-disk
    //! cache, 3 a compilation, 4 the code objects built ahead of time into the library (0: unknown)
    virtual int last_specialize_source() const { return 0; }
    //! remap_out apply (SparseLinearDesc::apply, libsanm/anm.cpp:55-75)
    //! dst = R * src; with `perm`, additionally dst2[perm[i]] = dst[i] (the right-hand side where the direct
    //! solver wants it: saves the solver's own permutation launch)
    virtual void gather_rows(const SparseRowsDev& R, const double* src, double* dst, const int32_t* perm = nullptr,
                             double* dst2 = nullptr) = 0;
    //! Jacobian values into a fixed CSR pattern (anm.cpp:362-438 + sparse_solver.cpp:250-305)
    //! val: the CSR values; grad_t: n doubles (the t column), with A.has_t only
    virtual void assemble(const AssemblyDev& A, const double* jac, double* val, double* grad_t) = 0;
    //! once per pattern: whatever the back end wants beside the tables (the device builds its gather lists); buffers
    //! it allocates go to `owned` (the pattern frees them)
    virtual void prepare_assembly(AssemblyDev& A, std::vector<void*>& owned) {
        (void)A;
        (void)owned;
    }
    //! y = A x  (SparseSolver::apply, sparse_solver.cpp:202-215)
    virtual void spmv(const CsrDev& A, const double* x, double* y) = 0;

    //! dst[i] = src[idx[i]]  (values of A' in CSR order from those of A)
    virtual void gather(size_t n, const double* src, const uint32_t* idx, double* dst) = 0;
    //! Normal equations of the Tikhonov path (libsanm/sparse_solver.cpp:366-395): M = A'A + lambda I on a fixed
    //! pattern.  At: A' in CSR form (row i = column i of A, ascending row indices); entry e of M sits in row
    //! mrow[e], column M.col[e]: M.val[e] = (column mrow[e] of A) . (column M.col[e] of A) + lambda [on the diagonal]
    virtual void ata(const CsrDev& At, const CsrDev& M, const uint32_t* mrow, double lambda) = 0;
    //! r = b - A x for iterative refinement, accumulated in twice the working precision (the default, in
    //! backend_common.cpp, is the plain fp64 form); r may alias b
    virtual void residual(const CsrDev& A, const double* b, const double* x, double* r);

    // BLAS-1 on device vectors (libsanm/tensor.cpp:644-668, tensor_elemwise.cpp)
    virtual double dot(size_t n, const double* x, const double* y) = 0;
    //! out = a*x + b*y  (y may be null iff b == 0; out may alias x or y)
    virtual void axpby(size_t n, double a, const double* x, double b, const double* y,
                       double* out) = 0;
    //! axpby over n entries, then out[n] = tail (the t component appended to an x coefficient)
    virtual void axpby_tail(size_t n, double a, const double* x, double b, const double* y, double* out,
                            double tail) = 0;
    //! out = sum_j coefs[j] * ptrs[j]  (nvec <= 24; out may alias one of the inputs
    //! only if that input has index 0)
    virtual void lincomb(size_t n, int nvec, const double* const* ptrs, const double* coefs,
                         double* out);
    //! out_host[j] = x . ys[j]  for j < nvec (nvec <= 24), one synchronisation
    virtual void multi_dot(size_t n, const double* x, int nvec, const double* const* ys,
                           double* out_host);
    //! u = sum_j c1[j]*ptrs[j], w = sum_j c2[j]*ptrs[j], d = scale*w - u:  out_host = {d.d, u.u}.
    //! The Pade range test (pade.cpp:143-165) on two numerators without materialising them.
    virtual void lincomb2_diff_norms(size_t n, int nvec, const double* const* ptrs, const double* c1,
                                     const double* c2, double scale, double out_host[2]);
    //! the same for `ncand` (<= 7) coefficient sets at once (row c of c1 / c2 has nvec entries): the vectors
    //! are read once and there is one synchronisation for all of them; out_host = {d.d, u.u} per candidate
    virtual void lincomb2_diff_norms_multi(size_t n, int nvec, const double* const* ptrs, int ncand,
                                           const double* c1, const double* c2, const double* scale,
                                           double* out_host);
    //! out = x .* y
    virtual void vmul(size_t n, const double* x, const double* y, double* out) = 0;
    //! d[i] = 1 / A[i,i]  (scaled by `scale`)
    virtual void csr_inv_diag(const CsrDev& A, double scale, double* d) = 0;
    //! number of non-finite entries
    virtual int64_t count_nonfinite(size_t n, const double* x) = 0;
    //! the same without waiting: the count lands in *out (memory the host can read after a sync)
    virtual void count_nonfinite_async(size_t n, const double* x, double* out) { *out = (double)count_nonfinite(n, x); }
    //! max_i ( |a_i - b_i| - eps*max(1, min(|a_i|,|b_i|)) ) ; <0 means allclose
    //! (TensorND::assert_allclose, libsanm/tensor.cpp:670-684)
    virtual double allclose_excess(size_t n, const double* a, const double* b, double eps) = 0;
    /*!
     * Jacobi-preconditioned conjugate gradients on (sign*A) x = sign*b.
     *
     * The forward-FEA Jacobian is minus the (scaled) energy Hessian, hence
     * symmetric negative definite at stable states (hard part 2):
     * sign = -1 there.  dinv holds 1/(sign*diag(A)).  Stops when
     * |r| <= rtol*|b|.  Returns the iteration count in *iters (negative if
     * the operator is found indefinite: p'Ap <= 0) and the final relative
     * residual in *relres.  The default implementation is built from the
     * primitives above; HipBackend overrides it with device-resident scalars.
     */
    virtual void pcg(const CsrDev& A, double sign, const double* dinv, const double* b, double* x,
                     double rtol, int maxit, int* iters, double* relres);

    //! numeric multifrontal LU of A (values in A.val, pattern analysed in mf):
    //! scatter + extend-add + blocked partial LU of every front, level by level.
    //! Returns the number of (near-)zero pivots met.
    virtual int mf_factor(const MfDev& mf, const MfSchedule& sch, const CsrDev& A) = 0;
    //! the same without waiting: the number of rejected pivots lands in *status (as above) once the stream gets there
    virtual void mf_factor_async(const MfDev& mf, const MfSchedule& sch, const CsrDev& A, double* status) {
        *status = mf_factor(mf, sch, A);
    }
    // -- the same in pieces, for the distributed schedule (MfSchedule::Dist; anm.cpp: DirectSolver): the caller
    //    runs levels [l0, l1) and puts its exchanges between the pieces.  Nothing here waits for the device.
    //! numeric factorisation of levels [l0, l1); prologue: clear the fronts, scatter A, pivot threshold first
    virtual void mf_factor_piece(const MfDev&, const MfSchedule&, const CsrDev& A, int l0, int l1, bool prologue);
    //! *out (device memory) = number of perturbed pivots so far, as a double
    virtual void mf_factor_status(const MfDev&, double* out);
    //! forward (levels l0 .. l1-1) or backward (l1-1 .. l0) sweep over mf.work
    virtual void mf_solve_piece(const MfDev&, const MfSchedule&, bool fwd, int l0, int l1);
    //! b != null: mf.work[perm[i]] = b[i];  x != null: x[i] = mf.work[perm[i]]
    virtual void mf_permute(const MfDev&, const double* b, double* x);
    //! a batch of strided block copies (descriptors in device memory), src_base -> dst_base
    virtual void copy2d_batch(const MfCopy2D* d, int count, int max_rows, int max_cols, const double* src_base,
                              double* dst_base);
    //! x = A^-1 b with the factors of the last mf_factor (b, x: n doubles, may alias)
    virtual void mf_solve(const MfDev& mf, const MfSchedule& sch, const double* b, double* x) = 0;
    //! the same with the two ends of the solve fused into its neighbours in the order loop: b == nullptr means the
    //! permuted right-hand side already sits in mf.work (gather_rows with perm); with dot_y the kernel that writes
    //! x also forms x . dot_y into *dot_out (device memory, like dot_async)
    virtual void mf_solve_fused(const MfDev& mf, const MfSchedule& sch, const double* b, double* x,
                                const double* dot_y, double* dot_out);  // default: backend_common.cpp

    //! bracket every run_pass launch with device events (measurement runs only);
    //! pass_timing() returns the summed duration in ms and the launch count since
    //! the last enable
    virtual void enable_pass_timing(bool on) { (void)on; }
    virtual void pass_timing(double* total_ms, int64_t* count) {
        *total_ms = 0;
        *count = 0;
    }

    // ---- collective of the tet-sharded mode (one process per GPU; RCCL over xGMI in the HIP backend) ----------
    //! 128-byte identifier of a new communicator (rank 0 creates it, the caller hands it to every rank).
    //! The defaults (backend_common.cpp) report UNSUPPORTED.
    virtual bool comm_available() { return false; }
    virtual void comm_unique_id(void* id128);
    //! join the communicator; collective over all `world` ranks
    virtual void comm_init(int rank, int world, const void* id128);
    virtual void comm_destroy() {}
    virtual int comm_world() const { return 0; }  // 0: no communicator
    virtual int comm_rank() const { return 0; }
    //! size and rank as the communication library reports them for the live communicator (0, 0 without one)
    virtual void comm_query(int* world, int* rank) { *world = *rank = 0; }
    //! in-place sum of `count` doubles over all ranks, queued on the backend's stream (no host synchronisation)
    virtual void allreduce_sum(double* buf, int64_t count);
    //! the live communicator offers grouped send / receive and broadcast (ncclSend / ncclRecv / ncclBroadcast bound)
    virtual bool comm_p2p_available() { return false; }
    //! n transfers of ranges of one device buffer as ONE group, queued on the backend's stream: x[i] with dst >= 0: rank
    //! src sends base[off, off + cnt) and rank dst receives it at the same place; dst = -1: broadcast in place from
    //! src to every rank.  The same list on every rank (MfSchedule::Xfer).
    virtual void comm_exchange(double* base, const MfSchedule::Xfer* x, int n);

    //! HyperParam::solver_kind == 2: a linear solver supplied by the backend itself.  The HIP backend has none
    //! (nullptr: the driver rejects the setting); the CPU baseline harness (tests/hostsim) returns MKL PARDISO,
    //! the reference's own solver (libsanm/sparse_solver.cpp:107-127).
    virtual LinearSolver* make_external_solver(const JacobianPattern& pat, const HyperParam& hp) {
        (void)pat; (void)hp;
        return nullptr;
    }

    //! Phase timing without host synchronisation (measurement runs: HyperParam::profile == 2): the launches queued
    //! between phase_begin(tag) and the matching phase_end() are bracketed by device events on the backend's
    //! stream; phase_collect() waits for the stream and adds the elapsed seconds of every closed bracket to
    //! acc[tag] / the number of brackets to cnt[tag].  Brackets nest.  Backends without events do nothing.
    //! Dense LU with partial pivoting for the small general systems of graphs on the vector interpreter (their
    //! Jacobians may have a zero diagonal -- a transpose, a random sparse input map -- which the multifrontal
    //! solver's static pivoting does not serve and the reference's PARDISO does): lu (n x n, row-major) is filled
    //! from the CSR matrix and factored in place, piv[c] = the row swapped into position c; status[0] receives the
    //! smallest |pivot| (0: singular), status[1] the largest.
    virtual void dense_lu_factor(const CsrDev& A, double* lu, int32_t* piv, double* status) = 0;
    //! x = A^-1 b with the factors above (b and x may alias)
    virtual void dense_lu_solve(int64_t n, const double* lu, const int32_t* piv, const double* b, double* x) = 0;
    //! one pass of a vector-graph program (vecprog.h): mode EVAL0 / COEFF (xin = the placeholder's values of this
    //! order, (B, idim) in device memory), BIAS, GRAD
    virtual void run_vec_pass(const struct VecProgDev& P, int mode, int order, const double* xin) = 0;
    //! kernel launches issued so far (0 for backends that do not launch kernels)
    virtual int64_t launch_count() const { return 0; }
    virtual void phase_begin(const char* tag) { (void)tag; }
    virtual void phase_end() {}
    virtual void phase_collect(std::map<std::string, double>& acc, std::map<std::string, double>* cnt) {
        (void)acc; (void)cnt;
    }

    //! average duration in ms of `reps` back-to-back launches of one kernel,
    //! measured with device events on the backend's stream (bench.py roofline).
    //! kernel: 0 = taylor pass (mode, order given), 1 = spmv, 2 = pcg spmv+dot.
    virtual double time_kernel(int kernel, int reps, const ProgramDev* P, int mode, int order,
                               const CsrDev* A, const double* x, double* y) {
        (void)kernel; (void)reps; (void)P; (void)mode; (void)order; (void)A; (void)x; (void)y;
        return -1.0;
    }

    //! the two reductions of the per-order ANM sanity check in one launch / one
    //! synchronisation: out[0] = allclose_excess(a, b, eps) over n entries,
    //! out[1] = x . y over n1 entries  (libsanm/anm.cpp:271-285)
    virtual void sanity_reduce(size_t n, const double* a, const double* b, double eps, size_t n1,
                               const double* x, const double* y, double out[2]) {
        out[0] = allclose_excess(n, a, b, eps);
        out[1] = dot(n1, x, y);
    }
    //! the whole per-order sanity check (libsanm/anm.cpp:271-285) in one pass over the matrix:
    //! out[0] = allclose_excess(A xi, -ti*grad_t - bi, eps) over the n rows, out[1] = x1 . xi over n1
    //! entries.  `tmp0` / `tmp1` (n doubles each) are scratch for backends that take the unfused route.
    virtual void sanity_check(const CsrDev& A, const double* xi, double ti, const double* grad_t,
                              const double* bi, double eps, size_t n1, const double* x1, double* tmp0,
                              double* tmp1, double out[2]) {
        spmv(A, xi, tmp0);
        axpby(A.n, -ti, grad_t, -1.0, bi, tmp1);
        sanity_reduce(A.n, tmp0, tmp1, eps, n1, x1, xi, out);
    }
    // Queue-only forms used inside the order loop (no host synchronisation): scalars flow from kernel to
    // kernel through device-visible memory and the host reads them once, after the last order.
    //! *out = x . y  (out: device or pinned host memory)
    virtual void dot_async(size_t n, const double* x, const double* y, double* out) = 0;
    //! t = *num * scale;  out[0..n) = -t * x - y;  out[n] = t;  *t_out = t   (anm.cpp:258-264)
    //! next_coeff and the COEFF(order) + BIAS(order + 1) pass that follows it as ONE launch: the pass forms x_i in its
    //! gather, extra workgroups store it (and carry a pending Gram-Schmidt scaling phase).  false: not offered for
    //! this program here -- the caller queues the two separately.  Same arithmetic either way.
    virtual bool run_pass_next_coeff(const ProgramDev&, int /*order*/, const NextCoeff&) { return false; }
    virtual void next_coeff_async(const NextCoeff& nc) = 0;
    //! Order 1 without a host round trip (anm.cpp:228-245): t_1 = 1 / sqrt(*xgt2 + 1) formed on the device from the
    //! reduction's result, out = -t_1 xg - xb (n entries), out[n] = t_1; t_1 also to sc[0] (device) and *t_host
    //! (pinned).  false: not offered -- the caller takes the reductions to the host.
    virtual bool x1_async(size_t /*n*/, const double* /*xgt2*/, const double* /*xg*/, const double* /*xb*/, double* /*out*/,
                          double* /*sc*/, double* /*t_host*/) {
        return false;
    }
    // One classical Gram-Schmidt step of the Pade basis (pade.cpp:36-70) is three queued kernels; scalars stay
    // in device memory.  A vector whose norm underflows (sqrt(norm2) < eps) is normalised a second time by
    // its own norm: that rare fix-up is applied by the NEXT step's projection kernel, which reads the vector
    // anyway (gs_renorm_async for the last one).
    //! out[j] = x . ys[j].  First, if last_norm2 != null and sqrt(*last_norm2) < eps: ys[nvec-1] *= 1/sqrt(*last_nn2)
    virtual void multi_dot_async(size_t n, const double* x, int nvec, double* const* ys, double* out,
                                 const double* last_norm2, const double* last_nn2, double eps) = 0;
    //! out = x - sum_{j >= first} coefs[j] * qs[j];  *norm2 = out . out   (coefs, norm2: device memory)
    virtual void gs_update_async(size_t n, const double* x, int nvec, const double* const* qs,
                                 const double* coefs, int first, double* out, double* norm2) = 0;
    //! v *= 1 / max(sqrt(*norm2), eps);  *nn2 = v . v  (of the scaled vector)
    virtual void scale_rsqrt_async(size_t n, double* v, const double* norm2, double eps, double* nn2) = 0;
    //! if sqrt(*norm2) < eps: v *= 1/sqrt(*nn2)
    virtual void gs_renorm_async(size_t n, double* v, const double* norm2, const double* nn2, double eps) = 0;
    //! sanity_check with t_i read from xi[n]; out2 as in sanity_check
    virtual void sanity_check_async(const CsrDev& A, const double* xi, const double* grad_t, const double* bi,
                                    double eps, size_t n1, const double* x1, double* tmp0, double* tmp1,
                                    double* out2) = 0;
    //! The checks of `nvec` orders at once: out[2q], out[2q+1] as sanity_check_async for (xs[q], bs[q]).  The
    //! matrix is read once for all of them instead of once per order (the order loop examines the results after
    //! its last order anyway).  `out` as in sanity_check_async (memory the host can read after a sync).
    virtual void sanity_check_batch_async(const CsrDev& A, int nvec, const double* const* xs, const double* grad_t,
                                          const double* const* bs, double eps, size_t n1, const double* x1,
                                          double* tmp0, double* tmp1, double* out) {
        for (int q = 0; q < nvec; ++q)
            sanity_check_async(A, xs[q], grad_t, bs[q], eps, n1, x1, tmp0, tmp1, out + 2 * q);
    }
    //! like allclose_excess for check_t0v_match (anm.cpp:343-360): a + b*t0 vs 0
    virtual double t0v_excess(size_t n, const double* fx, const double* v, double t0, double tol) = 0;
};

//! factory of the one backend linked into the library
Backend* make_backend(int device);

}  // namespace sanm_hip
