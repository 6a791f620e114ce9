// Sparse direct solver for the ANM Jacobian: multifrontal LU on the device.
//
// Role in the reference: SparseSolver::prepare / solve (libsanm/sparse_solver.cpp:
// 327-421, :154-180), i.e. MKL PARDISO with mtype 11 (unsymmetric LU), one
// factorisation and `order` forward/backward substitutions per ANM step.  The
// sparsity pattern is fixed per model, so everything symbolic happens once on
// the host (this file); each step only re-runs the numeric kernels on the GPU.
//
// Method (classic multifrontal, structurally symmetric pattern, no pivoting):
//   * indistinguishable rows are merged into supervariables (the 3 dofs of a
//     vertex), the compressed graph is ordered by nested dissection: cuts
//     along the principal directions of the unknowns' coordinates when they
//     are known (a two-source graph-distance key otherwise), the separator a
//     minimum vertex cover of the cut edges refined by Fiduccia-Mattheyses
//     passes (multifrontal.cpp: NestedDissection::bisect);
//   * every dissection-tree node is a front: a dense m x m matrix whose first
//     k rows/cols are its own (pivot) variables and the rest its boundary in
//     the ancestors.  Fronts of equal height are independent and are
//     processed by the same kernel launches;
//   * factor: scatter A, extend-add the children's Schur complements, blocked
//     right-looking LU of the leading k columns (32-wide panels, one launch
//     each; large fronts with a second blocking level).  Each front carries k
//     extra identity columns and rows; the same row/column operations turn
//     them into L11^-1, -L21 L11^-1, U11^-1 and -U11^-1 U12;
//   * solve: with those blocks every level of the forward (resp. backward)
//     sweep is one dense mat-vec per front, all rows in parallel: the `order`
//     sequential solves of an ANM step are launch- and bandwidth-bound instead
//     of being chains of dependent triangular substitutions.
// The forward-FEA Jacobian is minus a Hessian (definite at stable states), so
// unpivoted LU is stable there; tiny pivots are detected and reported.
#pragma once
#include <cstdint>
#include <functional>
#include <future>
#include <memory>
#include <vector>

#include "backend.h"
#include "graph.h"
#include "mf_types.h"

namespace sanm_hip {

class Multifrontal {
public:
    //! pattern of the n x n matrix (CSR, original numbering); coords (n,3) or null.  world > 1: this rank's part of
    //! the subtree-to-rank distribution (MfSchedule::Dist): every rank runs the same analysis on the same pattern
    //! and keeps its own subtrees plus the replicated top of the tree in its schedule.
    //! defer_device: the constructor does the host analysis only and touches no backend -- it may run on a thread of
    //! its own beside the owner of the backend --; finish_device(), called by that owner, then makes the device copies.
    Multifrontal(Backend* be, int64_t n, const std::vector<uint32_t>& rowptr,
                 const std::vector<uint32_t>& col, const double* coords, int rank = 0, int world = 1,
                 bool defer_device = false);
    //! The pattern given by BLOCKS: unknowns block * v .. block * v + block - 1 (v < n / block) have one column list, the
    //! unknowns of the blocks qcol[qptr[v] .. qptr[v + 1]) (ascending; the block v itself among them) -- the Jacobian of
    //! a mesh with `block` unknowns per vertex, whose block rows the driver knows before the rows themselves (round 6:
    //! the analysis starts while the pattern of the unknowns is still being written out).  Same supervariables, ordering
    //! and tables as the constructor above on the expanded pattern; nnz = block^2 * qcol.size() entries.
    struct BlockPattern {
        int block = 3;
        std::shared_ptr<const std::vector<uint32_t>> qptr, qcol;
    };
    Multifrontal(Backend* be, int64_t n, const BlockPattern& blocks, const double* coords, int rank = 0, int world = 1,
                 bool defer_device = false);
    ~Multifrontal();
    void finish_device();
    double analysis_seconds = 0;  // wall clock of the constructor
    Multifrontal(const Multifrontal&) = delete;

    const MfDev& dev() const { return m_dev; }
    const MfSchedule& schedule() const { return m_sched; }

    // statistics of the analysis
    int64_t nnz_factors = 0;   // entries of L + U
    double factor_flops = 0;
    int64_t front_doubles = 0;
    int32_t nr_front = 0, nr_level = 0, max_front = 0, root_pivots = 0, nr_supervar = 0;
    bool used_coords = false;

private:
    //! the constructors' body: rowptr / col may be null when `blocks` describes the pattern
    void analyse(int64_t n, const std::vector<uint32_t>* rowptr, const std::vector<uint32_t>* col, const BlockPattern* blocks,
                 const double* coords, int rank, int world);
    Backend* m_be;
    MfDev m_dev{};
    MfSchedule m_sched;
    std::vector<void*> m_bufs;
    // device work of the constructor: done on the spot, or kept for finish_device()
    bool m_defer = false;
    struct DeviceOp {
        std::shared_ptr<void> keep;  // the host copy of an upload
        const void* src = nullptr;
        size_t bytes = 0;
        bool zero = false;
        std::function<void(void*)> set;  // stores the device pointer where it belongs
        bool detached = false;           // the block m_front_job asked for, if it got one
    };
    //! the front store asked for beside the analysis (Backend::alloc_detached); given back if nobody adopts it
    struct DetachedAlloc {
        Backend* be = nullptr;
        std::future<void*> job;
        ~DetachedAlloc() {
            if (job.valid())
                if (void* p = job.get()) be->free_detached(p);
        }
    } m_front_job;
    std::vector<DeviceOp> m_pending;
    void run_op(DeviceOp& op);
    template <class T>
    T* upload(const std::vector<T>& v);
    //! device copy of v into `target` (v is left empty when the copy is deferred and v was given as an rvalue)
    template <class P, class T>
    void upload_to(P& target, const std::vector<T>& v);
    template <class P, class T>
    void upload_to(P& target, std::vector<T>&& v);
    //! device copy of count elements that `keep` holds alive (no host copy when deferred)
    template <class P, class T>
    void upload_kept(P& target, std::shared_ptr<void> keep, const T* src, size_t count);
    template <class P>
    void alloc_to(P& target, size_t bytes, bool zero);
};

}  // namespace sanm_hip
