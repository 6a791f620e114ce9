// Device-visible structures of the multifrontal solver (see multifrontal.h).
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

namespace sanm_hip {

constexpr int MF_NB = 32;  // panel / tile width
// widest pivot block of a level whose forward boundary operator is kept transposed (Level::fwd_t; SANM_MF_FWD_T_MAX_K)
constexpr int kFwdTMaxK = 256;
// Static pivot perturbation, as PARDISO does for unsymmetric matrices (iparm[9] = 13, the setting the reference's
// pardisoinit leaves in place, libsanm/sparse_solver.cpp:107-127): a pivot smaller in magnitude than
// MF_PIVOT_EPS * max|a_ij| is replaced by that value with the pivot's sign and counted; a factorisation with
// perturbed pivots is followed by iterative refinement in every solve (anm.cpp: DirectSolver).
constexpr double MF_PIVOT_EPS = 1e-13;

struct MfFrontDev {
    int64_t off;        // offset of the dense ld*ld augmented front (row-major) in the front storage
    int32_t k, m;       // pivots, front size
    int32_t ld;         // m + k.  Row / column order of the dense front: [pivot P (k) | augmentation A (k) |
                        // boundary B (m-k)].  A starts as identity blocks F[P,A] = F[A,P] = I; the LU of the
                        // leading 2k x 2k block leaves F[P,A] = L11^-1 and F[A,P] = U11^-1; one GEMM pass then
                        // writes F[B,B] -= L21 U12 (Schur complement), F[B,A] = -L21 L11^-1, F[A,B] = -U11^-1 U12
    int64_t tmp_off;    // offset of this front's GEMM workspace (2*k*(m-k) doubles) in tmp_store
    int32_t own_start;  // first own variable (new numbering)
    int32_t bnd_off;    // offset into bnd_idx (m-k entries, new numbering, ascending)
    int32_t parent;     // -1 for roots
    int32_t rel_off;    // offset into rel (m-k entries): position of each boundary row in the parent front
    int32_t nch;        // number of children
    int32_t inbox_off;  // offset of this front's inbox (nch x m doubles, [child slot][logical row]) in inbox_store
};

// Everything the numeric kernels need, resident on the device.
struct MfDev {
    int64_t n, nnzA;
    int32_t nr_front, nr_level;
    const MfFrontDev* fronts;
    const int32_t* level_fronts;  // front ids grouped by level, by decreasing k inside a level
    const MfFrontDev* lfronts;    // fronts[level_fronts[i]]: the descriptors themselves in level order
    const int32_t* upd_dst;       // parallel to bnd_idx: inbox_store slot (in the parent's inbox) of each boundary row
    double* inbox_store;          // solve workspace: children's update entries, one slot per (child, parent row);
                                  // slots no child writes stay zero forever
    const int32_t* bnd_idx;
    const int32_t* rel;
    const int32_t* perm;          // original -> new numbering
    const int32_t* own_front;     // new index -> owning front (identity init of the augmentation)
    // scatter of A: front_store[a_dst[p]] = A.val[p]
    const int64_t* a_dst;
    // extend-add: child lists per level and round
    double* front_store;          // sum of m*m
    double* work;                 // n doubles (permuted rhs / solution)
    double* work2;                // n doubles (forward-solved vector z)
    double* tmp_store;            // per-level workspace: L11^-1 F12 (k x b) and F21 U11^-1 (b x k) per front
    int32_t* status;              // [0]: number of perturbed pivots
    double* piv_amax;             // max |a_ij| of the matrix being factored: pivots below 1e-13 times that are
                                  // perturbed (MF_PIVOT_EPS)
    int64_t front_store_size;
};

// Operand of the device GEMMs: element (i, j) at p[i*ld + j] inside (rows, cols), else 0.  The index maps are read
// by the GEMMs of the merged top block only: column j of an A operand at cidx[j], row i of a B operand at ridx[i].
struct MatView {
    const double* p;
    int ld, rows, cols;
    const int32_t* ridx = nullptr;
    const int32_t* cidx = nullptr;
};
// one product of the merged top block (mf_kernels.h, top_gemm_kernel): C (M x N) = [C +] A (M x K) B (K x N)
struct TopGemm {
    MatView A, B;
    double* C;
    int32_t ldc, M, N, K, acc;
};
constexpr int MF_TOP_MAXF = 8;

// one strided block copy of a batch (Backend::copy2d_batch): rows x cols doubles from src_base[src + i * lds + j] to
// dst_base[dst + i * ldd + j]
struct MfCopy2D {
    int64_t src, dst;
    int32_t rows, cols, lds, ldd;
};

// Host-side schedule (what to launch, in which order).  Within a level the
// fronts are sorted by decreasing pivot count, so the fronts that still have a
// panel p form a prefix of the level's list.
struct MfSchedule {
    struct Level {
        int32_t front_begin, front_end;  // into level_fronts
        int32_t nr_panel;
        int32_t max_m, max_k, max_b;
        int64_t sum_m, sum_k;            // rows of the level's forward / backward solve
        // Two-phase level (multifrontal.cpp): the fronts keep -L21 = -F[B,P] U11^-1 and -U12 = -L11^-1 F[P,B] in the
        // F[B,A] / F[A,B] slots instead of the products with the pivot block's inverses (2 k^2 b flops per front not
        // done); a solve sweep over the level is then two dependent launches (pivot block, then boundary block).
        bool two_phase = false;
        // Transposed forward operator (round 5): the boundary block of the forward sweep, F[B,A] = -L21 L11^-1 (b rows of
        // k entries at a stride of 2k + b), is written TRANSPOSED into the F[P,B] slot, which is dead once the
        // triangular products have read it: k long rows of b entries.  The forward kernel then gives a boundary row to
        // a thread (coalesced loads down the columns, no cross-lane reduction) instead of a row of a few dozen entries to
        // a lane group.  Levels of short pivot blocks (max_k <= kFwdTMaxK) that are not two-phase.
        bool fwd_t = false;
        std::vector<int32_t> panel_cnt;  // number of fronts with k > p*NB
        std::vector<int32_t> front_k;    // pivot counts of the level's fronts in launch order (decreasing)
        // The 64 x 64 tiles of the level's two GEMM passes as flat lists (device; two words per tile: the front's
        // position in the level, which << 30 | ti << 15 | tj): a launch of exactly the tiles that exist.  The grids of
        // rounds 1-4 were boxes (largest tile count of the level)^2 x fronts x products -- on the middle levels of a
        // big tree 90-95 % of their workgroups found nothing to do, and there were enough of them to BE the launch
        // time (mf_kernels.h, gemm1_list_kernel).
        const uint32_t* g1_tiles = nullptr;
        const uint32_t* g2_tiles = nullptr;
        int32_t n_g1 = 0, n_g2 = 0;
        // extend-add rounds: round r holds the r-th child of every front of the
        // level; [begin,end) into ea_children
        std::vector<std::pair<int32_t, int32_t>> ea_rounds;
        std::vector<int32_t> ea_max_b;   // max boundary size of the children of a round
    };
    std::vector<Level> levels;
    const int32_t* ea_children = nullptr;  // device
    // Subtree-to-rank distribution (stage 1 of DESIGN.md section 7; multifrontal.cpp).  The elimination tree is cut
    // into subtrees, each owned by one rank; the fronts above the cut (`top`) are replicated.  This rank's level list
    // holds its own subtrees' fronts first -- levels [0, cut) -- then the top fronts -- levels [cut, size) --, and
    // three exchanges (sums over ranks in which every entry has exactly one non-zero contributor, i.e. gathers) tie
    // the ranks together:
    //   factor, between the two parts: the Schur complements F[B,B] of all cut roots (packed into `stage`);
    //   forward solve, between the two parts: the cut roots' update rows in their parents' inboxes;
    //   backward solve, at the end: the solution entries of the subtrees' pivots (the permuted vector `work`, the
    //   ranges this rank does not speak for zeroed first).
    struct Dist {
        bool enabled = false;
        int32_t rank = 0, world = 1;
        int32_t cut = 0;
        int64_t schur_doubles = 0, inbox_doubles = 0;
        const MfCopy2D* schur_pack = nullptr;    // device; own cut roots: front_store -> stage
        const MfCopy2D* schur_unpack = nullptr;  // the other ranks' cut roots: stage -> front_store
        int32_t n_schur_pack = 0, n_schur_unpack = 0, schur_max_b = 0;
        const MfCopy2D* inbox_pack = nullptr;    // inbox_store -> stage, one row per cut root
        const MfCopy2D* inbox_unpack = nullptr;
        int32_t n_inbox_pack = 0, n_inbox_unpack = 0, inbox_max_m = 0;
        std::vector<std::pair<int32_t, int32_t>> zero_ranges;  // [begin, end) of `work` before the last exchange
        double* stage = nullptr;                               // device: max(schur_doubles, inbox_doubles)
        // what this rank factors: its subtrees and the replicated top (flops as Multifrontal::factor_flops counts them)
        double flops_own = 0, flops_top = 0;
        double imbalance = 1;  // largest subtree load of a rank over the mean
        std::vector<double> rank_flops;  // subtree flops per rank (the same table on every rank)
        std::vector<double> rank_nnz;    // factor entries of each rank's subtrees; nnz_top: of the replicated top
        double nnz_top = 0;
        int32_t nr_front_own = 0, nr_front_top = 0, nr_subtree = 0, nr_subtree_own = 0;
    } dist;
    // The root and the fronts of the level below it as one dense operator (device back end; mf_kernels.h).
    // Fronts in block order: the level below the root, then the root; off[] = first row of each in the block.
    struct Top {
        bool enabled = false;
        int32_t nf = 0, n = 0;
        int32_t off[MF_TOP_MAXF + 1] = {};
        double* M = nullptr;                 // device: n x n
        const TopGemm* gemms = nullptr;      // device: the products, stage by stage
        int32_t stage_begin[3] = {};         // [stage_begin[s], stage_begin[s + 1]) into gemms, two stages
        int32_t stage_dim[2] = {};           // largest M / N of a stage (grid size)
        bool indexed = false;                // some operand goes through an index map (MatView::ridx / cidx)
        // right-hand side of the block, entry i: work[wsrc[i]] + sum_s inbox_store[ell[s * n + i]], s < W (lists
        // padded with a slot that stays zero): the fronts' own inboxes and, for the root's rows, the boundary
        // rows of the block's other fronts, in a fixed order
        const int32_t* wsrc = nullptr;
        const int32_t* ell = nullptr;
        int32_t W = 0;
        // The block's solution goes to work[n + row], not over the right-hand side the other workgroups of the
        // launch are still reading: what reads it afterwards -- the boundary lists of the levels below and the
        // permutation on the way out -- uses these copies of MfDev::bnd_idx / perm with the block's variables
        // redirected there.
        const int32_t* bnd_x = nullptr;
        const int32_t* perm_x = nullptr;
    } top;
};

}  // namespace sanm_hip
