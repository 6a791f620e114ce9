// Device-visible structures of the multifrontal solver (see multifrontal.h).
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

namespace sanm_hip {

constexpr int MF_NB = 32;  // panel / tile width
constexpr int MF_ZERO_ROWS = 16;  // rows of a front per workgroup of zero_kernel
// F[B,B] -= tmpL tmpU of a big front: the interior of the block is computed in 128 x 64 tiles (mf_kernels.h,
// gemm2_tall_list_kernel), the 64 x 64 tile rows at its lower / right edge and everything on smaller fronts by the
// GEMM-2 pass.  One predicate for the host's tile lists and the kernels: is the 64 x 64 tile (ti, tj) of a b x b
// Schur complement with k pivots part of a tall tile?
constexpr int MF_GT = 64;  // GEMM tile edge
constexpr int MF_TALL_MIN_K = 512, MF_TALL_MIN_B = 1024;
#if defined(__HIPCC__)
#define MF_HD __host__ __device__
#else
#define MF_HD
#endif
MF_HD inline bool mf_gemm2_is_tall(int k, int b, int ti, int tj) {
    return k >= MF_TALL_MIN_K && b >= MF_TALL_MIN_B && ((ti & ~1) + 2) * MF_GT <= b && (tj + 1) * MF_GT <= b;
}
// The tall tiles of a level go out as a flat list ordered for the eight L2s: workgroup n runs on the XCD n % 8 (observed
// round-robin placement: speed only), so the list interleaves eight queues, each a sequence of SUPERTILES of
// MF_ST_R x MF_ST_C tall tiles -- the ~96 workgroups an XCD holds at a time then share 8 row panels of tmpL and 12
// column panels of tmpU instead of one row panel and 96 column panels.
constexpr int MF_ST_R = 8, MF_ST_C = 12;
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
    // scatter of A: front_store[a_dst[p]] = A.val[p].  Filled by the backend from the matrix's pattern when the first
    // factorisation runs (mf_scatter_slot below; MfSchedule::a_dst_ready) -- 8 bytes per entry that the analysis neither
    // computes on the host nor uploads (round 6)
    int64_t* a_dst;
    // distributed solver: 1 for the fronts this rank factors (it stores those, and the Schur blocks of the children it
    // receives from other ranks -- nothing else has a place in its front store); nullptr: every front
    const uint8_t* front_here;
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

//! where entry (i, j) of the matrix (new numbering: pi, pj) sits in the front storage: in the front that owns the smaller
//! of the two, at the positions of both among its [pivots | augmentation | boundary]
//! (-1: the front is another rank's, `here` says so)
MF_HD inline int64_t mf_scatter_slot(const MfFrontDev* fronts, const int32_t* own_front, const int32_t* bnd_idx, int32_t pi,
                                     int32_t pj, const uint8_t* here = nullptr) {
    const int32_t fi = own_front[pi < pj ? pi : pj];
    if (here && !here[fi]) return -1;
    const MfFrontDev& f = fronts[fi];
    int32_t pos[2];
    for (int w = 0; w < 2; ++w) {
        const int32_t x = w == 0 ? pi : pj;
        if (x >= f.own_start && x < f.own_start + f.k) {
            pos[w] = x - f.own_start;
            continue;
        }
        const int32_t* b = bnd_idx + f.bnd_off;
        int32_t lo = 0, hi = f.m - f.k;  // first boundary entry >= x (the analysis made sure it is x)
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (b[mid] < x) lo = mid + 1;
            else hi = mid;
        }
        pos[w] = 2 * f.k + lo;
    }
    return f.off + (int64_t)pos[0] * f.ld + pos[1];
}

// One workgroup of a level's solve launch: the front's descriptor and which block of its rows.  The box grids of the
// sweeps -- (row blocks of the level's LARGEST front) x fronts -- leave 40-60 % of the workgroups of the lower levels
// without rows (a leaf level's fronts average 54 pivots and peak at 96); a flat list of the blocks that exist, the
// descriptor inside the entry (no extra dependent load), launches exactly those (backend_hip.hip, solve_blocks).
struct MfSolveBlock {
    MfFrontDev f;
    int32_t bx, pad;
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
        const uint32_t* gt_tiles = nullptr;  // tall tiles (front or ~0u for a padding entry, tp << 15 | tj)
        int32_t n_g1 = 0, n_g2 = 0, n_gt = 0;
        // extend-add rounds: round r holds the r-th child of every front of the
        // level; [begin,end) into ea_children
        std::vector<std::pair<int32_t, int32_t>> ea_rounds;
        std::vector<int32_t> ea_max_b;   // max boundary size of the children of a round
        int32_t ea0_max_bp = 0;          // largest boundary of a front of the level that has children (round 0's gather)
    };
    std::vector<Level> levels;
    std::vector<MfFrontDev> h_lfronts;     // host copy of MfDev::lfronts (the solve's block lists are made from it)
    mutable bool a_dst_ready = false;      // MfDev::a_dst has been filled (by the first factorisation)
    const int32_t* ea_children = nullptr;  // device
    // Round 0 of the extend-add ASSIGNS the parent's F[B,B] instead of adding to a zeroed block (round 6): for boundary
    // position i of a front with children, ea_inv[bnd_off + i] = the boundary position of its FIRST child that lands there,
    // -1 for none -- a parent-side gather, so every entry of the block is written once and the block needs no zero-fill
    // (the fronts without children never read theirs: mf_kernels.h, gemm2_tile).  The rest of the first child's Schur
    // complement -- what lands in the parent's pivot rows and columns -- is added like the other children's.
    const int32_t* ea_inv = nullptr;       // device; parallel to bnd_idx
    // What the factorisation's prologue zeroes instead of the whole front storage: per front the rows P in full, the
    // columns P and A of the rows A, the columns P of the rows B (zero_kernel); flat list of (front, first row) blocks
    // of ZERO_ROWS rows each
    const int32_t* zero_blocks = nullptr;  // device; 2 words per block
    int32_t n_zero_blocks = 0;
    int64_t zero_doubles = 0;              // what that writes (statistics)
    // Tree-to-ranks distribution (multifrontal.cpp).  EVERY front has one owner.  The elimination
    // tree is mapped from the root down onto sets of ranks (proportional mapping: a node with the rank set R gives its
    // children disjoint slices of R in proportion to their subtree work); a subtree whose set is one rank belongs to it
    // entirely -- stage 0 --, the fronts above -- the top -- belong to the first rank of their set, so a separator
    // and the heaviest of its children share their owner and sibling separators are factored beside each other by
    // different ranks.  Top fronts are grouped in stages: stage(f) = max over children c of stage(c) + (owner(c) !=
    // owner(f)), at least 1; fronts of one stage on a path of the tree share their owner, so all communication sits
    // BETWEEN stages.  A rank's level list holds its own fronts only, stage after stage (stage_level[s] .. stage_level[s
    // + 1]), heights ascending inside a stage.  The exchanges (each entry of every buffer has exactly one writer, so the
    // factors and solutions are the single-rank solver's bit for bit):
    //   factor, before stage s >= 1:   the Schur complements F[B,B] of the children (of stage-s fronts) that another
    //                                  rank owns: owner(child) -> owner(parent);
    //   forward sweep, before stage s: those children's update rows in their parents' inboxes, the same way;
    //   backward sweep, after stage s: the solution entries of the stage's pivots, owner -> everyone (the levels below
    //                                  read their boundary values there);
    //   backward sweep, at the end:    the pivots of the subtrees (stage 0), owner -> everyone.
    // With a communicator that offers point-to-point transfers the first two are grouped send / receive pairs and the
    // last two grouped broadcasts in place; through the all-reduce callback of the C ABI each is a sum over a zeroed
    // staging buffer.
    struct Xfer {
        int32_t src, dst;   // ranks; dst = -1: to every rank
        int64_t off, cnt;   // doubles: [off, off + cnt) of the staging buffer (Schur, inbox) or of `work` (solution)
        int32_t src_stage;  // the stage in which src produces it (the plan's dependency model; not read by the transfer)
    };
    struct Exchange {
        std::vector<Xfer> xfers;               // the same list on every rank
        int64_t doubles = 0;                   // staging doubles (sum of cnt)
        const MfCopy2D* pack = nullptr;        // device; what this rank sends: its source -> stage
        const MfCopy2D* unpack = nullptr;      // device; what this rank receives: stage -> its destination
        int32_t n_pack = 0, n_unpack = 0, max_rows = 0, max_cols = 0;
    };
    struct Dist {
        bool enabled = false;
        int32_t rank = 0, world = 1;
        int32_t nr_stage = 1;                    // stages 0 .. nr_stage - 1
        std::vector<int32_t> stage_level;        // nr_stage + 1 entries into this rank's levels
        std::vector<Exchange> schur, inbox, sol; // per stage (entry 0 of schur / inbox unused; sol[0] = the subtrees' pivots)
        int64_t schur_doubles = 0, inbox_doubles = 0;  // totals over the stages (statistics)
        int64_t stage_doubles = 0;
        double* stage = nullptr;                 // device: the largest exchange
        // ranges of front_store this rank's fronts occupy (the factorisation zeroes these only) and of `work` it does
        // not speak for at the end of a solve (the callback form of the last exchange zeroes them first)
        std::vector<std::pair<int64_t, int64_t>> own_store;
        std::vector<std::pair<int32_t, int32_t>> zero_ranges;
        // what this rank factors: its subtrees and its fronts of the top (flops as Multifrontal::factor_flops counts
        // them); flops_top: the whole top; flops_critical: the factorisation's critical path in flops -- a rank runs its
        // stages one after the other and a stage starts when the stages that produce its incoming Schur complements are
        // done (stage_finish_time with flops as the clock)
        double flops_own = 0, flops_top = 0, flops_top_own = 0, flops_critical = 0;
        double imbalance = 1;  // largest subtree load of a rank over the mean
        std::vector<double> rank_flops;      // subtree flops per rank (the same table on every rank)
        std::vector<double> rank_top_flops;  // top flops per rank
        std::vector<double> rank_nnz;        // factor entries of each rank's fronts (subtrees and top)
        std::vector<double> stage_flops;     // nr_stage x world: flops of rank r in stage s at [s * world + r]
        std::vector<double> stage_nnz;       // the same for factor entries (the solve's bytes)
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
