// Device-visible description of a compiled computing graph ("program").
//
// The reference interprets its graph operator by operator on host tensors
// (libsanm/symbolic.cpp:162-289).  Here the topologically sorted graph is
// compiled once into a flat array of OpDesc / VarDesc records; a single HIP
// kernel launch (one lane per tet) then runs a whole pass (order-0 eval,
// reverse-mode Jacobian, order-k bias, order-k coefficient) over it.
//
// Storage: one arena of doubles in HBM.  Every per-tet quantity is stored
// SoA: element c of tet e of a tensor at arena offset `off` lives at
// off + c*Tpad + e, so the 64 lanes of a wavefront read 512 contiguous bytes.
#pragma once
#if !defined(__HIPCC_RTC__)
#include <cstdint>
#else  // run-time compilation has no standard headers; its built-ins cover what is used
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
#endif

namespace sanm_hip {

enum OpType : int32_t {
    OP_PLACEHOLDER = 0,  // libsanm/oprs/misc.cpp:13-44
    OP_CONSTANT = 1,     // libsanm/oprs/misc.cpp:48-100
    OP_LINCOMB = 2,      // libsanm/oprs/elem_arith.cpp:42-124
    OP_MULTIPLY = 3,     // libsanm/oprs/elem_arith.cpp:128-217
    OP_LOG = 4,          // libsanm/analytic_unary.cpp:13-34
    OP_POW = 5,          // libsanm/analytic_unary.cpp:36-139
    OP_REDUCE_SUM = 6,   // libsanm/oprs/reduce.cpp:11-102 (axis=-1)
    OP_MATMUL = 7,       // libsanm/oprs/linalg.cpp:339-418
    OP_MATINVMUL = 8,    // libsanm/oprs/linalg.cpp:67-217
    OP_DET = 9,          // libsanm/oprs/linalg.cpp:221-282
    OP_TRANSPOSE = 10,   // libsanm/oprs/linalg.cpp:286-335
    OP_MULEYE = 11,      // libsanm/oprs/linalg.cpp:422-479
    OP_SVDW = 12,        // libsanm/oprs/linalg.cpp:483-615 (pw_mode)
    // vector graphs only (vecprog.h): batch-1-style (batch, n) tensors
    OP_SLICE = 13,       // libsanm/oprs/misc.cpp:104-231 (axis 1, stride 1)
    OP_CONCAT = 14,      // libsanm/oprs/misc.cpp:233-331 (axis 1)
};

enum PassMode : int32_t {
    PASS_EVAL0 = 0,  // infer_shape_eval_bias, symbolic.cpp:172-176
    PASS_GRAD = 1,   // ensure_jacobian reverse sweep, symbolic.cpp:206-247
    PASS_BIAS = 2,   // compute_next_order_bias, symbolic.cpp:249-289
    PASS_COEFF = 3,  // push_xi at order >= 1, symbolic.cpp:177-178
    // COEFF(k) then BIAS(k+1) in one launch: the two are back to back in the order loop.  Only among the kernels
    // compiled per graph (ProgramDev::spec_id >= 0).
    PASS_COEFF_BIAS = 4,
};

constexpr int OP_FLAG_IS_LEFT = 1;         // MATINVMUL: Y X = A
constexpr int OP_FLAG_USE_IDENTITY = 2;    // MATINVMUL: A = I
constexpr int OP_FLAG_REQUIRE_ROT = 4;     // SVDW: force det(W) = +1
constexpr int OP_FLAG_SVDW_FULL = 8;       // SVDW: U or S is read: the full U, S, W recurrences (oprs/linalg.cpp:533)
constexpr int OP_FLAG_SVDW_GU = 16;        // SVDW, reverse sweep: the gradient slot of U / S / W is live (the output
constexpr int OP_FLAG_SVDW_GS = 32;        // has a reader or is the graph output)
constexpr int OP_FLAG_SVDW_GW = 64;
constexpr int OP_FLAG_VECTOR = 128;        // PLACEHOLDER: a (batch, n) vector, not a (T,3,3) matrix (vecprog.h)
constexpr int MAX_OP_IN = 4;

struct VarDesc {
    int64_t coef;  // arena offset of coefficient 0; order k at coef + k*size*Tpad
    int64_t bias;  // arena offset of cur_order_bias
    int64_t jac;   // arena offset of the Jacobian, tet-major [Tpad][odim][size]: the placeholder only; -1 otherwise
    int32_t size;  // 1 (batched scalar), 3 (singular values) or 9 (3x3)
    int32_t is_const;  // coefficients of order >= 1 are identically zero
    int32_t cur;       // offset (in doubles) of the current-order value in the per-lane scratch
    int32_t hist;      // coefficients of order >= 1 are kept in the arena: some convolution reads them back
                       // (or the program keeps every series for the operator-level API)
    int32_t alias;     // >= 0: this variable is the batched_transpose of variable `alias`, whose series is kept: the
                       // fused convolution loop reads that series (one load serves both); -1 otherwise
    int32_t pad_;
};

struct OpDesc {
    int32_t type, nin, nout, flags;
    int32_t in[MAX_OP_IN];
    int32_t out[3];
    int32_t grad_zero;  // GRAD pass: bit i set = this operator is the first (in reverse order) to accumulate
                        // into the gradient slot of input i and clears it first
    double p[MAX_OP_IN + 2];  // LINCOMB: coeffs then bias at p[MAX_OP_IN]; POW: p[0]=exponent
    int64_t aux[4];           // arena offsets of per-operator scratch (see tet_ops.h)
    // fused convolution loop of the kernels compiled per graph (tet_ops.h, conv_term): this operator's sums live at
    // [conv_off, conv_off + conv_n) of the pass's accumulator array; conv_n = 0: the operator has no convolution
    int32_t conv_off, conv_n;
};

// remap_in as an ELL table: for output element (tet e, comp c) and slot s,
// idx/coef at [(s*9 + c)*Tpad + e]; unused slots carry coef 0.
struct RemapInDev {
    const uint32_t* idx;
    // null: every coefficient of the table is +1, -1 or 0 (the edge vectors of a tet: x_j - x_0) and sits in the two
    // top bits of its index word -- 00: +1, 10: -1, 11: 0 (empty slot) --: 4 bytes per entry instead of 12
    // (SANM_RIN_DECODE below; Program::set_remap_in packs when it can)
    const double* coef;
    int32_t nslot;
    // next_coeff fused into the gather (Backend::run_pass_next_coeff): with xg set, the gathered vector is not read
    // from memory but formed on the way, x[j] = -t * xg[j] - xvec[j] -- the x_i of the order loop (anm.cpp:261-264)
    // with xvec = A^-1 b_i and xg = A^-1 g_t; same two roundings as the kernel that stores x_i
    const double* xg = nullptr;
    double t = 0;
};

// index and coefficient of a packed remap_in word (RemapInDev::coef == nullptr)
#define SANM_RIN_INDEX(w) ((w) & 0x3fffffffu)
#define SANM_RIN_COEF(w) (((w) >> 30) == 0u ? 1.0 : (((w) >> 30) == 2u ? -1.0 : 0.0))

struct ProgramDev {
    const OpDesc* ops;
    const VarDesc* vars;
    double* arena;
    int32_t nops;
    int32_t out_var;   // the graph output (3x3)
    int32_t odim;      // size of the output var (9)
    int32_t max_order;
    int32_t cur_size;  // doubles of per-lane scratch (sum of the sizes of the non-constant vars)
    int32_t desc_lines;  // 64-byte lines of the block holding ops and vars (a multiple of 8)
    int32_t conv_total;  // accumulators of the fused convolution loop (sum of OpDesc::conv_n)
    int64_t T, Tpad;
    // the graph output as remap_out gathers it: tet-major [T][9] (order-0 value after EVAL0, order-k bias after
    // BIAS(k)).  A row of remap_out takes 3 entries of each adjacent tet and the 3 rows of a vertex the same tets:
    // tet-major they share cache lines, component-major every entry sits in a line of its own.
    int64_t out_aos;
    RemapInDev rin;
    // coefficients and bias of the linear combinations, MAX_OP_IN + 2 doubles per LINCOMB operator in operator order:
    // what the kernels compiled per graph read in place of OpDesc::p (tet_ops.h, SANM_LC_PARAM)
    const double* lc_params;
    // host side only: handle of the kernels compiled for this very program (Backend::specialize), -1 = none
    int32_t spec_id;
};

}  // namespace sanm_hip
