// Host-side computing graph and its compilation into a device program.
//
// Mirrors the operator-construction API of the reference (libsanm/oprs.h:14-103,
// oprs.cpp:16-102): the same operators with the same argument meaning, but
// variables are small integer ids instead of VarNode pointers so the graph can
// cross the C ABI.  `compile()` plays the role of TaylorCoeffProp's constructor
// (libsanm/symbolic.cpp:142-160): it topologically sorts the operators needed
// by the output, counts readers, and lays the per-variable state out in one
// device arena (see program.h).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "backend.h"
#include "program.h"

namespace sanm_hip {

struct SanmError {
    int code;
    std::string msg;
};
constexpr int SANM_OK = 0;
constexpr int SANM_ERR_ASSERT = 1;     // SANMAssertionError (libsanm/utils.h:34-50)
constexpr int SANM_ERR_NUMERICAL = 2;  // SANMNumericalError
constexpr int SANM_ERR_HIP = 3;
constexpr int SANM_ERR_UNSUPPORTED = 4;

[[noreturn]] void sanm_throw(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define sanm_check(cond, ...)                                        \
    do {                                                             \
        if (!(cond)) ::sanm_hip::sanm_throw(SANM_ERR_ASSERT, __VA_ARGS__); \
    } while (0)

struct GraphVar {
    int size;      // element count per batch item: 1, 3 or 9 on the per-tet path; anything up to 256 on the vector path
    int producer;  // index into Graph::ops
    int out_idx;
    int rows = 0, cols = 0;  // (batch, rows, cols); cols = 0: a (batch, rows) tensor (batched vector / scalar)
    // a constant of nine values given without a shape: (3,3) by default -- the (T,3,3) constants of the FEA graphs --
    // but it takes the shape of a flat (batch, 9) operand it meets in an elementwise operator (the reference's
    // tensors carry their own shape; here a flat constant must not turn a graph over vectors of length 9 into one
    // over matrices)
    bool soft9 = false;
    bool is_matrix() const { return cols > 0; }
};

struct GraphOp {
    OpType type;
    int flags = 0;
    std::vector<int> in, out;
    std::vector<double> coeffs;  // LINCOMB
    double bias = 0;             // LINCOMB
    double exponent = 0;         // POW
    std::vector<double> value;   // CONSTANT, row-major (T, size)
    int64_t batch = 0;           // CONSTANT
    int begin = 0;               // SLICE: first element taken (resolved, misc.cpp:104-133)
};

class Graph {
public:
    std::vector<GraphOp> ops;
    std::vector<GraphVar> vars;

    int placeholder();
    //! a (batch, size) vector input: graphs over it run on the vector interpreter (vecprog.h)
    int placeholder_vector(int size);
    //! a (batch, rows, cols) matrix input; sizes other than 3 x 3 run on the vector interpreter too
    int placeholder_matrix(int rows, int cols);
    int constant(const double* val, int64_t batch, int size);
    //! constant of shape (batch, rows, cols)
    int constant_matrix(const double* val, int64_t batch, int rows, int cols);
    //! x[:, begin:end] (SymbolVar::slice, oprs.h:60; misc.cpp:104-231): axis 1, stride 1 like the reference's
    //! implementation; has_begin / has_end = 0 stand for None
    int slice(int x, int axis, int has_begin, int begin, int has_end, int end, int stride);
    //! concatenation along axis 1 (misc.cpp:233-331)
    int concat(int n, const int* vars, int axis);
    int linear_combine(int n, const double* coeffs, const int* vars, double bias);
    int multiply(int a, int b);
    int pow(int x, double e);
    int log(int x);
    int reduce_sum(int x, int axis);
    int batched_matmul(int a, int b);
    int batched_mat_inv_mul(int x, int a /* -1: identity */, bool is_left);
    int batched_det(int x);
    int batched_transpose(int x);
    int batched_mul_eye(int x, int dim);
    void batched_svd_w(int x, bool require_rotation, int out[3]);

private:
    struct Shape {
        int rows, cols;
        bool soft9 = false;
    };
    int add(GraphOp op, std::initializer_list<Shape> out_shapes);
    Shape shape(int v) const { return {vars[v].rows, vars[v].cols, vars[v].soft9}; }
    Shape elemwise_shape(Shape a, Shape b) const;
    void chk(int v) const;
};

// A graph compiled for T tets and a maximum expansion order, resident on the
// device.  Owns the arena.
class Program {
public:
    //! T: tets handled by this program.  When the batch is sharded (one rank of a
    //! tet-sharded run) the program covers tets [tet_begin, tet_begin + T) of
    //! T_global and batched constants are sliced accordingly (ConstantOprMeta
    //! under ParallelTaylorCoeffProp, libsanm/oprs/misc.cpp:51-72).
    //! tet_order (T_global entries, optional): the program's tet e is the caller's tet tet_order[e] -- batched
    //! constants and the rows of remap_in are read through it (the driver's spatial renumbering, anm.cpp)
    //! full_history: keep the series of every variable (what sanm_taylor_get_var exposes, like
    //! VarNodeExeCtx::coeffs of the reference); otherwise only the series some convolution reads back
    Program(Backend* be, const Graph& g, int out_var, int64_t T, int max_order,
            int64_t tet_begin = 0, int64_t T_global = -1, bool full_history = true,
            const int64_t* tet_order = nullptr);
    ~Program();
    Program(const Program&) = delete;

    //! attach the remap_in table (ELL), converting flattened AoS output
    //! indices e*9+c (fea/mesh_template.h:73-110) to the SoA layout
    void set_remap_in(int64_t n_in, const uint64_t* rowptr, const uint64_t* idx,
                      const double* coef);
    //! the host half of set_remap_in (no backend: any thread) and the device half (the backend's owner thread)
    struct RemapInHost {
        std::unique_ptr<uint32_t[]> idx;
        std::unique_ptr<double[]> coef;
        size_t tab = 0;
        int nslot = 1;
        bool packed = false;
        int64_t n_in = 0;
    };
    RemapInHost prepare_remap_in(int64_t n_in, const uint64_t* rowptr, const uint64_t* idx, const double* coef) const;
    void set_remap_in(RemapInHost&& table);  // rowptr indexed by GLOBAL output element (of the caller's numbering)

    ProgramDev dev() const { return m_dev; }
    //! HIP source of the four pass kernels with this program's records as compile-time constants
    std::string spec_source() const;
    int64_t T() const { return m_dev.T; }
    int64_t Tpad() const { return m_dev.Tpad; }
    int max_order() const { return m_dev.max_order; }
    int placeholder_var() const { return m_placeholder_var; }
    int local_var(int graph_var) const { return m_var_map.at(graph_var); }
    const std::vector<VarDesc>& vars() const { return m_vars; }
    int64_t n_in() const { return m_n_in; }

    //! device pointer of the output's order-0 value (after EVAL0) / order-k bias (after BIAS(k)), tet-major [T][9]
    const double* out_coef0() const { return m_dev.arena + m_dev.out_aos; }
    const double* out_bias() const { return m_dev.arena + m_dev.out_aos; }
    const double* placeholder_jac() const { return m_dev.arena + m_vars[m_placeholder_var].jac; }
    //! copy a coefficient (or bias when order < 0) of a graph var to host, AoS (T,size)
    void download_var(int graph_var, int order, double* dst) const;
    void download_jacobian(double* dst) const;  // (T, odim, 9)
    size_t arena_bytes() const { return m_arena_doubles * sizeof(double); }
    //! per pow operator with an exponent other than 2: (arena offset of the zero flag -- one double shared by all of
    //! them --, exponent).  The order-0 pass writes 1 there for "0^p with p not an integer above 1/2" (SANMNumericalError in the reference,
    //! analytic_unary.cpp:115-120) and 2 for "integer power of a series through zero beyond the order the device
    //! path carries" (32).
    struct PowFlag {
        int64_t off;
        double exponent;
    };
    const std::vector<PowFlag>& pow_flags() const { return m_pow_flags; }
    double* arena_dev() const { return m_dev.arena; }
    //! run-time specialisation of the pass kernels (constructor): host seconds it took and where the code object came
    //! from -- 0: not specialised, 1: the process-wide cache, 2: the on-disk cache, 3: compiled now, 4: built ahead of time (embedded)
    double jit_seconds = 0;
    int jit_source = 0;

private:
    Backend* m_be;
    ProgramDev m_dev{};
    std::vector<PowFlag> m_pow_flags;
    int64_t m_pow_flag_off = -1;
    std::vector<OpDesc> m_ops;
    std::vector<VarDesc> m_vars;
    std::vector<int> m_var_map;  // graph var -> local var (-1 if unused)
    int m_placeholder_var = -1;
    int64_t m_arena_doubles = 0, m_jac_begin = 0, m_jac_end = 0;
    int64_t m_n_in = 0;
    int64_t m_tet_begin = 0;
    std::vector<int64_t> m_tet_order;  // this program's tets in the caller's numbering (empty: tet_begin + e)
    void* m_d_ops = nullptr;
    void* m_d_vars = nullptr;
    void* m_d_lc_params = nullptr;
    void* m_d_rin_idx = nullptr;
    void* m_d_rin_coef = nullptr;
};

}  // namespace sanm_hip
