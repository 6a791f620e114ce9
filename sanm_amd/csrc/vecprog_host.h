// Host side of the vector-graph interpreter (vecprog.h): compiles a Graph whose variables are batched vectors
// (any length up to VEC_MAX_SIZE, Slice / Concat allowed) into a VecProgDev resident on the device.
#pragma once
#include <memory>
#include <vector>

#include "graph.h"
#include "vecprog.h"

namespace sanm_hip {

//! does the part of `g` that `out_var` depends on need the vector interpreter (Slice / Concat, or a variable whose
//! size is not 1, 3 or 9, or a placeholder declared as a vector)?
bool graph_is_vector(const Graph& g, int out_var);

class VecProgram {
public:
    //! B: batch size (the placeholder is (B, idim))
    VecProgram(Backend* be, const Graph& g, int out_var, int64_t B, int max_order);
    ~VecProgram();
    VecProgram(const VecProgram&) = delete;

    const VecProgDev& dev() const { return m_dev; }
    int64_t B() const { return m_dev.B; }
    int idim() const { return m_dev.idim; }
    int odim() const { return m_dev.odim; }
    int max_order() const { return m_dev.max_order; }
    //! coefficient `order` (or the current bias when order < 0) of a graph variable, (B, size) row-major
    void download_var(int graph_var, int order, double* dst) const;
    void download_out(int order, double* dst) const { download_var(m_out_graph_var, order, dst); }
    void download_jacobian(double* dst) const;  // (B, odim, idim)
    //! the raise-only error words written by the order-0 pass (0^p): returns and clears them
    void take_flags(double fl[2]);
    // device pointers for the ANM order loop: the output's order-0 value / current bias ([B][odim]), the Jacobian
    // blocks ([B][odim][idim]), the two error words
    const double* out_coef0_dev() const { return m_dev.arena + m_vars[m_dev.out_var].coef; }
    const double* out_bias_dev() const { return m_dev.arena + m_vars[m_dev.out_var].bias; }
    const double* jac_dev() const { return m_dev.arena + m_dev.jac; }
    double* flag_dev() const { return m_dev.arena + m_dev.flag; }
    size_t arena_bytes() const { return m_arena_doubles * sizeof(double); }
    //! exponents of the pow operators other than squares (for the 0^p error message)
    const std::vector<double>& pow_exponents() const { return m_pow_exponents; }

private:
    Backend* m_be;
    VecProgDev m_dev{};
    std::vector<VecVar> m_vars;
    std::vector<int> m_var_map;  // graph var -> local var
    int m_out_graph_var = -1;
    int64_t m_arena_doubles = 0;
    std::vector<double> m_pow_exponents;
    void *m_d_ops = nullptr, *m_d_vars = nullptr;
};

}  // namespace sanm_hip
