// Container-only link test of adapter/anm_hip.h: graphs built with the REFERENCE's own operator
// API -- SymbolVar and the OperatorMeta classes of libsanm/{symbolic,oprs}.cpp and oprs/{elem_arith,linalg,misc,
// reduce}.cpp, compiled from where they lie under /root/reference -- are exported by sanm::hip::export_graph (the
// adapter reads the real OperatorNodes, their metas and their private Param records) into a sanm_graph of the C ABI,
// and evaluated there (sanm_taylor_*: value, Jacobian, biases and coefficients of a Taylor series).  tests/
// test_adapter_link.py compares what this prints with the same graphs built through the Python binding.
//
// What is NOT here, and why: the reference's numerical kernels (tensor*.cpp) and AnalyticUnaryOprMeta
// (oprs/analytic_unary.cpp) need Eigen, which this image lacks; every symbol they would provide is resolved by an
// ABORTING stub generated from the linker's own list of unresolved names (test_adapter_link.py) -- nothing is
// re-implemented, nothing numerical of the reference runs -- so graphs with log / pow cannot be built here, and
// fea/material.cpp itself (it includes Eigen through fea/typedefs.h) is restated below with the same operator calls
// (fea/material.cpp:55-99) for the two energies that need neither.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "anm_hip.h"
#include "libsanm/oprs.h"

using namespace sanm;
using symbolic::SymbolVar;

namespace {
#define CHECK(call)                                                                          \
    do {                                                                                     \
        const int rc_ = (call);                                                              \
        if (rc_ != SANM_HIP_OK) {                                                            \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, sanm_hip_last_error()); \
            std::exit(1);                                                                    \
        }                                                                                    \
    } while (0)

// evaluate the exported graph: identity remap of x (T*9) onto the placeholder, orders 0..N
void run(const char* name, symbolic::VarNode* y, int T, int order, const std::vector<std::vector<double>>& xs) {
    int out = -1;
    hip::GraphPtr g = hip::export_graph(y, &out);
    const int64_t n = (int64_t)T * 9;
    std::vector<uint64_t> rp(n + 1), ix(n);
    std::vector<double> cf(n, 1.0);
    for (int64_t i = 0; i <= n; ++i) rp[i] = i;
    for (int64_t i = 0; i < n; ++i) ix[i] = i;
    sanm_sparse_desc* d = nullptr;
    CHECK(sanm_sparse_desc_create(n, n, rp.data(), ix.data(), cf.data(), &d));
    sanm_taylor_prop* p = nullptr;
    CHECK(sanm_taylor_create(g.get(), out, d, order, &p));
    int osz = 0;
    CHECK(sanm_taylor_output_size(p, &osz));
    std::vector<double> yk((size_t)T * osz), bias((size_t)T * osz), jac((size_t)T * osz * 9);
    std::printf("graph %s out_size %d\n", name, osz);
    auto dump = [&](const char* what, int k, const std::vector<double>& v) {
        std::printf("%s %d", what, k);
        for (double x : v) std::printf(" %.17g", x);
        std::printf("\n");
    };
    CHECK(sanm_taylor_push_xi(p, xs[0].data(), yk.data()));
    dump("y", 0, yk);
    CHECK(sanm_taylor_get_jacobian(p, jac.data()));
    dump("jac", 0, jac);
    for (int k = 1; k <= order; ++k) {
        CHECK(sanm_taylor_compute_next_order_bias(p, bias.data()));
        dump("bias", k, bias);
        CHECK(sanm_taylor_push_xi(p, xs[k].data(), yk.data()));
        dump("y", k, yk);
    }
    sanm_taylor_destroy(p);
    sanm_sparse_desc_destroy(d);
}
}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int T = std::atoi(argv[1]), order = std::atoi(argv[2]);
    // inputs: (order + 1) vectors of T*9 doubles from a file of whitespace-separated numbers
    std::vector<std::vector<double>> xs(order + 1, std::vector<double>((size_t)T * 9));
    {
        FILE* f = std::fopen(argv[3], "r");
        if (!f) return 2;
        for (auto& v : xs)
            for (double& x : v)
                if (std::fscanf(f, "%lf", &x) != 1) return 2;
        std::fclose(f);
    }
    CHECK(sanm_hip_init(0));
    const double mu = 1.25, lambda = 0.75;
    {  // ARAP, fea/material.cpp:84-90: P = mu (F - W), W the rotation of the polar decomposition (batched_svd_w)
        symbolic::ComputingGraph cg;
        SymbolVar F = symbolic::placeholder(cg);
        auto usw = F.batched_svd_w(true);
        SymbolVar P = (F - usw[2]) * mu;
        run("arap", P.node(), T, order, xs);
    }
    {  // St. Venant-Kirchhoff stretch term, fea/material.cpp:91-96: P = mu (F F' F - F)
        symbolic::ComputingGraph cg;
        SymbolVar F = symbolic::placeholder(cg);
        SymbolVar FFtF = F.batched_matmul(F.batched_transpose()).batched_matmul(F);
        SymbolVar P = symbolic::linear_combine({{mu, FFtF}, {-mu, F}});
        run("stvk", P.node(), T, order, xs);
    }
    {  // the operators of the Neo-Hookean graphs other than log / pow (fea/material.cpp:55-82, :20-53): F^-T,
       // det F (a batched scalar times a matrix), the sum of squares by multiply + reduce_sum, mul_eye
        symbolic::ComputingGraph cg;
        SymbolVar F = symbolic::placeholder(cg);
        SymbolVar FTinv = symbolic::batched_mat_inv_mul(F, {}, true).batched_transpose();
        SymbolVar J = F.batched_det();
        SymbolVar Ic = (F * F).reduce_sum(-1);
        SymbolVar t2 = symbolic::linear_combine({{mu / -3.0, J * Ic}, {lambda, J * J}, {-lambda, J}}, 0.5) * FTinv;
        SymbolVar P = symbolic::linear_combine({{mu, J * F}, {1.0, t2}, {0.25, Ic.batched_mul_eye(3)}});
        run("nh_parts", P.node(), T, order, xs);
    }
    return 0;
}
