// A consumer of include/sanm_hip.h written in C++, compiled by g++ and linked against libsanm_hip.so -- what a
// maintainer of the reference does at fea/main.cpp:393-420 (construct an ANMEqnSolver over the material's graph and
// the mesh's remaps, run_anm, write the vertices), through the C ABI alone: no Python, no ctypes mirror of the
// records.  tests/test_abi_cpp.py builds and runs it on the GPU and compares what it prints with the oracle's golden
// continuations (tests/golden/anm_*.json).
//
//   driver nx ny nz spacing young poisson density gx gy gz thresh px py pz order use_pade
//
// The graph is built HERE with sanm_graph_* -- F = (placeholder + fixed-vertex constant) Dm^-1 (fea/mesh_template.h:
// 191-219) and the compressible Neo-Hookean first Piola-Kirchhoff stress P = mu F - mu F^-T + lambda log(det F) F^-T
// (fea/material.cpp:72-82) --, not taken from sanm_fea_model_graph; the remaps, the rest state and the load come from
// the fea entry points (mesh_template.h:20-161, fea/main.cpp:921-1036).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sanm_hip.h"
#include "sanm_hip_layout.h"  // static_assert(sizeof / offsetof) of the records against the values api.py's mirrors have

#define CHECK(call)                                                                        \
    do {                                                                                   \
        const int rc_ = (call);                                                            \
        if (rc_ != SANM_HIP_OK) {                                                          \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, sanm_hip_last_error()); \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

// TetrahedralMesh::make_cuboid, fea/tetrahedral_mesh.cpp:93-204: nx x ny x nz vertices, 5 tets per cell
static void make_cuboid(int nx, int ny, int nz, double size, std::vector<double>& V, std::vector<int32_t>& T,
                        std::vector<uint8_t>& surf) {
    auto gid = [&](int x, int y, int z) { return (x * ny + y) * nz + z; };
    V.resize((size_t)nx * ny * nz * 3);
    surf.assign((size_t)nx * ny * nz, 0);
    for (int i = 0; i < nx; ++i)
        for (int j = 0; j < ny; ++j)
            for (int k = 0; k < nz; ++k) {
                const int v = gid(i, j, k);
                V[v * 3 + 0] = i * size;
                V[v * 3 + 1] = j * size;
                V[v * 3 + 2] = k * size;
                surf[v] = i == 0 || i == nx - 1 || j == 0 || j == ny - 1 || k == 0 || k == nz - 1;
            }
    static const int pat[5][4] = {{0, 2, 1, 5}, {0, 4, 7, 5}, {0, 2, 5, 7}, {2, 6, 5, 7}, {0, 7, 3, 2}};
    for (int i = 0; i + 1 < nx; ++i)
        for (int j = 0; j + 1 < ny; ++j)
            for (int k = 0; k + 1 < nz; ++k) {
                const int h[8] = {gid(i, j, k),         gid(i + 1, j, k),         gid(i + 1, j + 1, k),
                                  gid(i, j + 1, k),     gid(i, j, k + 1),         gid(i + 1, j, k + 1),
                                  gid(i + 1, j + 1, k + 1), gid(i, j + 1, k + 1)};
                for (const auto& p : pat)
                    for (int q = 0; q < 4; ++q) T.push_back(h[p[q]]);
            }
}

int main(int argc, char** argv) {
    if (argc != 17) {
        std::fprintf(stderr, "usage: %s nx ny nz spacing young poisson density gx gy gz thresh px py pz order use_pade\n", argv[0]);
        return 2;
    }
    const int nx = std::atoi(argv[1]), ny = std::atoi(argv[2]), nz = std::atoi(argv[3]);
    const double spacing = std::atof(argv[4]), young = std::atof(argv[5]), poisson = std::atof(argv[6]),
                 density = std::atof(argv[7]);
    const double g[3] = {std::atof(argv[8]), std::atof(argv[9]), std::atof(argv[10])};
    const double thresh = std::atof(argv[11]);
    const double proj[3] = {std::atof(argv[12]), std::atof(argv[13]), std::atof(argv[14])};
    const int order = std::atoi(argv[15]), use_pade = std::atoi(argv[16]);

    if (sanm_hip_abi_version() != SANM_HIP_ABI_VERSION) {
        std::fprintf(stderr, "library ABI %d, header ABI %d\n", sanm_hip_abi_version(), SANM_HIP_ABI_VERSION);
        return 1;
    }
    CHECK(sanm_hip_init(0));
    std::vector<double> V;
    std::vector<int32_t> T;
    std::vector<uint8_t> surf;
    make_cuboid(nx, ny, nz, spacing, V, T, surf);
    const int64_t nv = (int64_t)V.size() / 3, nt = (int64_t)T.size() / 4;

    // gravity(), fea/main.cpp:984-1046: fixed set by projection threshold, nodal load
    std::vector<uint8_t> fixed((size_t)nv * 3);
    CHECK(sanm_fea_boundary_by_threshold(nv, V.data(), surf.data(), proj, thresh, nullptr, 0, 0, fixed.data()));
    std::vector<double> f_load((size_t)nv * 3);
    CHECK(sanm_fea_gravity_load(nv, V.data(), nt, T.data(), density, g, f_load.data()));

    // the mesh side of the model: remap_inp (x -> F per tet), remap_out (P per tet -> nodal forces), the rest state
    sanm_fea_model* model = nullptr;
    CHECK(sanm_fea_model_create(nv, V.data(), nt, T.data(), fixed.data(), SANM_ENERGY_NEOHOOKEAN_C, young, poisson, 0,
                                nullptr, nullptr, &model));
    int64_t n = 0;
    CHECK(sanm_fea_model_nr_unknown(model, &n));
    std::vector<double> x0((size_t)n), y((size_t)n);
    CHECK(sanm_fea_model_x0(model, x0.data()));
    CHECK(sanm_fea_model_copy_vtx_values(model, f_load.data(), y.data()));

    // the material side, built by this program: pk1() of fea/material.cpp:72-82 with the operator API
    const double mu = young / (2 * (1 + poisson)), lambda = young * poisson / ((1 + poisson) * (1 - 2 * poisson));
    // DeformableBody::make_forward, fea/mesh_template.h:191-219: the placeholder receives the free vertices' part of the
    // deformed shape matrix Ds = [x1 - x0, x2 - x0, x3 - x0] through remap_inp, the fixed vertices' part is a per-tet
    // constant, and F = Ds Dm^-1 with Dm the rest shape matrix
    std::vector<double> bias((size_t)nt * 9), dminv((size_t)nt * 9);
    for (int64_t e = 0; e < nt; ++e) {
        const int32_t* t = &T[e * 4];
        double dm[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                dm[r * 3 + c] = V[t[c + 1] * 3 + r] - V[t[0] * 3 + r];
                bias[e * 9 + r * 3 + c] = (fixed[t[c + 1] * 3 + r] ? V[t[c + 1] * 3 + r] : 0.0) -
                                          (fixed[t[0] * 3 + r] ? V[t[0] * 3 + r] : 0.0);
            }
        const double det = dm[0] * (dm[4] * dm[8] - dm[5] * dm[7]) - dm[1] * (dm[3] * dm[8] - dm[5] * dm[6]) +
                           dm[2] * (dm[3] * dm[7] - dm[4] * dm[6]);
        double* o = &dminv[e * 9];
        o[0] = (dm[4] * dm[8] - dm[5] * dm[7]) / det, o[1] = (dm[2] * dm[7] - dm[1] * dm[8]) / det, o[2] = (dm[1] * dm[5] - dm[2] * dm[4]) / det;
        o[3] = (dm[5] * dm[6] - dm[3] * dm[8]) / det, o[4] = (dm[0] * dm[8] - dm[2] * dm[6]) / det, o[5] = (dm[2] * dm[3] - dm[0] * dm[5]) / det;
        o[6] = (dm[3] * dm[7] - dm[4] * dm[6]) / det, o[7] = (dm[1] * dm[6] - dm[0] * dm[7]) / det, o[8] = (dm[0] * dm[4] - dm[1] * dm[3]) / det;
    }
    sanm_graph* gr = nullptr;
    CHECK(sanm_graph_create(&gr));
    int X, Cb, Ds, Cd, F, Finv, FTinv, J, logJ, lF, P;
    CHECK(sanm_graph_placeholder(gr, &X));
    CHECK(sanm_graph_constant(gr, bias.data(), nt, 9, &Cb));
    {
        const double one[2] = {1.0, 1.0};
        const int v[2] = {X, Cb};
        CHECK(sanm_graph_linear_combine(gr, 2, one, v, 0.0, &Ds));
    }
    CHECK(sanm_graph_constant(gr, dminv.data(), nt, 9, &Cd));
    CHECK(sanm_graph_batched_matmul(gr, Ds, Cd, &F));
    CHECK(sanm_graph_batched_mat_inv_mul(gr, F, -1, 1, &Finv));
    CHECK(sanm_graph_batched_transpose(gr, Finv, &FTinv));
    CHECK(sanm_graph_batched_det(gr, F, &J));
    CHECK(sanm_graph_log(gr, J, &logJ));
    CHECK(sanm_graph_multiply(gr, logJ, FTinv, &lF));
    {
        const double c[3] = {mu, -mu, lambda};
        const int v[3] = {F, FTinv, lF};
        CHECK(sanm_graph_linear_combine(gr, 3, c, v, 0.0, &P));
    }

    sanm_hyper_param hp;
    sanm_hyper_param_default(&hp, 1);
    hp.order = order;
    hp.use_pade = use_pade;
    hp.sanity_check = 1;
    hp.converge_rms = 1e-10;        // run_and_save, fea/main.cpp:382-385
    hp.solution_check_tol = 1e-3;
    hp.solver_rtol = 1e-15;
    sanm_anm_solver* s = nullptr;
    CHECK(sanm_anm_eqn_solver_create(gr, P, sanm_fea_model_remap_inp(model), sanm_fea_model_remap_out(model), x0.data(),
                                     y.data(), n, &hp, &s));
    // run_anm, fea/main.cpp:172-190
    int conv = 0;
    CHECK(sanm_anm_converged(s, &conv));
    for (int it = 0; !conv && it < 1000; ++it) {
        CHECK(sanm_anm_next_iter(s));
        CHECK(sanm_anm_converged(s, &conv));
    }
    int64_t iter = 0;
    double rms = 0;
    CHECK(sanm_anm_nr_iter(s, &iter));
    CHECK(sanm_anm_residual_rms(s, &rms));
    std::vector<double> x((size_t)n), Vout(V);
    CHECK(sanm_anm_get_x(s, x.data()));
    CHECK(sanm_fea_model_scatter(model, x.data(), Vout.data()));
    sanm_anm_stats st;
    std::memset(&st, 0x5a, sizeof st);
    CHECK(sanm_anm_get_stats(s, &st));
    std::printf("backend %s\nconverged %d\niter %lld\nrms %.17g\nnr_unknown %lld\nnr_tet %lld\njacobian_nnz %lld\nnr_linear_solve %lld\n",
                sanm_hip_backend_name(), conv, (long long)iter, rms, (long long)st.nr_unknown, (long long)st.nr_tet,
                (long long)st.jacobian_nnz, (long long)st.nr_linear_solve);
    std::printf("vertices %lld\n", (long long)nv);
    for (int64_t i = 0; i < nv; ++i) std::printf("%.17g %.17g %.17g\n", Vout[i * 3], Vout[i * 3 + 1], Vout[i * 3 + 2]);
    sanm_anm_solver_destroy(s);
    sanm_graph_destroy(gr);
    sanm_fea_model_destroy(model);
    return conv ? 0 : 3;
}
