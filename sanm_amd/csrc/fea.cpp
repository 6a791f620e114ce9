#include "fea.h"

#include <cmath>
#include <limits>

#include "tet_ops.h"

namespace sanm_hip {

Material Material::from_young_poisson(double E, double nu, double density) {
    Material r;
    r.young = E;
    r.poisson = nu;
    r.bulk = E / (3 * (1 - nu * 2));
    r.shear = E / (2 * (1 + nu));
    r.lame_first = E * nu / ((1 + nu) * (1 - nu * 2));
    r.density = density;
    return r;
}

namespace {
int lincomb(Graph& g, std::initializer_list<std::pair<double, int>> terms, double bias = 0) {
    std::vector<double> c;
    std::vector<int> v;
    for (auto& t : terms) {
        c.push_back(t.first);
        v.push_back(t.second);
    }
    return g.linear_combine(c.size(), c.data(), v.data(), bias);
}
}  // namespace

int pk1(Graph& g, EnergyModel e, const Material& m, int F) {
    // fea/material.cpp:55-99
    switch (e) {
        case ENERGY_NEOHOOKEAN_I: {
            double k = m.bulk, mu = m.shear;
            int FTinv = g.batched_transpose(g.batched_mat_inv_mul(F, -1, true));
            int J = g.batched_det(F);
            int Ic = g.reduce_sum(g.pow(F, 2), -1);
            int J23 = g.pow(J, -2. / 3.);
            int t2 = g.multiply(
                    lincomb(g, {{mu / -3.0, g.multiply(J23, Ic)}, {k, g.multiply(J, J)}, {-k, J}}, 0),
                    FTinv);
            return lincomb(g, {{mu, g.multiply(J23, F)}, {1.0, t2}});
        }
        case ENERGY_NEOHOOKEAN_C: {
            double mu = m.shear, lambda = m.lame_first;
            int FTinv = g.batched_transpose(g.batched_mat_inv_mul(F, -1, true));
            int J = g.batched_det(F);
            return lincomb(g, {{mu, F}, {-mu, FTinv}, {lambda, g.multiply(g.log(J), FTinv)}});
        }
        case ENERGY_ARAP: {
            int usw[3];
            g.batched_svd_w(F, true, usw);
            return lincomb(g, {{m.shear, lincomb(g, {{1.0, F}, {-1.0, usw[2]}})}}, 0);
        }
        case ENERGY_STVK_STRETCH: {
            double mu = m.shear;
            int FFtF = g.batched_matmul(g.batched_matmul(F, g.batched_transpose(F)), F);
            return lincomb(g, {{mu, FFtF}, {-mu, F}});
        }
    }
    sanm_throw(SANM_ERR_ASSERT, "pk1 unimplemented for energy model %d", (int)e);
}

int cauchy_stress(Graph& g, EnergyModel e, const Material& m, int F) {
    // fea/material.cpp:20-53
    switch (e) {
        case ENERGY_NEOHOOKEAN_I: {
            double k = m.bulk, mu = m.shear;
            int b = g.batched_matmul(F, g.batched_transpose(F));
            int J = g.batched_det(F);
            int Ic = g.reduce_sum(g.pow(F, 2), -1);
            int J53 = g.pow(J, -5. / 3.);
            int t2 = g.batched_mul_eye(lincomb(g, {{mu / -3.0, g.multiply(J53, Ic)}, {k, J}}, -k), 3);
            return lincomb(g, {{mu, g.multiply(J53, b)}, {1.0, t2}});
        }
        case ENERGY_NEOHOOKEAN_C: {
            double lambda = m.lame_first, mu = m.shear;
            int b = g.batched_matmul(F, g.batched_transpose(F));
            int Jinv = g.pow(g.batched_det(F), -1);
            int xI = lincomb(g, {{mu, Jinv}, {lambda, g.multiply(Jinv, g.log(Jinv))}});
            return lincomb(g, {{mu, g.multiply(Jinv, b)}, {-1.0, g.batched_mul_eye(xI, 3)}});
        }
        default:
            sanm_throw(SANM_ERR_ASSERT, "cauchy_stress unimplemented for energy model %d", (int)e);
    }
}

void tet_geometry(int64_t nv, const double* V, int64_t T, const int32_t* tets,
                  std::vector<double>& norms, std::vector<double>& vol,
                  std::vector<double>& shape_mat) {
    // fea/tetrahedral_mesh.cpp:31-69
    norms.assign(T * 12, 0.0);
    vol.assign(T, 0.0);
    shape_mat.assign(T * 9, 0.0);
    for (int64_t i = 0; i < T; ++i) {
        const int32_t* t = tets + i * 4;
        for (int k = 0; k < 4; ++k) sanm_check(t[k] >= 0 && t[k] < nv, "tet %ld: bad vertex", (long)i);
        const double *x0 = V + 3 * t[0], *x1 = V + 3 * t[1], *x2 = V + 3 * t[2], *x3 = V + 3 * t[3];
        double v1[3], v2[3], v3[3];
        for (int r = 0; r < 3; ++r) {
            v1[r] = x1[r] - x0[r];
            v2[r] = x2[r] - x0[r];
            v3[r] = x3[r] - x0[r];
            shape_mat[i * 9 + r * 3 + 0] = v1[r];
            shape_mat[i * 9 + r * 3 + 1] = v2[r];
            shape_mat[i * 9 + r * 3 + 2] = v3[r];
        }
        double t1[3], t2[3], t3[3];
        cross3(v2, v3, t1);
        cross3(v3, v1, t2);
        cross3(v1, v2, t3);
        double det = v1[0] * t1[0] + v1[1] * t1[1] + v1[2] * t1[2];
        vol[i] = std::fabs(det) / 6;
        double sg = det > 0 ? -1.0 : 1.0;
        for (int r = 0; r < 3; ++r) {
            double a = sg * t1[r], b = sg * t2[r], c = sg * t3[r];
            norms[i * 12 + 0 + r] = -(a + b + c) * (1.0 / 6);
            norms[i * 12 + 3 + r] = a * (1.0 / 6);
            norms[i * 12 + 6 + r] = b * (1.0 / 6);
            norms[i * 12 + 9 + r] = c * (1.0 / 6);
        }
    }
}

namespace {
// MeshShapeMatTrans, fea/mesh_template.h:20-111
void build_remaps(ElasticForceModel& m, int64_t nv, const double* V, int64_t T,
                  const int32_t* tets, const uint8_t* fixed, const double* init_vtx,
                  const double* vtx_delta) {
    const double* V0 = init_vtx ? init_vtx : V;
    m.nv = nv;
    m.T = T;
    m.has_delta = vtx_delta != nullptr;
    m.vtx2uidx.assign(nv * 3, -1);
    m.x0.clear();
    m.vertex_loc.clear();
    for (int64_t i = 0; i < nv; ++i)
        for (int j = 0; j < 3; ++j)
            if (!fixed[i * 3 + j]) {
                m.vtx2uidx[i * 3 + j] = m.x0.size();
                m.x0.push_back(V0[i * 3 + j]);
                m.vertex_loc.emplace_back(i, j);
            }
    m.n = m.x0.size();
    sanm_check(m.n > 0, "all coordinates are fixed");
    m.bias.assign(T * 9, 0.0);

    SparseDesc& in = m.lt_inp;
    in.out_size = T * 9;
    in.in_size = m.n + (m.has_delta ? 1 : 0);
    in.rowptr.assign(T * 9 + 1, 0);
    in.idx.clear();
    in.coef.clear();
    for (int64_t e = 0; e < T; ++e) {
        int32_t v0 = tets[e * 4];
        // outputs must be emitted in flattened order e*9 + r*3 + (dm-1)
        for (int r = 0; r < 3; ++r)
            for (int dm = 1; dm <= 3; ++dm) {
                int32_t vi = tets[e * 4 + dm];
                int64_t oidx = e * 9 + r * 3 + (dm - 1);
                if (int64_t u = m.vtx2uidx[v0 * 3 + r]; u < 0) {
                    m.bias[oidx] -= V0[v0 * 3 + r];
                } else {
                    in.idx.push_back(u);
                    in.coef.push_back(-1.0);
                }
                if (int64_t u = m.vtx2uidx[vi * 3 + r]; u < 0) {
                    m.bias[oidx] += V0[vi * 3 + r];
                } else {
                    in.idx.push_back(u);
                    in.coef.push_back(1.0);
                }
                if (vtx_delta) {
                    double d = vtx_delta[vi * 3 + r] - vtx_delta[v0 * 3 + r];
                    if (d != 0) {
                        in.idx.push_back(m.n);
                        in.coef.push_back(d);
                    }
                }
                in.rowptr[oidx + 1] = in.idx.size();
            }
    }

    // MeshForceOutputTrans, fea/mesh_template.h:132-161; adjacency in tet
    // order (MeshVertexReverseList, fea/mesh.cpp:27-52)
    std::vector<double> norms, vol, sm;
    tet_geometry(nv, V, T, tets, norms, vol, sm);
    std::vector<uint32_t> vptr(nv + 1, 0);
    for (int64_t e = 0; e < T; ++e)
        for (int k = 0; k < 4; ++k) vptr[tets[e * 4 + k] + 1]++;
    for (int64_t i = 0; i < nv; ++i) {
        sanm_check(vptr[i + 1] > 0, "dangling vertex %ld", (long)i);
        vptr[i + 1] += vptr[i];
    }
    std::vector<std::pair<int32_t, int32_t>> adj(T * 4);
    {
        std::vector<uint32_t> fill(vptr.begin(), vptr.end() - 1);
        for (int64_t e = 0; e < T; ++e)
            for (int k = 0; k < 4; ++k) adj[fill[tets[e * 4 + k]]++] = {(int32_t)e, k};
    }
    SparseDesc& out = m.lt_out;
    out.out_size = m.n;
    out.in_size = T * 9;
    out.rowptr.assign(m.n + 1, 0);
    out.idx.clear();
    out.coef.clear();
    for (int64_t i = 0; i < m.n; ++i) {
        auto [vtx, coord] = m.vertex_loc[i];
        for (uint32_t p = vptr[vtx]; p < vptr[vtx + 1]; ++p) {
            auto [e, k] = adj[p];
            for (int j = 0; j < 3; ++j) {
                out.coef.push_back(norms[(int64_t)e * 12 + k * 3 + j]);
                out.idx.push_back((int64_t)e * 9 + coord * 3 + j);
            }
        }
        out.rowptr[i + 1] = out.idx.size();
    }
}
}  // namespace

void make_forward(ElasticForceModel& m, int64_t nv, const double* V, int64_t T, const int32_t* tets,
                  const uint8_t* fixed_mask, EnergyModel e, const Material& mat,
                  const double* init_vtx_coord, const double* vtx_delta) {
    // fea/mesh_template.h:191-219
    build_remaps(m, nv, V, T, tets, fixed_mask, init_vtx_coord, vtx_delta);
    std::vector<double> norms, vol, sm;
    tet_geometry(nv, V, T, tets, norms, vol, sm);
    std::vector<double> dminv(T * 9);
    for (int64_t i = 0; i < T; ++i) inv3(&sm[i * 9], &dminv[i * 9]);
    Graph& g = m.graph;
    double one[2] = {1.0, 1.0};
    int vs[2] = {g.placeholder(), g.constant(m.bias.data(), T, 9)};
    int Ds = g.linear_combine(2, one, vs, 0);
    int DmInv = g.constant(dminv.data(), T, 9);
    m.F = g.batched_matmul(Ds, DmInv);
    m.y = pk1(g, e, mat, m.F);
}

void make_inverse(ElasticForceModel& m, int64_t nv, const double* V, int64_t T, const int32_t* tets,
                  const uint8_t* fixed_mask, EnergyModel e, const Material& mat) {
    // fea/mesh_template.h:174-189
    build_remaps(m, nv, V, T, tets, fixed_mask, nullptr, nullptr);
    std::vector<double> norms, vol, sm;
    tet_geometry(nv, V, T, tets, norms, vol, sm);
    Graph& g = m.graph;
    double one[2] = {1.0, 1.0};
    int vs[2] = {g.placeholder(), g.constant(m.bias.data(), T, 9)};
    int Dm = g.linear_combine(2, one, vs, 0);
    int Ds = g.constant(sm.data(), T, 9);
    m.F = g.batched_mat_inv_mul(Dm, Ds, true);
    m.y = cauchy_stress(g, e, mat, m.F);
}

void gravity_load(int64_t nv, const double* V, int64_t T, const int32_t* tets, double density,
                  const double g[3], std::vector<double>& f_load) {
    std::vector<double> norms, vol, sm;
    tet_geometry(nv, V, T, tets, norms, vol, sm);
    f_load.assign(nv * 3, 0.0);
    for (int64_t i = 0; i < T; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 3; ++r) f_load[tets[i * 4 + j] * 3 + r] += vol[i] * density * g[r] / 4;
}

void boundary_by_threshold(int64_t nv, const double* V, const uint8_t* is_surface,
                           const double proj_dir_in[3], double thresh_ratio, const double* filter_dir,
                           double filter_min, double filter_max, std::vector<uint8_t>& fixed_mask) {
    double d[3] = {proj_dir_in[0], proj_dir_in[1], proj_dir_in[2]};
    double nrm = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (double& x : d) x /= nrm;
    double pmin = std::numeric_limits<double>::infinity(), pmax = -pmin;
    auto proj = [&](int64_t i, const double* dir) {
        return V[i * 3] * dir[0] + V[i * 3 + 1] * dir[1] + V[i * 3 + 2] * dir[2];
    };
    for (int64_t i = 0; i < nv; ++i) {
        double p = proj(i, d);
        pmin = std::min(pmin, p);
        pmax = std::max(pmax, p);
    }
    double thresh = pmin + (pmax - pmin) * thresh_ratio;
    double th0 = 0, th1 = 0;
    if (filter_dir) {
        double qmin = std::numeric_limits<double>::infinity(), qmax = -qmin;
        for (int64_t i = 0; i < nv; ++i) {
            double q = proj(i, filter_dir);
            qmin = std::min(qmin, q);
            qmax = std::max(qmax, q);
        }
        th0 = qmin + (qmax - qmin) * filter_min;
        th1 = qmin + (qmax - qmin) * filter_max;
    }
    fixed_mask.assign(nv * 3, 0);
    for (int64_t i = 0; i < nv; ++i) {
        if (proj(i, d) <= thresh && is_surface[i]) {
            if (filter_dir) {
                double q = proj(i, filter_dir);
                if (!(q >= th0 && q <= th1)) continue;
            }
            fixed_mask[i * 3] = fixed_mask[i * 3 + 1] = fixed_mask[i * 3 + 2] = 1;
        }
    }
}

}  // namespace sanm_hip
