// Host-side counterpart of the parts of the reference's fea/ layer that
// define the hot-path inputs: remap tables, constitutive graphs, loads.
//   remap_in   MeshShapeMatTrans      fea/mesh_template.h:20-111
//   remap_out  MeshForceOutputTrans   fea/mesh_template.h:132-161
//   normals    TetrahedralMesh::vertex_norms  fea/tetrahedral_mesh.cpp:31-69
//   graphs     pk1 / cauchy_stress    fea/material.cpp:20-99
//   models     make_forward / make_inverse  fea/mesh_template.h:174-219
#pragma once
#include <cstdint>
#include <vector>

#include "graph.h"
#include "sparse.h"

namespace sanm_hip {

enum EnergyModel : int {  // fea/material.h:47-52
    ENERGY_NEOHOOKEAN_I = 0,
    ENERGY_NEOHOOKEAN_C = 1,
    ENERGY_ARAP = 2,
    ENERGY_STVK_STRETCH = 3,
};

struct Material {  // fea/material.cpp:10-18
    double young, poisson, bulk, shear, lame_first, density;
    static Material from_young_poisson(double E, double nu, double density = 0);
};

int pk1(Graph& g, EnergyModel e, const Material& m, int F);
int cauchy_stress(Graph& g, EnergyModel e, const Material& m, int F);

struct ElasticForceModel {
    Graph graph;
    int y = -1;  // output var (P or sigma)
    int F = -1;
    SparseDesc lt_inp, lt_out;
    std::vector<double> x0;    // n
    std::vector<double> bias;  // (T,3,3)
    std::vector<int64_t> vtx2uidx;                // (nv,3), -1 if fixed
    std::vector<std::pair<int32_t, int32_t>> vertex_loc;  // unknown -> (vtx, coord)
    int64_t n = 0, T = 0, nv = 0;
    bool has_delta = false;
};

//! per-tet normals (T,4,3), volumes (T) and shape matrices (T,3,3)
void tet_geometry(int64_t nv, const double* V, int64_t T, const int32_t* tets,
                  std::vector<double>& norms, std::vector<double>& vol,
                  std::vector<double>& shape_mat);

//! forward model: unknown = deformed positions, y = PK1 stress
//! (init_vtx_coord / vtx_delta may be null)
void make_forward(ElasticForceModel& m, int64_t nv, const double* V, int64_t T, const int32_t* tets,
                  const uint8_t* fixed_mask, EnergyModel e, const Material& mat,
                  const double* init_vtx_coord, const double* vtx_delta);

//! inverse model: unknown = rest positions, y = Cauchy stress on deformed normals
void make_inverse(ElasticForceModel& m, int64_t nv, const double* V, int64_t T, const int32_t* tets,
                  const uint8_t* fixed_mask, EnergyModel e, const Material& mat);

//! nodal gravity load vol*density*g/4 (fea/main.cpp:1025-1036), (nv,3)
void gravity_load(int64_t nv, const double* V, int64_t T, const int32_t* tets, double density,
                  const double g[3], std::vector<double>& f_load);

//! fix surface vertices whose projection is in the lowest slab
//! (setup_boundary_by_config, fea/main.cpp:921-982; filter optional)
void boundary_by_threshold(int64_t nv, const double* V, const uint8_t* is_surface,
                           const double proj_dir[3], double thresh, const double* filter_dir,
                           double filter_min, double filter_max, std::vector<uint8_t>& fixed_mask);

}  // namespace sanm_hip
