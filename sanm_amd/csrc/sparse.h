// Sparse linear maps and the Jacobian CSR pattern.
//
// SparseDesc is the host form of the reference's SparseLinearDescCompressed
// (libsanm/anm.h:76-85): for every output element a list of (coeff, input
// index).  JacobianPattern is the symbolic product
//     remap_out . blockdiag(J_e) . remap_in
// of libsanm/anm.cpp:362-438 / :520-608, computed ONCE per model: the sparsity
// pattern never changes between ANM steps (only J_e does), so the reference's
// per-step sort+merge (sparse_solver.cpp:250-305) becomes a fixed gather list
// per non-zero that a HIP kernel evaluates every step.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <vector>

#include "backend.h"
#include "graph.h"

namespace sanm_hip {

struct SparseDesc {
    int64_t out_size = 0, in_size = 0;
    std::vector<uint64_t> rowptr;  // out_size+1
    std::vector<uint64_t> idx;
    std::vector<double> coef;
    //! optional (out_size,3) spatial position of every output element; used as an
    //! ordering hint by the direct solver when this map is remap_out
    std::vector<double> out_coords;

    SparseDesc() = default;
    SparseDesc(int64_t out_size, int64_t in_size, const uint64_t* rowptr, const uint64_t* idx,
               const double* coef);
};

//! device copy of the rows of a SparseDesc whose *inputs* are a flattened
//! (T,9) AoS tensor (tet-major, like the output buffer of the program: ProgramDev::out_aos)
class DeviceRows {
public:
    //! only the entries whose input tet lies in [tet_begin, tet_end) are kept (all of
    //! them by default); T / Tpad describe the local SoA tensor
    //! block: elements per batch item of the input tensor (9 for the (T,3,3) output of a tet program)
    //! tet_inv (optional): batch item b of the map's inputs is item tet_inv[b] of the tensor (the driver's renumbering)
    DeviceRows(Backend* be, const SparseDesc& d, int64_t T, int64_t Tpad, int64_t tet_begin = 0,
               int64_t tet_end = -1, int64_t block = 9, const int64_t* tet_inv = nullptr);
    //! the host half of the constructor above (no backend: any thread) and the device half (the backend's owner thread)
    struct Packed {
        std::vector<uint32_t> ptr;
        std::unique_ptr<uint32_t[]> idx;
        std::unique_ptr<double[]> coef;
        int64_t nr = 0;
        bool triples = false;
    };
    static Packed pack_host(const SparseDesc& d, int64_t T, int64_t Tpad, int64_t tet_begin = 0, int64_t tet_end = -1,
                            int64_t block = 9, const int64_t* tet_inv = nullptr);
    DeviceRows(Backend* be, Packed&& rows);
    ~DeviceRows();
    SparseRowsDev dev() const { return m_dev; }

private:
    Backend* m_be;
    SparseRowsDev m_dev{};
    void *m_ptr = nullptr, *m_idx = nullptr, *m_coef = nullptr;
    void *m_bptr = nullptr, *m_bidx = nullptr, *m_bcoef = nullptr;  // rows in triples (SparseRowsDev)
};

class JacobianPattern {
public:
    //! n = number of unknowns; remap_in may have n or n+1 columns (column n is
    //! the continuation parameter t of ANMImplicitSolver, anm.cpp:575-579)
    //! T: global number of tets; contributions of tets outside [tet_begin, tet_end)
    //! are left to the other ranks (the CSR pattern itself is always the global one)
    //! odim / idim: elements per batch item of the graph's output / placeholder (the blocks of the Jacobian are
    //! batch-major [T][odim][idim])
    //! tet_order / tet_inv (optional, both or none): the device's batch item e is the maps' item tet_order[e]; the
    //! shard [tet_begin, tet_end) and the Jacobian blocks are in the device's numbering
    //! on_blocks (optional): called -- from the constructor, before the rows of the unknowns are written out -- with the
    //! pattern of the 3 x 3 BLOCKS when the maps have that structure (the three rows of a vertex touch the same batch
    //! items, every item's columns come in whole triples, no t column): block row v lists the blocks of its columns,
    //! ascending.  The rows of the pattern are then those lists written out.  Not called otherwise.
    using BlockRows = std::shared_ptr<const std::vector<uint32_t>>;
    JacobianPattern(Backend* be, const SparseDesc& remap_out, const SparseDesc& remap_in, int64_t n,
                    int64_t T, int64_t Tpad, int odim, int64_t tet_begin = 0, int64_t tet_end = -1, int idim = 9,
                    const int64_t* tet_order = nullptr, const int64_t* tet_inv = nullptr, bool defer_device = false,
                    const std::function<void(BlockRows qptr, BlockRows qcol)>& on_blocks = {});
    ~JacobianPattern();
    //! defer_device: the constructor builds the host pattern only (h_rowptr / h_col are valid -- what the analysis of a
    //! direct solver needs --, nothing else is); finish_device, with the constructor's maps, makes the device side
    void finish_device(const SparseDesc& remap_out, const SparseDesc& remap_in);
    //! a deferred pattern of ALL batch items (no shard: the host pattern does not depend on the numbering) built before
    //! the renumbering was known: hand it in before finish_device
    void set_tet_order(const int64_t* tet_order, const int64_t* tet_inv) {
        sanm_check(!tet_order == !tet_inv, "a renumbering of the batch items comes with its inverse");
        sanm_check(m_tet_begin == 0 && m_tet_end == m_T, "set_tet_order: the pattern of a shard was built in a numbering");
        m_tet_order = tet_order, m_tet_inv = tet_inv;
    }

    CsrDev csr() const { return m_csr; }
    AssemblyDev assembly() const { return m_asm; }
    bool has_t() const { return m_has_t; }
    int64_t n() const { return m_csr.n; }
    int64_t nnz() const { return m_csr.nnz; }
    int64_t nr_contrib() const { return m_nr_contrib; }
    const std::vector<uint32_t>& h_rowptr() const { return m_h_rowptr; }
    const std::vector<uint32_t>& h_col() const { return m_h_col; }

private:
    Backend* m_be;
    CsrDev m_csr{};
    AssemblyDev m_asm{};
    bool m_has_t = false;
    int64_t m_nr_contrib = 0;
    std::vector<uint32_t> m_h_rowptr, m_h_col;
    std::vector<void*> m_bufs;
    // what finish_device needs of the constructor's arguments
    int64_t m_n = 0, m_T = 0, m_tet_begin = 0, m_tet_end = 0;
    int m_odim = 9, m_idim = 9;
    const int64_t *m_tet_order = nullptr, *m_tet_inv = nullptr;
    template <class T>
    T* upload(const std::vector<T>& v);
    template <class T>
    T* upload(const T* v, size_t count);
};

}  // namespace sanm_hip
