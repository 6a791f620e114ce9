#include "sparse.h"

#include <algorithm>
#include <cstring>
#include <limits>

namespace sanm_hip {

SparseDesc::SparseDesc(int64_t out_size_, int64_t in_size_, const uint64_t* rp, const uint64_t* ix,
                       const double* cf)
        : out_size{out_size_}, in_size{in_size_} {
    sanm_check(out_size > 0 && in_size > 0, "empty sparse map");
    rowptr.assign(rp, rp + out_size + 1);
    sanm_check(rowptr[0] == 0, "rowptr[0] != 0");
    for (int64_t i = 0; i < out_size; ++i)
        sanm_check(rowptr[i + 1] >= rowptr[i], "rowptr not monotone at %ld", (long)i);
    uint64_t nnz = rowptr[out_size];
    idx.assign(ix, ix + nnz);
    coef.assign(cf, cf + nnz);
    for (uint64_t p = 0; p < nnz; ++p)
        sanm_check((int64_t)idx[p] < in_size, "sparse map index %lu out of range (%ld)",
                   (unsigned long)idx[p], (long)in_size);
}

DeviceRows::DeviceRows(Backend* be, const SparseDesc& d, int64_t T, int64_t Tpad, int64_t tet_begin,
                       int64_t tet_end)
        : m_be{be} {
    sanm_check(d.in_size % 9 == 0, "remap_out expects a (T,3,3) input, got %ld elements",
               (long)d.in_size);
    if (tet_end < 0) tet_end = d.in_size / 9;
    sanm_check(tet_end - tet_begin == T, "remap_out: shard size mismatch");
    sanm_check(d.idx.size() < std::numeric_limits<uint32_t>::max(), "remap_out too large");
    std::vector<uint32_t> ptr(d.out_size + 1, 0), idx;
    std::vector<double> coef;
    idx.reserve(d.idx.size());
    coef.reserve(d.idx.size());
    for (int64_t i = 0; i < d.out_size; ++i) {
        for (uint64_t p = d.rowptr[i]; p < d.rowptr[i + 1]; ++p) {
            int64_t e = d.idx[p] / 9, c = d.idx[p] % 9;
            if (e < tet_begin || e >= tet_end) continue;
            idx.push_back(c * Tpad + (e - tet_begin));
            coef.push_back(d.coef[p]);
        }
        ptr[i + 1] = idx.size();
    }
    m_ptr = be->alloc(ptr.size() * 4);
    m_idx = be->alloc(std::max<size_t>(idx.size(), 1) * 4);
    m_coef = be->alloc(std::max<size_t>(idx.size(), 1) * 8);
    be->h2d(m_ptr, ptr.data(), ptr.size() * 4);
    be->h2d(m_idx, idx.data(), idx.size() * 4);
    be->h2d(m_coef, coef.data(), idx.size() * 8);
    m_dev = {static_cast<uint32_t*>(m_ptr), static_cast<uint32_t*>(m_idx),
             static_cast<double*>(m_coef), d.out_size};
}

DeviceRows::~DeviceRows() {
    m_be->free(m_ptr);
    m_be->free(m_idx);
    m_be->free(m_coef);
}

template <class T>
T* JacobianPattern::upload(const std::vector<T>& v) {
    void* p = m_be->alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (!v.empty()) m_be->h2d(p, v.data(), v.size() * sizeof(T));
    m_bufs.push_back(p);
    return static_cast<T*>(p);
}

JacobianPattern::JacobianPattern(Backend* be, const SparseDesc& ro, const SparseDesc& ri, int64_t n,
                                 int64_t T, int64_t Tpad, int odim, int64_t tet_begin, int64_t tet_end)
        : m_be{be} {
    if (tet_end < 0) tet_end = T;
    const int idim = 9;
    sanm_check(ro.out_size == n, "remap_out must produce %ld unknowns, got %ld", (long)n,
               (long)ro.out_size);
    sanm_check(ro.in_size == T * odim && ri.out_size == T * idim, "remap shapes mismatch");
    sanm_check(ri.in_size == n || ri.in_size == n + 1, "remap_in must take n or n+1 inputs");
    m_has_t = ri.in_size == n + 1;

    struct Contrib {
        uint32_t col, jidx;
        double coef;
        bool mine;  // the tet belongs to this rank's shard
    };
    std::vector<Contrib> row;
    std::vector<uint32_t> rowptr(n + 1, 0), col;
    std::vector<uint32_t> aptr{0}, ajidx;
    std::vector<double> acoef;
    std::vector<uint32_t> tptr(n + 1, 0), tjidx;
    std::vector<double> tcoef;

    for (int64_t i = 0; i < n; ++i) {
        row.clear();
        for (uint64_t p = ro.rowptr[i]; p < ro.rowptr[i + 1]; ++p) {
            uint64_t b = ro.idx[p] / odim, o = ro.idx[p] % odim;
            double c_out = ro.coef[p];
            for (int m = 0; m < idim; ++m) {
                uint64_t irow = b * idim + m;
                const bool mine = (int64_t)b >= tet_begin && (int64_t)b < tet_end;
                uint64_t jidx = mine ? ((uint64_t)o * idim + m) * Tpad + (b - tet_begin) : 0;
                sanm_check(jidx < std::numeric_limits<uint32_t>::max(), "mesh too large for u32 jidx");
                for (uint64_t q = ri.rowptr[irow]; q < ri.rowptr[irow + 1]; ++q) {
                    row.push_back({(uint32_t)ri.idx[q], (uint32_t)jidx, c_out * ri.coef[q], mine});
                }
            }
        }
        std::stable_sort(row.begin(), row.end(),
                         [](const Contrib& a, const Contrib& b) { return a.col < b.col; });
        size_t k = 0;
        bool any = false;
        while (k < row.size()) {
            uint32_t c = row[k].col;
            if ((int64_t)c == n) {  // the t column -> grad_t
                for (; k < row.size() && row[k].col == c; ++k) {
                    if (!row[k].mine) continue;
                    tjidx.push_back(row[k].jidx);
                    tcoef.push_back(row[k].coef);
                }
                continue;
            }
            col.push_back(c);
            any = true;
            for (; k < row.size() && row[k].col == c; ++k) {
                if (!row[k].mine) continue;
                ajidx.push_back(row[k].jidx);
                acoef.push_back(row[k].coef);
            }
            sanm_check(ajidx.size() < std::numeric_limits<uint32_t>::max(), "assembly list too large");
            aptr.push_back(ajidx.size());
        }
        sanm_check(any, "empty row %ld", (long)i);  // sparse_solver.cpp:251-252
        rowptr[i + 1] = col.size();
        tptr[i + 1] = tjidx.size();
    }
    m_nr_contrib = ajidx.size();
    m_h_rowptr = rowptr;
    m_h_col = col;

    m_csr.n = n;
    m_csr.nnz = col.size();
    m_csr.rowptr = upload(rowptr);
    m_csr.col = upload(col);
    std::vector<double> zeros(col.size(), 0.0);
    m_csr.val = upload(zeros);
    m_asm = {upload(aptr), upload(ajidx), upload(acoef), (int64_t)col.size()};
    if (m_has_t) m_asm_t = {upload(tptr), upload(tjidx), upload(tcoef), n};
}

JacobianPattern::~JacobianPattern() {
    for (void* p : m_bufs) m_be->free(p);
}

}  // namespace sanm_hip
