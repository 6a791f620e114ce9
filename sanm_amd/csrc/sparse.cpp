#include "sparse.h"
#include "host_parallel.h"

#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <limits>
#include <string>
#include <thread>

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
                       int64_t tet_end, int64_t block)
        : m_be{be} {
    sanm_check(block >= 1 && d.in_size % block == 0, "sparse map over a (T,%ld) tensor: got %ld input elements",
               (long)block, (long)d.in_size);
    if (tet_end < 0) tet_end = d.in_size / block;
    sanm_check(tet_end - tet_begin == T, "remap_out: shard size mismatch");
    sanm_check(d.idx.size() < std::numeric_limits<uint32_t>::max(), "remap_out too large");
    std::vector<uint32_t> ptr(d.out_size + 1, 0), idx;
    std::vector<double> coef;
    const bool whole = tet_begin == 0 && tet_end * block == d.in_size;
    parallel_ranges(d.out_size, 16384, [&](int64_t i0, int64_t i1, int) {
        for (int64_t i = i0; i < i1; ++i) {
            uint32_t cnt = 0;
            if (whole)
                cnt = (uint32_t)(d.rowptr[i + 1] - d.rowptr[i]);
            else
                for (uint64_t p = d.rowptr[i]; p < d.rowptr[i + 1]; ++p) {
                    const int64_t e = d.idx[p] / block;
                    cnt += e >= tet_begin && e < tet_end;
                }
            ptr[i + 1] = cnt;
        }
    });
    for (int64_t i = 0; i < d.out_size; ++i) ptr[i + 1] += ptr[i];
    idx.resize(ptr[d.out_size]);
    coef.resize(ptr[d.out_size]);
    parallel_ranges(d.out_size, 16384, [&](int64_t i0, int64_t i1, int) {
        for (int64_t i = i0; i < i1; ++i) {
            uint32_t w = ptr[i];
            for (uint64_t p = d.rowptr[i]; p < d.rowptr[i + 1]; ++p) {
                int64_t e = d.idx[p] / block, c = d.idx[p] % block;
                if (e < tet_begin || e >= tet_end) continue;
                idx[w] = (uint32_t)((e - tet_begin) * block + c);
                coef[w++] = d.coef[p];
            }
        }
    });
    m_ptr = be->alloc(ptr.size() * 4);
    m_idx = be->alloc(std::max<size_t>(idx.size(), 1) * 4);
    m_coef = be->alloc(std::max<size_t>(idx.size(), 1) * 8);
    be->h2d(m_ptr, ptr.data(), ptr.size() * 4);
    be->h2d(m_idx, idx.data(), idx.size() * 4);
    be->h2d(m_coef, coef.data(), idx.size() * 8);
    m_dev = {static_cast<uint32_t*>(m_ptr), static_cast<uint32_t*>(m_idx),
             static_cast<double*>(m_coef), d.out_size, nullptr, nullptr, nullptr};
    // rows in triples?  (SparseRowsDev: same coefficients, indices shifted by 0 / 3 / 6 inside one tet's block)
    const int64_t nr = d.out_size;
    bool triples = nr > 0 && nr % 3 == 0 && block == 9;
    if (triples) {
        std::vector<char> ok(64, 1);
        parallel_ranges(nr / 3, 16384, [&](int64_t u0, int64_t u1, int t) {
            bool good = true;
            for (int64_t u = u0; good && u < u1; ++u) {
                const uint32_t p0 = ptr[3 * u], len = ptr[3 * u + 1] - p0;
                for (int c = 1; good && c < 3; ++c) {
                    const uint32_t pc = ptr[3 * u + c];
                    good = ptr[3 * u + c + 1] - pc == len;
                    for (uint32_t q = 0; good && q < len; ++q)
                        good = idx[pc + q] == idx[p0 + q] + 3u * c && coef[pc + q] == coef[p0 + q] && idx[p0 + q] % 9 < 3;
                }
            }
            if (!good) ok[t % 64] = 0;
        });
        for (char c : ok) triples = triples && c;
    }
    if (std::getenv("SANM_DEBUG"))
        std::fprintf(stderr, "remap_out: %ld rows, %zu entries, rows in triples: %s\n", (long)nr, idx.size(), triples ? "yes" : "no");
    if (triples) {
        std::vector<uint32_t> bptr(nr / 3 + 1, 0), bidx;
        std::vector<double> bcoef;
        for (int64_t u = 0; u < nr / 3; ++u) {
            bidx.insert(bidx.end(), idx.begin() + ptr[3 * u], idx.begin() + ptr[3 * u + 1]);
            bcoef.insert(bcoef.end(), coef.begin() + ptr[3 * u], coef.begin() + ptr[3 * u + 1]);
            bptr[u + 1] = bidx.size();
        }
        m_bptr = be->alloc(bptr.size() * 4);
        m_bidx = be->alloc(std::max<size_t>(bidx.size(), 1) * 4);
        m_bcoef = be->alloc(std::max<size_t>(bidx.size(), 1) * 8);
        be->h2d(m_bptr, bptr.data(), bptr.size() * 4);
        be->h2d(m_bidx, bidx.data(), bidx.size() * 4);
        be->h2d(m_bcoef, bcoef.data(), bidx.size() * 8);
        m_dev.bptr = static_cast<uint32_t*>(m_bptr);
        m_dev.bidx = static_cast<uint32_t*>(m_bidx);
        m_dev.bcoef = static_cast<double*>(m_bcoef);
    }
}

DeviceRows::~DeviceRows() {
    m_be->free(m_ptr);
    m_be->free(m_idx);
    m_be->free(m_coef);
    if (m_bptr) {
        m_be->free(m_bptr);
        m_be->free(m_bidx);
        m_be->free(m_bcoef);
    }
}

template <class T>
T* JacobianPattern::upload(const std::vector<T>& v) {
    void* p = m_be->alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (!v.empty()) m_be->h2d(p, v.data(), v.size() * sizeof(T));
    m_bufs.push_back(p);
    return static_cast<T*>(p);
}

JacobianPattern::JacobianPattern(Backend* be, const SparseDesc& ro, const SparseDesc& ri, int64_t n,
                                 int64_t T, int64_t Tpad, int odim, int64_t tet_begin, int64_t tet_end, int idim)
        : m_be{be} {
    if (tet_end < 0) tet_end = T;
    sanm_check(ro.out_size == n, "remap_out must produce %ld unknowns, got %ld", (long)n,
               (long)ro.out_size);
    sanm_check(ro.in_size == T * odim && ri.out_size == T * idim, "remap shapes mismatch");
    sanm_check(ri.in_size == n || ri.in_size == n + 1, "remap_in must take n or n+1 inputs");
    m_has_t = ri.in_size == n + 1;

    // The pattern: row i of remap_out . blockdiag(J_e) . remap_in holds the columns of the remap_in rows of the batch
    // items row i of remap_out touches.  Only the pattern is built here; the values are assembled from the two remap
    // tables on the device (backend.h: AssemblyDev).  Rows are independent: several host threads, each with a dense
    // marker over the columns.
    sanm_check(ro.idx.size() < std::numeric_limits<uint32_t>::max() && ri.idx.size() < std::numeric_limits<uint32_t>::max() &&
                       (uint64_t)T * std::max(odim, idim) < std::numeric_limits<uint32_t>::max(),
               "remap tables too large for 32-bit indices");
    sanm_check((uint64_t)(tet_end - tet_begin) * odim * idim < std::numeric_limits<uint32_t>::max(),
               "mesh too large for 32-bit Jacobian block indices");
    struct Part {
        std::vector<uint32_t> row_nnz, col;
        int64_t contrib = 0;
        std::string error;
    };
    const int nthread = (int)std::max<int64_t>(
            1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 16, n / 4096 + 1}));
    std::vector<Part> parts(nthread);
    auto build = [&](int t) {
        Part& P = parts[t];
        const int64_t r0 = n * t / nthread, r1 = n * (t + 1) / nthread;
        std::vector<uint8_t> mark(n + 1, 0);
        std::vector<uint32_t> ucol;
        try {
            for (int64_t i = r0; i < r1; ++i) {
                ucol.clear();
                for (uint64_t p = ro.rowptr[i]; p < ro.rowptr[i + 1]; ++p) {
                    const uint64_t b = ro.idx[p] / odim;
                    const bool mine = (int64_t)b >= tet_begin && (int64_t)b < tet_end;
                    for (int m = 0; m < idim; ++m) {
                        const uint64_t irow = b * idim + m;
                        for (uint64_t q = ri.rowptr[irow]; q < ri.rowptr[irow + 1]; ++q) {
                            const uint32_t c = (uint32_t)ri.idx[q];
                            if (!mark[c]) {
                                mark[c] = 1;
                                ucol.push_back(c);
                            }
                        }
                        if (mine) P.contrib += (int64_t)(ri.rowptr[irow + 1] - ri.rowptr[irow]);
                    }
                }
                std::sort(ucol.begin(), ucol.end());
                uint32_t nnz_row = 0;
                for (uint32_t c : ucol) {
                    mark[c] = 0;
                    if ((int64_t)c == n) continue;  // the t column -> grad_t
                    P.col.push_back(c);
                    ++nnz_row;
                }
                sanm_check(nnz_row > 0, "empty row %ld", (long)i);  // sparse_solver.cpp:251-252
                P.row_nnz.push_back(nnz_row);
            }
        } catch (const SanmError& e) {
            P.error = e.msg;
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nthread; ++t) th.emplace_back(build, t);
        build(0);
        for (auto& x : th) x.join();
    }
    for (const Part& P : parts) sanm_check(P.error.empty(), "%s", P.error.c_str());
    std::vector<uint32_t> rowptr(n + 1, 0), col;
    {
        size_t ncol = 0;
        for (const Part& P : parts) ncol += P.col.size();
        sanm_check(ncol < std::numeric_limits<uint32_t>::max(), "Jacobian pattern too large");
        col.reserve(ncol);
        int64_t i = 0;
        for (const Part& P : parts) {
            for (size_t r = 0; r < P.row_nnz.size(); ++r, ++i) rowptr[i + 1] = rowptr[i] + P.row_nnz[r];
            col.insert(col.end(), P.col.begin(), P.col.end());
            m_nr_contrib += P.contrib;
        }
    }
    m_h_rowptr = rowptr;
    m_h_col = col;

    m_csr.n = n;
    m_csr.nnz = col.size();
    m_csr.rowptr = upload(rowptr);
    m_csr.col = upload(col);
    std::vector<double> zeros(col.size(), 0.0);
    m_csr.val = upload(zeros);
    // the remap tables as the assembly reads them
    auto narrow = [](const std::vector<uint64_t>& v) { return std::vector<uint32_t>(v.begin(), v.end()); };
    m_asm.ro_ptr = upload(narrow(ro.rowptr));
    m_asm.ro_idx = upload(narrow(ro.idx));
    m_asm.ro_coef = upload(ro.coef);
    m_asm.ri_ptr = upload(narrow(ri.rowptr));
    m_asm.ri_idx = upload(narrow(ri.idx));
    m_asm.ri_coef = upload(ri.coef);
    m_asm.rowptr = m_csr.rowptr;
    m_asm.col = m_csr.col;
    m_asm.n = n;
    m_asm.odim = odim;
    m_asm.idim = idim;
    m_asm.tet_begin = tet_begin;
    m_asm.tet_end = tet_end;
    m_asm.has_t = m_has_t ? 1 : 0;
    m_asm.nnz = m_csr.nnz;
    be->prepare_assembly(m_asm, m_bufs);
}

JacobianPattern::~JacobianPattern() {
    for (void* p : m_bufs) m_be->free(p);
}

}  // namespace sanm_hip
