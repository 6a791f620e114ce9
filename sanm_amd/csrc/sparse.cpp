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
                       int64_t tet_end, int64_t block, const int64_t* tet_inv)
        : DeviceRows(be, pack_host(d, T, Tpad, tet_begin, tet_end, block, tet_inv)) {}

DeviceRows::Packed DeviceRows::pack_host(const SparseDesc& d, int64_t T, int64_t Tpad, int64_t tet_begin,
                                         int64_t tet_end, int64_t block, const int64_t* tet_inv) {
    sanm_check(block >= 1 && d.in_size % block == 0, "sparse map over a (T,%ld) tensor: got %ld input elements",
               (long)block, (long)d.in_size);
    if (tet_end < 0) tet_end = d.in_size / block;
    sanm_check(tet_end - tet_begin == T, "remap_out: shard size mismatch");
    sanm_check(d.idx.size() < std::numeric_limits<uint32_t>::max(), "remap_out too large");
    (void)Tpad;
    const bool whole = tet_begin == 0 && tet_end * block == d.in_size;
    // entry p of the source as this rank sees it: (kept?, index into the local (T, block) tensor)
    auto local = [&](uint64_t p, uint32_t& li) {
        int64_t e = d.idx[p] / block;
        const int64_t c = d.idx[p] % block;
        if (tet_inv) e = tet_inv[e];
        li = (uint32_t)((e - tet_begin) * block + c);
        return e >= tet_begin && e < tet_end;
    };
    const int64_t nr = d.out_size;
    SetupLaps laps("remap_out rows");
    std::vector<uint32_t> ptr(nr + 1, 0);
    parallel_ranges(nr, 4096, [&](int64_t i0, int64_t i1, int) {
        for (int64_t i = i0; i < i1; ++i) {
            uint32_t cnt = 0, li;
            if (whole)
                cnt = (uint32_t)(d.rowptr[i + 1] - d.rowptr[i]);
            else
                for (uint64_t p = d.rowptr[i]; p < d.rowptr[i + 1]; ++p) cnt += local(p, li);
            ptr[i + 1] = cnt;
        }
    });
    for (int64_t i = 0; i < nr; ++i) ptr[i + 1] += ptr[i];
    // rows in triples?  (SparseRowsDev: same coefficients, indices shifted by 0 / 3 / 6 inside one tet's block)
    bool triples = nr > 0 && nr % 3 == 0 && block == 9;
    if (triples) {
        std::vector<char> ok(64, 1);
        parallel_ranges(nr / 3, 4096, [&](int64_t u0, int64_t u1, int t) {
            bool good = true;
            for (int64_t u = u0; good && u < u1; ++u) {
                const uint32_t len = ptr[3 * u + 1] - ptr[3 * u];
                for (int c = 1; good && c < 3; ++c) {
                    good = ptr[3 * u + c + 1] - ptr[3 * u + c] == len;
                    uint64_t p0 = d.rowptr[3 * u], pc = d.rowptr[3 * u + c];
                    for (uint32_t q = 0; good && q < len; ++q, ++p0, ++pc) {
                        uint32_t i0, ic;
                        while (!local(p0, i0)) ++p0;  // (len kept entries are known to follow)
                        while (!local(pc, ic)) ++pc;
                        good = ic == i0 + 3u * c && d.coef[pc] == d.coef[p0] && i0 % 9 < 3;
                    }
                }
            }
            if (!good) ok[t % 64] = 0;
        });
        for (char c : ok) triples = triples && c;
    }
    if (std::getenv("SANM_DEBUG"))
        std::fprintf(stderr, "remap_out: %ld rows, %u entries, rows in triples: %s\n", (long)nr, ptr[nr], triples ? "yes" : "no");
    // rows [r0 .. ) of the source, every `step`-th one, packed: the whole table (step 1) or the first row of each triple
    auto pack = [&](int step, std::vector<uint32_t>& optr, std::unique_ptr<uint32_t[]>& oidx, std::unique_ptr<double[]>& ocoef) {
        const int64_t rows = nr / step;
        optr.assign(rows + 1, 0);
        for (int64_t u = 0; u < rows; ++u) optr[u + 1] = optr[u] + (ptr[step * u + 1] - ptr[step * u]);
        oidx = raw_array<uint32_t>(optr[rows]);  // (the workers touch their own pages first)
        ocoef = raw_array<double>(optr[rows]);
        parallel_ranges(rows, 4096, [&](int64_t u0, int64_t u1, int) {
            for (int64_t u = u0; u < u1; ++u) {
                uint32_t w = optr[u], li;
                for (uint64_t p = d.rowptr[step * u]; p < d.rowptr[step * u + 1]; ++p) {
                    if (!local(p, li)) continue;
                    oidx[w] = li;
                    ocoef[w++] = d.coef[p];
                }
            }
        });
    };
    laps.lap("counts, triples");
    Packed out;
    out.nr = nr;
    out.triples = triples;
    pack(triples ? 3 : 1, out.ptr, out.idx, out.coef);
    laps.lap("pack");
    return out;
}

DeviceRows::DeviceRows(Backend* be, Packed&& rows) : m_be{be} {
    SetupLaps laps("remap_out rows");
    const size_t nent = rows.ptr.empty() ? 0 : rows.ptr.back();
    const int64_t nr = rows.nr;
    void* dptr = be->alloc(std::max<size_t>(rows.ptr.size(), 1) * 4);
    void* didx = be->alloc(std::max<size_t>(nent, 1) * 4);
    void* dcoef = be->alloc(std::max<size_t>(nent, 1) * 8);
    be->h2d(dptr, rows.ptr.data(), rows.ptr.size() * 4);
    be->h2d(didx, rows.idx.get(), nent * 4);
    be->h2d(dcoef, rows.coef.get(), nent * 8);
    laps.lap("upload");
    if (rows.triples) {
        // only the list of the first row of each triple is kept (row_ops.h: gather_row reads rows 3u+1, 3u+2 through it)
        m_bptr = dptr, m_bidx = didx, m_bcoef = dcoef;
        m_dev = {nullptr, nullptr, nullptr, nr, static_cast<uint32_t*>(dptr), static_cast<uint32_t*>(didx),
                 static_cast<double*>(dcoef)};
    } else {
        m_ptr = dptr, m_idx = didx, m_coef = dcoef;
        m_dev = {static_cast<uint32_t*>(dptr), static_cast<uint32_t*>(didx), static_cast<double*>(dcoef), nr,
                 nullptr, nullptr, nullptr};
    }
}

DeviceRows::~DeviceRows() {
    if (m_ptr) {
        m_be->free(m_ptr);
        m_be->free(m_idx);
        m_be->free(m_coef);
    }
    if (m_bptr) {
        m_be->free(m_bptr);
        m_be->free(m_bidx);
        m_be->free(m_bcoef);
    }
}

template <class T>
T* JacobianPattern::upload(const T* v, size_t count) {
    void* p = m_be->alloc(std::max<size_t>(count, 1) * sizeof(T));
    if (count) m_be->h2d(p, v, count * sizeof(T));
    m_bufs.push_back(p);
    return static_cast<T*>(p);
}
template <class T>
T* JacobianPattern::upload(const std::vector<T>& v) {
    return upload(v.data(), v.size());
}

JacobianPattern::JacobianPattern(Backend* be, const SparseDesc& ro, const SparseDesc& ri, int64_t n,
                                 int64_t T, int64_t Tpad, int odim, int64_t tet_begin, int64_t tet_end, int idim,
                                 const int64_t* tet_order, const int64_t* tet_inv, bool defer_device,
                                 const std::function<void(BlockRows, BlockRows)>& on_blocks)
        : m_be{be} {
    sanm_check(!tet_order == !tet_inv, "a renumbering of the batch items comes with its inverse");
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
    SetupLaps laps("pattern");
    const bool sharded = tet_begin != 0 || tet_end != T;
    struct Part {
        std::vector<uint32_t> row_nnz, col;
        int64_t contrib = 0;
        std::string error;
    };
    const int nthread = (int)std::max<int64_t>(
            1, std::min<int64_t>({(int64_t)host_thread_cap(), n / 2048 + 1}));
    std::vector<Part> parts(nthread);
    // the distinct columns of every batch item, once (a tet's 9 remap_in rows list its 12 unknowns three times over, and
    // every one of the ~24 tets around a vertex was walked in full for each of that vertex's rows: round 6); the list of
    // item b starts where its remap_in rows start
    auto item_col_raw = raw_array<uint32_t>(ri.rowptr[(size_t)T * idim]);
    uint32_t* const item_col = item_col_raw.get();
    std::vector<uint32_t> item_nr(T);
    parallel_ranges(T, 4096, [&](int64_t b0, int64_t b1, int) {
        for (int64_t b = b0; b < b1; ++b) {
            const uint64_t q0 = ri.rowptr[b * idim], q1 = ri.rowptr[(b + 1) * idim];
            uint32_t* out = item_col + q0;
            uint32_t m = 0;
            for (uint64_t q = q0; q < q1; ++q) out[m++] = (uint32_t)ri.idx[q];
            std::sort(out, out + m);
            item_nr[b] = (uint32_t)(std::unique(out, out + m) - out);
        }
    });
    laps.lap("columns of the items");
    std::vector<uint32_t>& rowptr = m_h_rowptr;
    std::vector<uint32_t>& col = m_h_col;
    // ---- 3 x 3 blocks (round 6): when the three rows of every vertex touch the same batch items and every item's columns
    // are whole triples, a vertex's rows are its BLOCK row written out three times over -- a third of the walk, marks over
    // vertices instead of unknowns, and the direct solver's analysis can start from the block rows before the rows of
    // the unknowns exist (on_blocks).  Anything else: the rows one by one, below.  SANM_PATTERN_NO_BLOCKS: never.
    bool blocked = n % 3 == 0 && n >= 3 && !m_has_t && !std::getenv("SANM_PATTERN_NO_BLOCKS");
    if (blocked) {
        std::vector<char> ok(64, 1);
        parallel_ranges(T, 4096, [&](int64_t b0, int64_t b1, int t) {
            bool good = true;
            for (int64_t b = b0; good && b < b1; ++b) {
                const uint32_t* lc = item_col + ri.rowptr[b * idim];
                const uint32_t m = item_nr[b];
                good = m % 3 == 0;
                for (uint32_t q = 0; good && q < m; q += 3)
                    good = lc[q] % 3 == 0 && lc[q + 1] == lc[q] + 1 && lc[q + 2] == lc[q] + 2 && (int64_t)lc[q + 2] < n;
            }
            if (!good) ok[t % 64] = 0;
        });
        for (char c : ok) blocked = blocked && c;
    }
    if (blocked) {
        const int64_t nv = n / 3;
        struct VPart {
            std::vector<uint32_t> len, col;
            int64_t contrib = 0;
            bool good = true;
            std::string error;
        };
        const int nvt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)host_thread_cap(), nv / 1024 + 1}));
        std::vector<VPart> vparts(nvt);
        auto vbuild = [&](int t) {
            VPart& P = vparts[t];
            const int64_t v0 = nv * t / nvt, v1 = nv * (t + 1) / nvt;
            std::vector<uint8_t> mark(nv, 0);
            std::vector<uint32_t> ucol;
            std::vector<uint64_t> items, other;
            for (int64_t v = v0; v < v1 && P.good; ++v) {
                for (int c = 0; c < 3; ++c) {
                    std::vector<uint64_t>& its = c == 0 ? items : other;
                    its.clear();
                    for (uint64_t p = ro.rowptr[3 * v + c]; p < ro.rowptr[3 * v + c + 1]; ++p) {
                        const uint64_t b = odim == 9 ? ro.idx[p] / 9 : ro.idx[p] / odim;
                        bool mine = true;
                        if (sharded) {
                            const int64_t bn = tet_inv ? tet_inv[b] : (int64_t)b;
                            mine = bn >= tet_begin && bn < tet_end;
                        }
                        if (mine) P.contrib += (int64_t)(ri.rowptr[(b + 1) * idim] - ri.rowptr[b * idim]);
                        if (its.empty() || its.back() != b) its.push_back(b);
                    }
                    if (c > 0 && other != items) P.good = false;
                }
                ucol.clear();
                for (uint64_t b : items) {
                    const uint32_t* lc = item_col + ri.rowptr[b * idim];
                    for (uint32_t q = 0; q < item_nr[b]; q += 3) {
                        const uint32_t u = lc[q] / 3;
                        if (!mark[u]) {
                            mark[u] = 1;
                            ucol.push_back(u);
                        }
                    }
                }
                std::sort(ucol.begin(), ucol.end());
                for (uint32_t u : ucol) mark[u] = 0;
                if (ucol.empty()) P.error = "empty row " + std::to_string(3 * v);  // sparse_solver.cpp:251-252
                P.col.insert(P.col.end(), ucol.begin(), ucol.end());
                P.len.push_back((uint32_t)ucol.size());
            }
        };
        {
            JoinedThreads jt;
            for (int t = 1; t < nvt; ++t) jt.run([&, t] { vbuild(t); });
            vbuild(0);
        }
        for (const VPart& P : vparts) {
            sanm_check(P.error.empty(), "%s", P.error.c_str());
            blocked = blocked && P.good;
        }
        if (blocked) {
            auto qptr = std::make_shared<std::vector<uint32_t>>(nv + 1, 0);
            auto qcol = std::make_shared<std::vector<uint32_t>>();
            {
                size_t nq = 0;
                int64_t v = 0;
                for (const VPart& P : vparts) {
                    for (uint32_t l : P.len) {
                        (*qptr)[v + 1] = (*qptr)[v] + l;
                        ++v;
                    }
                    nq += P.col.size();
                    m_nr_contrib += P.contrib;
                }
                sanm_check(nq * 9 < std::numeric_limits<uint32_t>::max(), "Jacobian pattern too large");
                qcol->resize(nq);
                size_t at = 0;
                for (const VPart& P : vparts) {
                    std::copy(P.col.begin(), P.col.end(), qcol->begin() + at);
                    at += P.col.size();
                }
            }
            laps.lap("block rows");
            if (on_blocks) on_blocks(qptr, qcol);
            // the rows of the unknowns: row 3 v + c = the unknowns of the blocks of block row v
            rowptr.resize(n + 1);
            col.resize(qcol->size() * 9);
            const std::vector<uint32_t>&QP = *qptr, &QC = *qcol;
            parallel_ranges(nv, 2048, [&](int64_t v0, int64_t v1, int) {
                for (int64_t v = v0; v < v1; ++v) {
                    const uint32_t len = QP[v + 1] - QP[v];
                    for (int c = 0; c < 3; ++c) {
                        const uint32_t at = 9 * QP[v] + c * 3 * len;
                        rowptr[3 * v + c] = at;
                        uint32_t* out = col.data() + at;
                        for (uint32_t q = 0; q < len; ++q) {
                            const uint32_t u3 = 3 * QC[QP[v] + q];
                            out[3 * q] = u3, out[3 * q + 1] = u3 + 1, out[3 * q + 2] = u3 + 2;
                        }
                    }
                }
            });
            rowptr[n] = (uint32_t)col.size();
            laps.lap("rows written out");
        } else {
            m_nr_contrib = 0;
        }
    }
    if (!blocked) {
    auto build = [&](int t) {
        Part& P = parts[t];
        const int64_t r0 = n * t / nthread, r1 = n * (t + 1) / nthread;
        std::vector<uint8_t> mark(n + 1, 0);
        std::vector<uint32_t> ucol;
        try {
            // a row whose batch items are those of the row before it (the three components of a vertex) has its columns
            std::vector<uint64_t> items, prev_items;
            uint32_t prev_nnz = 0;
            for (int64_t i = r0; i < r1; ++i) {
                items.clear();
                for (uint64_t p = ro.rowptr[i]; p < ro.rowptr[i + 1]; ++p) {
                    // (the caller's numbering, like the rows of ri; 9: a division the compiler turns into a multiply)
                    const uint64_t b = odim == 9 ? ro.idx[p] / 9 : ro.idx[p] / odim;
                    bool mine = true;
                    if (sharded) {
                        const int64_t bn = tet_inv ? tet_inv[b] : (int64_t)b;
                        mine = bn >= tet_begin && bn < tet_end;
                    }
                    if (mine) P.contrib += (int64_t)(ri.rowptr[(b + 1) * idim] - ri.rowptr[b * idim]);
                    if (items.empty() || items.back() != b) items.push_back(b);
                }
                if (i > r0 && items == prev_items) {
                    const size_t from = P.col.size() - prev_nnz;
                    for (uint32_t q = 0; q < prev_nnz; ++q) {
                        const uint32_t c = P.col[from + q];  // (by value: push_back may move the storage)
                        P.col.push_back(c);
                    }
                    P.row_nnz.push_back(prev_nnz);
                    continue;
                }
                ucol.clear();
                for (uint64_t b : items) {
                    const uint32_t* lc = item_col + ri.rowptr[b * idim];
                    for (uint32_t q = 0; q < item_nr[b]; ++q) {
                        const uint32_t c = lc[q];
                        if (!mark[c]) {
                            mark[c] = 1;
                            ucol.push_back(c);
                        }
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
                prev_nnz = nnz_row;
                prev_items.swap(items);
            }
        } catch (const SanmError& e) {
            P.error = e.msg;
        }
    };
    {
        JoinedThreads jt;
        for (int t = 1; t < nthread; ++t) jt.run([&, t] { build(t); });
        build(0);
    }
    for (const Part& P : parts) sanm_check(P.error.empty(), "%s", P.error.c_str());
    laps.lap("rows");
    rowptr.assign(n + 1, 0);
    {
        size_t ncol = 0;
        std::vector<size_t> part_off;
        for (const Part& P : parts) {
            part_off.push_back(ncol);
            ncol += P.col.size();
        }
        sanm_check(ncol < std::numeric_limits<uint32_t>::max(), "Jacobian pattern too large");
        col.resize(ncol);
        int64_t i = 0;
        for (const Part& P : parts) {
            for (size_t r = 0; r < P.row_nnz.size(); ++r, ++i) rowptr[i + 1] = rowptr[i] + P.row_nnz[r];
            m_nr_contrib += P.contrib;
        }
        JoinedThreads jt;
        for (int t = 1; t < nthread; ++t)
            jt.run([&, t] { std::copy(parts[t].col.begin(), parts[t].col.end(), col.begin() + part_off[t]); });
        std::copy(parts[0].col.begin(), parts[0].col.end(), col.begin());
    }
    laps.lap("merge");
    }  // (the rows one by one)
    m_n = n, m_T = T, m_tet_begin = tet_begin, m_tet_end = tet_end, m_odim = odim, m_idim = idim;
    m_tet_order = tet_order, m_tet_inv = tet_inv;
    if (!defer_device) finish_device(ro, ri);
}

void JacobianPattern::finish_device(const SparseDesc& ro, const SparseDesc& ri) {
    SetupLaps laps("pattern");
    Backend* be = m_be;
    const int64_t n = m_n, T = m_T, tet_begin = m_tet_begin, tet_end = m_tet_end;
    const int odim = m_odim, idim = m_idim;
    const int64_t *tet_order = m_tet_order, *tet_inv = m_tet_inv;
    const std::vector<uint32_t>& rowptr = m_h_rowptr;
    const std::vector<uint32_t>& col = m_h_col;
    m_tet_order = m_tet_inv = nullptr;  // (the caller's arrays need not outlive this call)
    m_csr.n = n;
    m_csr.nnz = col.size();
    m_csr.rowptr = upload(rowptr);
    m_csr.col = upload(col);
    {
        void* val = m_be->alloc(std::max<size_t>(col.size(), 1) * sizeof(double));
        m_be->zero(val, col.size() * sizeof(double));
        m_bufs.push_back(val);
        m_csr.val = static_cast<double*>(val);
    }
    laps.lap("csr upload");
    // the remap tables as the assembly reads them: 32-bit, in the renumbered batch order
    {
        std::vector<uint32_t> v32(ro.rowptr.begin(), ro.rowptr.end());
        m_asm.ro_ptr = upload(v32);
        auto o32 = raw_array<uint32_t>(ro.idx.size());
        parallel_ranges((int64_t)ro.idx.size(), 1 << 18, [&](int64_t q0, int64_t q1, int) {
            for (int64_t q = q0; q < q1; ++q)
                o32[q] = (uint32_t)(tet_inv ? tet_inv[ro.idx[q] / odim] * odim + ro.idx[q] % odim : ro.idx[q]);
        });
        laps.lap("remap tables: ro convert");
        m_asm.ro_idx = upload(o32.get(), ro.idx.size());
        m_asm.ro_coef = upload(ro.coef);
        laps.lap("remap tables: ro upload");
        // rows of ri: batch item e of the table is the caller's item tet_order[e]
        const int64_t nrow = ri.out_size;
        v32.assign(nrow + 1, 0);
        parallel_ranges(T, 4096, [&](int64_t e0, int64_t e1, int) {
            for (int64_t e = e0; e < e1; ++e) {
                const int64_t src = (tet_order ? tet_order[e] : e) * idim;
                for (int m = 0; m < idim; ++m) v32[e * idim + m + 1] = (uint32_t)(ri.rowptr[src + m + 1] - ri.rowptr[src + m]);
            }
        });
        for (int64_t r = 0; r < nrow; ++r) v32[r + 1] += v32[r];
        m_asm.ri_ptr = upload(v32);
        auto i32 = raw_array<uint32_t>(ri.idx.size());
        auto c64 = raw_array<double>(tet_order ? ri.coef.size() : 0);
        parallel_ranges(T, 4096, [&](int64_t e0, int64_t e1, int) {
            for (int64_t e = e0; e < e1; ++e) {
                const int64_t src = (tet_order ? tet_order[e] : e) * idim;
                uint32_t w = v32[e * idim];
                for (uint64_t q = ri.rowptr[src]; q < ri.rowptr[src + idim]; ++q, ++w) {
                    i32[w] = (uint32_t)ri.idx[q];
                    if (tet_order) c64[w] = ri.coef[q];
                }
            }
        });
        laps.lap("remap tables: ri convert");
        m_asm.ri_idx = upload(i32.get(), ri.idx.size());
        m_asm.ri_coef = upload(tet_order ? c64.get() : ri.coef.data(), ri.coef.size());
    }
    laps.lap("remap tables");
    m_asm.rowptr = m_csr.rowptr;
    m_asm.col = m_csr.col;
    m_asm.n = n;
    m_asm.odim = odim;
    m_asm.idim = idim;
    m_asm.tet_begin = tet_begin;
    m_asm.tet_end = tet_end;
    m_asm.has_t = m_has_t ? 1 : 0;
    m_asm.nnz = m_csr.nnz;
    // rows in triples?  (AssemblyDev::triples: the three components of a vertex share their columns and, up to a shift
    // of the Jacobian index, their gather lists)
    bool triples = n > 0 && n % 3 == 0 && odim == 9 && !m_has_t && !std::getenv("SANM_ASM_NO_TRIPLES");
    if (triples) {
        std::vector<char> ok(64, 1);
        parallel_ranges(n / 3, 4096, [&](int64_t u0, int64_t u1, int t) {
            bool good = true;
            for (int64_t u = u0; good && u < u1; ++u) {
                const uint64_t p0 = ro.rowptr[3 * u], len = ro.rowptr[3 * u + 1] - p0;
                const uint32_t c0 = rowptr[3 * u], clen = rowptr[3 * u + 1] - c0;
                for (int c = 1; good && c < 3; ++c) {
                    const uint64_t pc = ro.rowptr[3 * u + c];
                    good = ro.rowptr[3 * u + c + 1] - pc == len && rowptr[3 * u + c + 1] - rowptr[3 * u + c] == clen &&
                           rowptr[3 * u + c] == c0 + c * clen &&
                           std::equal(col.begin() + c0, col.begin() + c0 + clen, col.begin() + rowptr[3 * u + c]);
                    for (uint64_t q = 0; good && q < len; ++q)
                        good = ro.idx[pc + q] == ro.idx[p0 + q] + 3u * c && ro.coef[pc + q] == ro.coef[p0 + q] &&
                               ro.idx[p0 + q] % 9 < 3;
                }
            }
            if (!good) ok[t % 64] = 0;
        });
        for (char c : ok) triples = triples && c;
    }
    if (triples) {
        const size_t nts = col.size() / 3;
        auto tp = raw_array<uint32_t>(nts);
        auto tl = raw_array<uint32_t>(nts);
        parallel_ranges(n / 3, 4096, [&](int64_t u0, int64_t u1, int) {
            for (int64_t u = u0; u < u1; ++u) {
                const uint32_t c0 = rowptr[3 * u], clen = rowptr[3 * u + 1] - c0;
                for (uint32_t k = 0; k < clen; ++k) {
                    tp[c0 / 3 + k] = c0 + k;  // (rowptr[3u] = 3 x the lengths of the triples before u)
                    tl[c0 / 3 + k] = clen;
                }
            }
        });
        m_asm.triples = 1;
        m_asm.ntslot = (int64_t)nts;
        m_asm.tslot_p = upload(tp.get(), nts);
        m_asm.tslot_len = upload(tl.get(), nts);
    }
    be->prepare_assembly(m_asm, m_bufs);
    laps.lap("assembly lists");
}

JacobianPattern::~JacobianPattern() {
    for (void* p : m_bufs) m_be->free(p);
}

}  // namespace sanm_hip
