#include "multifrontal.h"
#include "host_parallel.h"

#include <algorithm>
#include <atomic>
#include <memory>
#include <chrono>
#include <thread>
#include <future>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>

namespace sanm_hip {

namespace {

// ---------------------------------------------------------------------------
// compressed (supervariable) graph
// ---------------------------------------------------------------------------
struct SvGraph {
    int32_t nsv = 0;
    std::vector<int32_t> sv_of;      // unknown -> supervariable
    std::vector<int32_t> sv_ptr;     // members of each supervariable (original unknown ids)
    std::vector<int32_t> sv_members;
    std::vector<int32_t> adj_ptr, adj;  // symmetric adjacency without self loops
    std::vector<double> xyz;         // nsv*3 centroid coordinates (empty if unknown)
    int32_t size(int32_t s) const { return sv_ptr[s + 1] - sv_ptr[s]; }
};

//! the supervariables of one pattern by hashing the closed neighbourhoods of A + A' (no coordinates)
SvGraph sv_graph_by_hash(int64_t n, const std::vector<uint32_t>& rowptr, const std::vector<uint32_t>& col) {
    SetupLaps laps("svgraph");
    std::vector<int32_t> uptr(n + 1, 0);
    std::vector<uint64_t> hash(n);
    std::unique_ptr<int32_t[]> unb_raw;
    int32_t* nb = nullptr;
    // A pattern that is symmetric with ascending rows -- the block rows of a mesh Jacobian -- IS its symmetrised adjacency:
    // the closed neighbourhoods are the rows (with the diagonal put in where it is missing), no transposed entries to
    // collect and nothing to sort (round 6: a third of this function's time).  Checked, not assumed.
    bool plain = true;
    {
        std::vector<char> ok(64, 1);
        parallel_ranges(n, 2048, [&](int64_t r0, int64_t r1, int t) {
            bool good = true;
            for (int64_t i = r0; good && i < r1; ++i)
                for (uint32_t p = rowptr[i]; good && p < rowptr[i + 1]; ++p) {
                    const int64_t j = col[p];
                    good = j < n && (p == rowptr[i] || col[p - 1] < col[p]);
                    if (good && j != i) good = std::binary_search(col.begin() + rowptr[j], col.begin() + rowptr[j + 1], (uint32_t)i);
                }
            if (!good) ok[t % 64] = 0;
        });
        for (char c : ok) plain = plain && c;
    }
    if (plain) {
        std::vector<int32_t> ulen(n);
        parallel_ranges(n, 2048, [&](int64_t r0, int64_t r1, int) {
            for (int64_t i = r0; i < r1; ++i)
                ulen[i] = (int32_t)(rowptr[i + 1] - rowptr[i]) +
                          (std::binary_search(col.begin() + rowptr[i], col.begin() + rowptr[i + 1], (uint32_t)i) ? 0 : 1);
        });
        for (int64_t i = 0; i < n; ++i) uptr[i + 1] = uptr[i] + ulen[i];
        unb_raw = raw_array<int32_t>(uptr[n]);
        nb = unb_raw.get();
        parallel_ranges(n, 2048, [&](int64_t r0, int64_t r1, int) {
            for (int64_t i = r0; i < r1; ++i) {
                int32_t* out = nb + uptr[i];
                bool self = false;
                for (uint32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
                    const int32_t j = (int32_t)col[p];
                    if (!self && j >= i) {
                        if (j > i) *out++ = (int32_t)i;
                        self = true;
                    }
                    *out++ = j;
                }
                if (!self) *out++ = (int32_t)i;
                uint64_t h = 1469598103934665603ull;
                for (int32_t q = uptr[i]; q < uptr[i + 1]; ++q) h = (h ^ (uint64_t)nb[q]) * 1099511628211ull;
                hash[i] = h;
            }
        });
        laps.lap("rows as neighbourhoods");
    } else {
    // symmetrised adjacency including the diagonal.  Every thread scans ALL rows and keeps what lands in its own range
    // of unknowns (the entries of its rows and the transposed entries pointing into them): no shared counters, and
    // each list receives its entries in the order one thread would append them.
    std::vector<int32_t> deg(n + 1, 0);
    std::vector<std::string> errs(64);
    parallel_ranges(n, 2048, [&](int64_t r0, int64_t r1, int t) {
        for (int64_t i = 0; i < n; ++i) {
            const bool own_row = i >= r0 && i < r1;
            int32_t own = 0;
            for (uint32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
                const int64_t j = col[p];
                if (j >= n) {
                    errs[t % 64] = "column index out of range";
                    return;
                }
                if (j == i) continue;
                ++own;
                if (j >= r0 && j < r1) deg[j + 1]++;
            }
            if (own_row) deg[i + 1] += own;
        }
    });
    for (const auto& e : errs) sanm_check(e.empty(), "%s", e.c_str());
    for (int64_t i = 0; i < n; ++i) deg[i + 1] += deg[i] + 1;  // +1: self
    auto nb_raw = raw_array<int32_t>(deg[n]);
    nb = nb_raw.get();
    {
        std::vector<int32_t> fill(deg.begin(), deg.end() - 1);
        parallel_ranges(n, 2048, [&](int64_t r0, int64_t r1, int) {
            for (int64_t i = r0; i < r1; ++i) nb[fill[i]++] = (int32_t)i;
            for (int64_t i = 0; i < n; ++i) {
                const bool own_row = i >= r0 && i < r1;
                for (uint32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
                    const int64_t j = col[p];
                    if (j == i) continue;
                    if (own_row) nb[fill[i]++] = (int32_t)j;
                    if (j >= r0 && j < r1) nb[fill[j]++] = (int32_t)i;
                }
            }
        });
    }
    laps.lap("symmetrise");
    // sort + unique every list, hash it (rows in parallel, in place), then pack the lists
    {
        std::vector<int32_t> ulen(n);
        parallel_ranges(n, 2048, [&](int64_t r0, int64_t r1, int) {
            for (int64_t i = r0; i < r1; ++i) {
                int32_t b = deg[i], e = deg[i + 1];
                std::sort(nb + b, nb + e);
                int32_t ne = std::unique(nb + b, nb + e) - nb;
                uint64_t h = 1469598103934665603ull;
                for (int32_t q = b; q < ne; ++q) h = (h ^ (uint64_t)nb[q]) * 1099511628211ull;
                hash[i] = h;
                ulen[i] = ne - b;
            }
        });
        for (int64_t i = 0; i < n; ++i) uptr[i + 1] = uptr[i] + ulen[i];
        unb_raw = raw_array<int32_t>(uptr[n]);
        int32_t* dst = unb_raw.get();
        parallel_ranges(n, 2048, [&](int64_t r0, int64_t r1, int) {
            for (int64_t i = r0; i < r1; ++i) std::memcpy(dst + uptr[i], nb + deg[i], (size_t)ulen[i] * sizeof(int32_t));
        });
        nb_raw.reset();
        nb = dst;
    }
    laps.lap("sort lists, pack");
    }  // (the pattern as it comes)
    // group indistinguishable unknowns (same closed neighbourhood)
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    parallel_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        if (hash[a] != hash[b]) return hash[a] < hash[b];
        return a < b;
    });
    laps.lap("sort by hash");
    SvGraph g;
    g.sv_of.assign(n, -1);
    auto same = [&](int32_t a, int32_t b) {
        int32_t la = uptr[a + 1] - uptr[a], lb = uptr[b + 1] - uptr[b];
        return la == lb && std::equal(nb + uptr[a], nb + uptr[a + 1], nb + uptr[b]);
    };
    std::vector<int32_t> rep;  // representative unknown of each supervariable
    for (int64_t q = 0; q < n;) {
        int64_t e = q + 1;
        while (e < n && hash[order[e]] == hash[order[q]]) ++e;
        // within one hash bucket compare against the bucket's representatives
        size_t first_rep = rep.size();
        for (int64_t t = q; t < e; ++t) {
            int32_t u = order[t];
            int32_t found = -1;
            for (size_t r = first_rep; r < rep.size(); ++r)
                if (same(rep[r], u)) {
                    found = r;
                    break;
                }
            if (found < 0) {
                found = rep.size();
                rep.push_back(u);
            }
            g.sv_of[u] = found;
        }
        q = e;
    }
    laps.lap("group");
    // renumber supervariables by their smallest member to keep locality
    g.nsv = rep.size();
    {
        std::vector<int32_t> minmem(g.nsv, INT32_MAX);
        for (int64_t i = 0; i < n; ++i) minmem[g.sv_of[i]] = std::min<int32_t>(minmem[g.sv_of[i]], i);
        std::vector<int32_t> o(g.nsv);
        std::iota(o.begin(), o.end(), 0);
        std::sort(o.begin(), o.end(), [&](int32_t a, int32_t b) { return minmem[a] < minmem[b]; });
        std::vector<int32_t> newid(g.nsv);
        for (int32_t k = 0; k < g.nsv; ++k) newid[o[k]] = k;
        for (int64_t i = 0; i < n; ++i) g.sv_of[i] = newid[g.sv_of[i]];
    }
    g.sv_ptr.assign(g.nsv + 1, 0);
    for (int64_t i = 0; i < n; ++i) g.sv_ptr[g.sv_of[i] + 1]++;
    for (int32_t s = 0; s < g.nsv; ++s) g.sv_ptr[s + 1] += g.sv_ptr[s];
    g.sv_members.resize(n);
    {
        std::vector<int32_t> fill(g.sv_ptr.begin(), g.sv_ptr.end() - 1);
        for (int64_t i = 0; i < n; ++i) g.sv_members[fill[g.sv_of[i]]++] = i;
    }
    laps.lap("renumber, members");
    // compressed adjacency from the first member of each supervariable (lists by ranges of supervariables, then
    // joined in order)
    g.adj_ptr.assign(g.nsv + 1, 0);
    {
        std::vector<std::vector<int32_t>> piece(64);
        std::vector<std::pair<int32_t, int32_t>> piece_range(64, {0, 0});
        parallel_ranges(g.nsv, 4096, [&](int64_t s0, int64_t s1, int t) {
            std::vector<int32_t>& out = piece[t];
            piece_range[t] = {(int32_t)s0, (int32_t)s1};
            std::vector<int32_t> tmp;
            for (int32_t s = (int32_t)s0; s < (int32_t)s1; ++s) {
                const int32_t u = g.sv_members[g.sv_ptr[s]];
                tmp.clear();
                for (int32_t q = uptr[u]; q < uptr[u + 1]; ++q) {
                    const int32_t v = g.sv_of[nb[q]];
                    if (v != s) tmp.push_back(v);
                }
                std::sort(tmp.begin(), tmp.end());
                tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
                out.insert(out.end(), tmp.begin(), tmp.end());
                g.adj_ptr[s + 1] = (int32_t)tmp.size();
            }
        });
        for (int32_t s = 0; s < g.nsv; ++s) g.adj_ptr[s + 1] += g.adj_ptr[s];
        g.adj.resize(g.adj_ptr[g.nsv]);
        for (int t = 0; t < 64; ++t)
            if (!piece[t].empty()) std::copy(piece[t].begin(), piece[t].end(), g.adj.begin() + g.adj_ptr[piece_range[t].first]);
    }
    laps.lap("adjacency");
    return g;
}

// The Jacobian of a 3D mesh has a 3 x 3 block for every pair of vertices: rows 3v .. 3v+2 list the same columns, and
// hashing 235 k closed neighbourhoods of 37 entries each to find that out was 0.13 of the 0.5 s of the analysis (round 6).
// Runs of CONSECUTIVE rows with the same column list that holds their own diagonals are found by comparing neighbours;
// the pattern of the runs (a ninth of the entries) then goes through the hashing above, which also finds what the runs
// do not (equal neighbourhoods that are not consecutive rows).  Rows of a run are indistinguishable in A + A' when the
// pattern is symmetric -- the supervariables, their numbers and the graph are then exactly what the hashing of the
// whole pattern gives (tests/test_direct_solver.py) --; in an unsymmetric pattern the members of a run may differ in
// their COLUMNS, and the run is then a supervariable with explicit zeros: every entry (i, j) of A still has its edge,
// because row i is the row the run's edges were taken from.  SANM_MF_SV_RUNS=0: hash the whole pattern.
SvGraph build_sv_graph(int64_t n, const std::vector<uint32_t>& rowptr,
                       const std::vector<uint32_t>& col, const double* coords) {
    SvGraph g;
    bool by_runs = false;
    const bool use_runs = !(std::getenv("SANM_MF_SV_RUNS") && std::atoi(std::getenv("SANM_MF_SV_RUNS")) == 0);
    if (use_runs && n >= 64) {
        SetupLaps laps("svruns");
        // head[i]: row i starts a run
        std::vector<uint8_t> head(n, 1);
        parallel_ranges(n, 4096, [&](int64_t r0, int64_t r1, int) {
            for (int64_t i = std::max<int64_t>(r0, 1); i < r1; ++i) {
                const uint32_t b = rowptr[i], e = rowptr[i + 1], pb = rowptr[i - 1];
                if (e - b != b - pb || std::memcmp(&col[b], &col[pb], (size_t)(e - b) * sizeof(uint32_t)) != 0) continue;
                bool di = false, dp = false;  // the diagonals of both rows
                for (uint32_t p = b; p < e; ++p) {
                    di = di || col[p] == (uint32_t)i;
                    dp = dp || col[p] == (uint32_t)(i - 1);
                }
                if (di && dp) head[i] = 0;
            }
        });
        std::vector<int32_t> run_of(n);
        std::vector<int32_t> first;  // first row of every run
        for (int64_t i = 0; i < n; ++i) {
            if (head[i]) first.push_back((int32_t)i);
            run_of[i] = (int32_t)first.size() - 1;
        }
        const int64_t nr = (int64_t)first.size();
        laps.lap("runs");
        if (nr * 4 <= n * 3) {
            // the pattern of the runs, from the first row of each
            std::vector<uint32_t> qptr(nr + 1, 0);
            for (int64_t r = 0; r < nr; ++r) qptr[r + 1] = qptr[r] + (rowptr[first[r] + 1] - rowptr[first[r]]);
            std::vector<uint32_t> qcol(qptr[nr]);
            std::vector<uint32_t> qlen(nr);
            std::vector<std::string> errs(64);
            parallel_ranges(nr, 2048, [&](int64_t r0, int64_t r1, int t) {
                for (int64_t r = r0; r < r1; ++r) {
                    uint32_t* out = qcol.data() + qptr[r];
                    uint32_t m = 0;
                    for (uint32_t p = rowptr[first[r]]; p < rowptr[first[r] + 1]; ++p) {
                        if ((int64_t)col[p] >= n) {
                            errs[t % 64] = "column index out of range";
                            return;
                        }
                        out[m++] = (uint32_t)run_of[col[p]];
                    }
                    if (!std::is_sorted(out, out + m)) std::sort(out, out + m);
                    qlen[r] = (uint32_t)(std::unique(out, out + m) - out);
                }
            });
            for (const auto& e : errs) sanm_check(e.empty(), "%s", e.c_str());
            std::vector<uint32_t> cptr(nr + 1, 0);
            for (int64_t r = 0; r < nr; ++r) cptr[r + 1] = cptr[r] + qlen[r];
            std::vector<uint32_t> ccol(cptr[nr]);
            parallel_ranges(nr, 2048, [&](int64_t r0, int64_t r1, int) {
                for (int64_t r = r0; r < r1; ++r) std::memcpy(ccol.data() + cptr[r], qcol.data() + qptr[r], (size_t)qlen[r] * sizeof(uint32_t));
            });
            laps.lap("pattern of the runs");
            SvGraph q = sv_graph_by_hash(nr, cptr, ccol);
            // supervariables of runs -> supervariables of unknowns (numbered by their smallest run = smallest unknown)
            g.nsv = q.nsv;
            g.adj_ptr = std::move(q.adj_ptr);
            g.adj = std::move(q.adj);
            g.sv_of.resize(n);
            parallel_ranges(n, 8192, [&](int64_t r0, int64_t r1, int) {
                for (int64_t i = r0; i < r1; ++i) g.sv_of[i] = q.sv_of[run_of[i]];
            });
            g.sv_ptr.assign(g.nsv + 1, 0);
            for (int64_t i = 0; i < n; ++i) g.sv_ptr[g.sv_of[i] + 1]++;
            for (int32_t s2 = 0; s2 < g.nsv; ++s2) g.sv_ptr[s2 + 1] += g.sv_ptr[s2];
            g.sv_members.resize(n);
            {
                std::vector<int32_t> fill(g.sv_ptr.begin(), g.sv_ptr.end() - 1);
                for (int64_t i = 0; i < n; ++i) g.sv_members[fill[g.sv_of[i]]++] = (int32_t)i;
            }
            laps.lap("expand");
            by_runs = true;
        }
    }
    if (!by_runs) g = sv_graph_by_hash(n, rowptr, col);
    if (coords) {
        g.xyz.assign((size_t)g.nsv * 3, 0.0);
        for (int32_t s = 0; s < g.nsv; ++s) {
            for (int32_t q = g.sv_ptr[s]; q < g.sv_ptr[s + 1]; ++q)
                for (int d = 0; d < 3; ++d) g.xyz[s * 3 + d] += coords[(int64_t)g.sv_members[q] * 3 + d];
            for (int d = 0; d < 3; ++d) g.xyz[s * 3 + d] /= g.size(s);
        }
    }
    return g;
}

//! the same graph from the pattern of the BLOCKS (Multifrontal::BlockPattern): what build_sv_graph finds from the expanded
//! rows -- there the runs of equal rows are the blocks (or runs of blocks with one neighbourhood, which the hashing of the
//! blocks' pattern joins just as well), the numbering goes by the smallest member either way
SvGraph build_sv_graph_of_blocks(int64_t n, int block, const std::vector<uint32_t>& qptr, const std::vector<uint32_t>& qcol,
                                 const double* coords) {
    const int64_t nq = n / block;
    SvGraph q = sv_graph_by_hash(nq, qptr, qcol);
    SvGraph g;
    g.nsv = q.nsv;
    g.adj_ptr = std::move(q.adj_ptr);
    g.adj = std::move(q.adj);
    g.sv_of.resize(n);
    parallel_ranges(n, 8192, [&](int64_t r0, int64_t r1, int) {
        for (int64_t i = r0; i < r1; ++i) g.sv_of[i] = q.sv_of[i / block];
    });
    g.sv_ptr.assign(g.nsv + 1, 0);
    for (int64_t i = 0; i < n; ++i) g.sv_ptr[g.sv_of[i] + 1]++;
    for (int32_t s2 = 0; s2 < g.nsv; ++s2) g.sv_ptr[s2 + 1] += g.sv_ptr[s2];
    g.sv_members.resize(n);
    {
        std::vector<int32_t> fill(g.sv_ptr.begin(), g.sv_ptr.end() - 1);
        for (int64_t i = 0; i < n; ++i) g.sv_members[fill[g.sv_of[i]]++] = (int32_t)i;
    }
    if (coords) {
        g.xyz.assign((size_t)g.nsv * 3, 0.0);
        for (int32_t s = 0; s < g.nsv; ++s) {
            for (int32_t p = g.sv_ptr[s]; p < g.sv_ptr[s + 1]; ++p)
                for (int d = 0; d < 3; ++d) g.xyz[s * 3 + d] += coords[(int64_t)g.sv_members[p] * 3 + d];
            for (int d = 0; d < 3; ++d) g.xyz[s * 3 + d] /= g.size(s);
        }
    }
    return g;
}

// ---------------------------------------------------------------------------
// nested dissection
// ---------------------------------------------------------------------------
struct NdNode {
    std::vector<int32_t> vars;  // own supervariables
    int32_t parent = -1;
    std::vector<int32_t> children;
};

class NestedDissection {
    const SvGraph& g;
    std::vector<int32_t> stamp, dist, queue;
    int32_t cur_stamp = 0;
    int threads = 1;  // host threads bisect() may use for its candidate cuts
    // fn(begin, end, piece) over pieces of [0, n) whose bounds are multiples of `align`, on this dissection's threads when
    // the set is big (the cuts at the top of the tree of a 2.7 M-tet mesh are the critical path of the constructor)
    template <class F>
    void over_pieces(size_t n, size_t align, F&& fn) const {
        // (from 256 k supervariables: on the GPU box's cores the threads cost a 78 k-supervariable root cut 0.02 s more than
        // they saved, session r6ab1, and a 538 k one breaks even, r6ab2; on 8 slower cores the dissection of the latter went
        // from 1.19 to 0.99 s.  SANM_MF_ND_TOP_SERIAL: never)
        static const bool serial = std::getenv("SANM_MF_ND_TOP_SERIAL") != nullptr;
        const int nt = n >= 262144 && !serial ? std::max(1, std::min(threads, 16)) : 1;
        if (nt <= 1) {
            fn((size_t)0, n, 0);
            return;
        }
        auto bound = [&](int t) { return t >= nt ? n : std::min(n, (n * t / nt) / align * align); };
        JoinedThreads jt;
        for (int t = 1; t < nt; ++t) jt.run([&, t] { fn(bound(t), bound(t + 1), t); });
        fn(bound(0), bound(1), 0);
    }
    const int LEAF = std::getenv("SANM_MF_LEAF") ? std::atoi(std::getenv("SANM_MF_LEAF")) : 32;

public:
    std::vector<NdNode> nodes;
    explicit NestedDissection(const SvGraph& g_) : g{g_}, stamp(g_.nsv, 0), dist(g_.nsv, 0) {}

    // BFS inside the set marked with `mark`; returns visit order in `queue`
    int32_t bfs(int32_t start, int32_t mark) {
        ++cur_stamp;
        queue.clear();
        queue.push_back(start);
        stamp[start] = cur_stamp;
        dist[start] = 0;
        for (size_t h = 0; h < queue.size(); ++h) {
            int32_t u = queue[h];
            for (int32_t q = g.adj_ptr[u]; q < g.adj_ptr[u + 1]; ++q) {
                int32_t v = g.adj[q];
                if (in_set[v] == mark && stamp[v] != cur_stamp) {
                    stamp[v] = cur_stamp;
                    dist[v] = dist[u] + 1;
                    queue.push_back(v);
                }
            }
        }
        return queue.back();
    }

    std::vector<int32_t> in_set;  // set id of every supervariable (-1: none)
    int32_t next_set = 0;

    // Subtrees of the dissection are independent of each other -- bisect() looks at its set and at the graph only --,
    // so the top of the tree hands its parts to threads (round 5: the analysis is part of the reference's time_solve;
    // 1.0 of the 1.8 s of a 235 k-unknown analysis was this loop).  Every thread works in a NestedDissection of its
    // own (private marks and queues); the nodes are then numbered exactly as the sequential loop below numbers them --
    // a node when it is popped, the parts of a set in the reverse of the order they were pushed --, so the tree, the
    // elimination order and with them every bit of the factorisation are those of one thread
    // (tests/test_direct_solver.py).  SANM_MF_ND_THREADS: the thread budget (1: sequential).
    static std::vector<NdNode> dissect(const SvGraph& g, std::vector<int32_t> set, int budget) {
        NestedDissection nd{g};
        nd.in_set.assign(g.nsv, -1);
        {
            const int32_t mark = nd.next_set++;
            for (int32_t u : set) nd.in_set[u] = mark;
        }
        if (budget <= 1 || (int)set.size() < 1024) {
            std::vector<std::pair<std::vector<int32_t>, int32_t>> work;
            work.emplace_back(std::move(set), -1);
            nd.drain(work);
            return std::move(nd.nodes);
        }
        std::vector<NdNode> out(1);
        std::vector<int32_t> sep, pa, pb;
        nd.threads = budget;
        nd.bisect(set, sep, pa, pb);
        nd.threads = 1;
        if (pa.empty() || pb.empty()) {
            out[0].vars = std::move(set);
            return out;
        }
        out[0].vars = std::move(sep);
        std::vector<std::pair<std::vector<int32_t>, int32_t>> parts;
        if (set.size() >= 262144) {
            // (the components of the second side on a thread and a dissection object of its own: the sides share no vertex)
            std::vector<std::pair<std::vector<int32_t>, int32_t>> parts_b;
            {
                JoinedThreads jt;
                jt.run([&] {
                    NestedDissection other{g};
                    other.in_set.assign(g.nsv, -1);
                    other.split_components(pb, 0, parts_b);
                });
                nd.split_components(pa, 0, parts);
            }
            for (auto& pr : parts_b) parts.push_back(std::move(pr));
        } else {
            nd.split_components(pa, 0, parts);
            nd.split_components(pb, 0, parts);
        }
        // the sequential loop pops the part pushed last first
        std::reverse(parts.begin(), parts.end());
        size_t total = 0;
        for (const auto& p : parts) total += p.first.size();
        std::vector<std::future<std::vector<NdNode>>> subs;
        for (auto& p : parts) {
            const int share = std::max<int>(1, (int)std::lround((double)budget * p.first.size() / std::max<size_t>(total, 1)));
            subs.push_back(std::async(std::launch::async, &NestedDissection::dissect, std::cref(g), std::move(p.first), share));
        }
        for (auto& f : subs) {
            std::vector<NdNode> sub = f.get();
            const int32_t off = (int32_t)out.size();
            out[0].children.push_back(off);
            for (auto& nd2 : sub) {
                nd2.parent = nd2.parent < 0 ? 0 : nd2.parent + off;
                for (auto& c : nd2.children) c += off;
                out.push_back(std::move(nd2));
            }
        }
        return out;
    }

    void run() {
        in_set.assign(g.nsv, -1);
        // connected components of the whole graph are independent roots
        std::vector<int32_t> all(g.nsv);
        std::iota(all.begin(), all.end(), 0);
        std::vector<std::pair<std::vector<int32_t>, int32_t>> work;  // (connected set, parent)
        split_components(all, -1, work);
        const char* env_thr = std::getenv("SANM_MF_ND_THREADS");
        const int budget = env_thr ? std::atoi(env_thr) : host_thread_cap();
        if (budget > 1 && g.nsv >= 8192) {
            // (the roots in the order the loop below pops them)
            for (size_t w = work.size(); w-- > 0;) {
                std::vector<NdNode> sub = dissect(g, std::move(work[w].first), budget);
                const int32_t off = (int32_t)nodes.size();
                for (auto& nd2 : sub) {
                    if (nd2.parent >= 0) nd2.parent += off;
                    for (auto& c : nd2.children) c += off;
                    nodes.push_back(std::move(nd2));
                }
            }
            return;
        }
        drain(work);
    }

    void drain(std::vector<std::pair<std::vector<int32_t>, int32_t>>& work) {
        while (!work.empty()) {
            auto [set, parent] = std::move(work.back());
            work.pop_back();
            int32_t id = nodes.size();
            nodes.emplace_back();
            nodes[id].parent = parent;
            if (parent >= 0) nodes[parent].children.push_back(id);
            if ((int)set.size() <= LEAF) {
                nodes[id].vars = std::move(set);
                continue;
            }
            std::vector<int32_t> sep, pa, pb;
            bisect(set, sep, pa, pb);
            if (pa.empty() || pb.empty()) {
                nodes[id].vars = std::move(set);
                continue;
            }
            nodes[id].vars = std::move(sep);
            split_components(pa, id, work);
            split_components(pb, id, work);
        }
    }

    void split_components(const std::vector<int32_t>& set, int32_t parent,
                          std::vector<std::pair<std::vector<int32_t>, int32_t>>& out) {
        int32_t mark = next_set++;
        for (int32_t u : set) in_set[u] = mark;
        // every found component is re-marked with a fresh id, so "still marked
        // with `mark`" means "not visited yet"
        for (int32_t u : set) {
            if (in_set[u] != mark) continue;
            bfs(u, mark);
            std::vector<int32_t> comp(queue.begin(), queue.end());
            int32_t cm = next_set++;
            for (int32_t v : comp) in_set[v] = cm;
            out.emplace_back(std::move(comp), parent);
        }
    }

    // Vertex separator of a connected set.  A geometric (or graph-distance) cut gives an EDGE separator; its
    // quality decides the size of the front and, at the top of the tree, the length of the panel chain of
    // the factorisation.  Three cheap steps bring the fill of the armadillo Jacobian from 2.6x to about
    // 1.5x that of PARDISO's multilevel ordering:
    //  1. several cuts are tried -- along each principal direction of the point cloud (or the graph-distance
    //     key) at a few positions around the median -- and the one with the lightest boundary is kept;
    //  2. the vertex separator is a MINIMUM VERTEX COVER of the cut edges (Koenig: from a maximum matching of
    //     the bipartite boundary graph), not one whole boundary layer;
    //  3. separator vertices left without a neighbour on one side are handed to the other side, and a few
    //     Fiduccia-Mattheyses passes (refine_separator) move the rest where that lightens the separator.
    void bisect(const std::vector<int32_t>& set, std::vector<int32_t>& sep, std::vector<int32_t>& pa,
                std::vector<int32_t>& pb) {
        const int32_t mark = in_set[set[0]];
        const size_t ns = set.size();
        std::vector<std::vector<double>> keys;  // candidate orderings
        std::vector<double> key2(ns, 0.0);
        if (!g.xyz.empty()) {
            // principal directions of the point cloud (Jacobi eigen-decomposition of the 3x3 covariance)
            double mean[3] = {0, 0, 0};
            for (int32_t u : set)
                for (int d = 0; d < 3; ++d) mean[d] += g.xyz[u * 3 + d];
            for (double& v : mean) v /= ns;
            double C[3][3] = {{0}};
            for (int32_t u : set) {
                double c[3];
                for (int d = 0; d < 3; ++d) c[d] = g.xyz[u * 3 + d] - mean[d];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) C[a][b] += c[a] * c[b];
            }
            const double var_axis[3] = {C[0][0], C[1][1], C[2][2]};  // (C is diagonalised in place below)
            double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
            for (int sweep = 0; sweep < 30; ++sweep)
                for (int p = 0; p < 2; ++p)
                    for (int q = p + 1; q < 3; ++q) {
                        if (std::fabs(C[p][q]) < 1e-300) continue;
                        const double th = 0.5 * std::atan2(2 * C[p][q], C[q][q] - C[p][p]);
                        const double cs = std::cos(th), sn = std::sin(th);
                        for (int k = 0; k < 3; ++k) {  // C <- C J
                            const double a = C[k][p], b = C[k][q];
                            C[k][p] = cs * a - sn * b;
                            C[k][q] = sn * a + cs * b;
                        }
                        for (int k = 0; k < 3; ++k) {  // C <- J' C
                            const double a = C[p][k], b = C[q][k];
                            C[p][k] = cs * a - sn * b;
                            C[q][k] = sn * a + cs * b;
                        }
                        for (int k = 0; k < 3; ++k) {
                            const double a = V[k][p], b = V[k][q];
                            V[k][p] = cs * a - sn * b;
                            V[k][q] = sn * a + cs * b;
                        }
                    }
            int dirs[3] = {0, 1, 2};
            std::sort(dirs, dirs + 3, [&](int a, int b) { return C[a][a] > C[b][b]; });
            // cutting across a direction along which the cloud barely extends makes no sense
            for (int t = 0; t < 3; ++t) {
                const int d = dirs[t];
                if (t > 0 && !(C[d][d] > 0.05 * C[dirs[0]][dirs[0]])) break;
                std::vector<double> key(ns);
                over_pieces(ns, 1, [&](size_t i0, size_t i1, int) {
                    for (size_t i = i0; i < i1; ++i) {
                        const int32_t u = set[i];
                        key[i] = (g.xyz[u * 3] - mean[0]) * V[0][d] + (g.xyz[u * 3 + 1] - mean[1]) * V[1][d] +
                                 (g.xyz[u * 3 + 2] - mean[2]) * V[2][d];
                    }
                });
                keys.push_back(std::move(key));
            }
            // A cloud with two (nearly) equal extents -- the halves of a cube, a plate -- has no principal directions
            // in that plane: the eigenvectors come out at whatever angle the rounding of the covariance gives, and a
            // DIAGONAL cut through a 60 x 60 x 30 half of a 60^3-vertex block has 3100 vertices where the cut across
            // one of the long edges has 1800 (block:60: 35.3 TFLOP per factorisation where the n^6 law from block:48 /
            // block:56 gives 19).  The coordinate axes join the candidates -- in the trees of big problems (30 k+
            // supervariables: there a cut decides teraflops), for sets of 1000+: every BASELINE mesh (at most 25.7 k
            // supervariables) keeps the candidates, hence the ordering and the bits, it had.  Ties go to the principal
            // directions (first in the list).  SANM_MF_AXIS_CUTS_MIN: the problem size from which (0: never),
            // SANM_MF_AXIS_CUTS_SET: the set size.
            static const int64_t axis_min = std::getenv("SANM_MF_AXIS_CUTS_MIN") ? std::atoll(std::getenv("SANM_MF_AXIS_CUTS_MIN")) : 30000;
            static const int64_t axis_set = std::getenv("SANM_MF_AXIS_CUTS_SET") ? std::atoll(std::getenv("SANM_MF_AXIS_CUTS_SET")) : 1000;
            if (axis_min > 0 && (int64_t)g.nsv >= axis_min && (int64_t)ns >= axis_set) {
                const double vmax = std::max(var_axis[0], std::max(var_axis[1], var_axis[2]));
                for (int d = 0; d < 3; ++d) {
                    if (!(var_axis[d] > 0.05 * vmax)) continue;
                    std::vector<double> key(ns);
                    over_pieces(ns, 1, [&](size_t i0, size_t i1, int) {
                        for (size_t i = i0; i < i1; ++i) key[i] = g.xyz[set[i] * 3 + d];
                    });
                    keys.push_back(std::move(key));
                }
            }
        } else {
            // two far-apart sources s, t; key = d(s,.) - d(t,.)
            int32_t s = bfs(set[0], mark);
            int32_t t = bfs(s, mark);
            std::vector<int32_t> ds(ns);
            for (size_t i = 0; i < ns; ++i) ds[i] = dist[set[i]];
            bfs(t, mark);
            std::vector<double> key(ns);
            for (size_t i = 0; i < ns; ++i) {
                key[i] = ds[i] - dist[set[i]];
                key2[i] = ds[i];
            }
            keys.push_back(std::move(key));
        }
        const int32_t markA = next_set++, markB = next_set++;
        // positions: the median and 5 % to either side.  Wider ranges (40..60 %, 30..70 %) find slightly
        // lighter separators but deepen the tree by a level, and every level costs two launches per solve
        // (measured on the three BASELINE meshes: 45..55 % is best or within 2 % of best)
        static const double fracs[] = {0.5, 0.45, 0.55};
        const int nf = ns < 200 ? 1 : 3;  // small sets: the median only
        const int nk = (int)keys.size();
        // The candidates read the graph, the set and its position map only (each writes a side array of its own):
        // the cuts near the root -- up to 6 orderings x 3 positions over the whole graph, 0.09 of the 0.17 s of
        // a 78 k-supervariable dissection -- are sorted and weighed by the threads the subtrees below will get.
        for (size_t i = 0; i < ns; ++i) dist[set[i]] = (int32_t)i;
        std::vector<std::vector<int32_t>> ords(nk);
        std::vector<int64_t> weight((size_t)nk * nf, 0);
        auto cut_of = [&](int f) { return std::min(ns - 1, std::max<size_t>(1, (size_t)(fracs[f] * ns))); };
        // A candidate needs the FIRST `cut` members in the order of its key, not the order itself: the members are
        // partitioned at the (up to three) cut positions -- the order is strict and total (ties go to the smaller
        // supervariable), so the parts are those of a full sort --, and one walk over the adjacency of the set weighs
        // the boundary layers of all the positions of a key: a member of part c with neighbours in parts minc..maxc
        // touches the other side of the cut after part j when c <= j < maxc or minc <= j < c.  (Round 6: sorting and
        // three walks per key were 0.5 of the 0.75 s of host work of a 78 k-supervariable dissection.)
        size_t pos_sorted[3];  // the cut positions, ascending
        int which[3];          // fracs[f] -> index into pos_sorted
        int ncut = 0;
        for (int f = 0; f < nf; ++f) pos_sorted[ncut++] = cut_of(f);
        std::sort(pos_sorted, pos_sorted + ncut);
        ncut = (int)(std::unique(pos_sorted, pos_sorted + ncut) - pos_sorted);
        for (int f = 0; f < nf; ++f) which[f] = (int)(std::find(pos_sorted, pos_sorted + ncut, cut_of(f)) - pos_sorted);
        struct Cand {
            double k, k2;
            int32_t id, idx;
        };
        auto weigh_key = [&](int k) {
            const auto& key = keys[k];
            std::vector<Cand> c(ns);
            for (size_t i = 0; i < ns; ++i) c[i] = {key[i], key2[i], set[i], (int32_t)i};
            auto less = [](const Cand& a, const Cand& b) {
                if (a.k != b.k) return a.k < b.k;
                if (a.k2 != b.k2) return a.k2 < b.k2;
                return a.id < b.id;
            };
            size_t from = 0;
            for (int j = 0; j < ncut; ++j) {
                std::nth_element(c.begin() + from, c.begin() + pos_sorted[j], c.end(), less);
                from = pos_sorted[j];
            }
            std::vector<int32_t>& ord = ords[k];
            ord.resize(ns);
            std::vector<int8_t> part(ns);
            {
                int j = 0;
                for (size_t i = 0; i < ns; ++i) {
                    while (j < ncut && i >= pos_sorted[j]) ++j;
                    ord[i] = c[i].idx;
                    part[c[i].idx] = (int8_t)j;
                }
            }
            int64_t wA[3] = {0, 0, 0}, wB[3] = {0, 0, 0};
            for (size_t i = 0; i < ns; ++i) {
                const int32_t u = set[i];
                int minc = 4, maxc = -1;
                for (int32_t q = g.adj_ptr[u]; q < g.adj_ptr[u + 1]; ++q) {
                    const int32_t v = g.adj[q];
                    if (in_set[v] != mark) continue;
                    const int cv = part[dist[v]];
                    minc = std::min(minc, cv);
                    maxc = std::max(maxc, cv);
                }
                const int cu = part[i];
                for (int j = 0; j < ncut; ++j) {
                    if (cu <= j) {
                        if (maxc > j) wA[j] += g.size(u);
                    } else if (minc <= j) {
                        wB[j] += g.size(u);
                    }
                }
            }
            for (int f = 0; f < nf; ++f) weight[(size_t)k * nf + f] = std::min(wA[which[f]], wB[which[f]]);
        };
        auto spread = [&](int count, auto&& fn) {
            const int nt = ns >= 2048 ? std::min(threads, count) : 1;
            if (nt <= 1) {
                for (int i = 0; i < count; ++i) fn(i);
                return;
            }
            JoinedThreads jt;
            for (int t = 1; t < nt; ++t)
                jt.run([&, t] { for (int i = t; i < count; i += nt) fn(i); });
            for (int i = 0; i < count; i += nt) fn(i);
        };
        spread(nk, weigh_key);
        int best_k = 0;
        size_t best_cut = 0;
        double best_score = 1e300;
        for (int k = 0; k < nk; ++k)
            for (int f = 0; f < nf; ++f) {
                // an unbalanced cut must pay for itself: the larger part is dissected one level deeper
                const double score = (double)weight[k * nf + f] * (1.0 + 2.0 * std::fabs(fracs[f] - 0.5));
                if (score < best_score) {
                    best_score = score;
                    best_cut = cut_of(f);
                    best_k = k;
                }
            }
        const std::vector<int32_t>& best_ord = ords[best_k];
        over_pieces(ns, 1, [&](size_t i0, size_t i1, int) {
            for (size_t i = i0; i < i1; ++i) in_set[set[best_ord[i]]] = i < best_cut ? markA : markB;
        });

        // boundary layers and the bipartite graph of the cut edges (found by pieces of the set, joined in their order)
        std::vector<int32_t> bA, bB;
        {
            std::vector<std::vector<int32_t>> pA(16), pB(16);
            over_pieces(ns, 1, [&](size_t i0, size_t i1, int t) {
                for (size_t i = i0; i < i1; ++i) {
                    const int32_t u = set[i];
                    const int32_t other = in_set[u] == markA ? markB : markA;
                    bool touch = false;
                    for (int32_t q = g.adj_ptr[u]; q < g.adj_ptr[u + 1] && !touch; ++q) touch = in_set[g.adj[q]] == other;
                    if (touch) (in_set[u] == markA ? pA : pB)[t].push_back(u);
                }
            });
            for (int t = 0; t < 16; ++t) {
                bA.insert(bA.end(), pA[t].begin(), pA[t].end());
                bB.insert(bB.end(), pB[t].begin(), pB[t].end());
            }
        }
        // maximum matching by augmenting paths (boundary layers have hundreds of vertices)
        std::vector<int32_t>& loc = dist;  // position of a boundary vertex in bA / bB
        for (size_t i = 0; i < bA.size(); ++i) loc[bA[i]] = i;
        for (size_t i = 0; i < bB.size(); ++i) loc[bB[i]] = i;
        std::vector<int32_t> matchA(bA.size(), -1), matchB(bB.size(), -1), seen(bB.size(), -1);
        std::vector<std::pair<int32_t, int32_t>> stack;  // (a, next adjacency position)
        std::vector<int32_t> path;
        for (int32_t root = 0; root < (int32_t)bA.size(); ++root) {
            // iterative DFS for an augmenting path from `root`
            stack.clear();
            stack.emplace_back(root, g.adj_ptr[bA[root]]);
            std::vector<int32_t> via(bA.size(), -1);  // b through which a was entered
            bool found = false;
            int32_t endb = -1;
            while (!stack.empty() && !found) {
                auto& [a, pos] = stack.back();
                if (pos >= g.adj_ptr[bA[a] + 1]) {
                    stack.pop_back();
                    continue;
                }
                const int32_t v = g.adj[pos++];
                if (in_set[v] != markB) continue;
                const int32_t b = loc[v];
                if (seen[b] == root) continue;
                seen[b] = root;
                if (matchB[b] < 0) {
                    found = true;
                    endb = b;
                } else {
                    const int32_t a2 = matchB[b];
                    via[a2] = b;
                    stack.emplace_back(a2, g.adj_ptr[bA[a2]]);
                }
            }
            if (found) {
                // flip along the stack
                int32_t b = endb;
                for (int32_t i = (int32_t)stack.size() - 1; i >= 0; --i) {
                    const int32_t a = stack[i].first;
                    const int32_t prev = matchA[a];
                    matchA[a] = b;
                    matchB[b] = a;
                    b = prev;
                }
            }
        }
        // Koenig: Z = vertices reachable from unmatched A vertices by alternating paths;
        // cover = (A \ Z) + (B & Z)
        std::vector<char> zA(bA.size(), 0), zB(bB.size(), 0);
        std::vector<int32_t> todo;
        for (size_t a = 0; a < bA.size(); ++a)
            if (matchA[a] < 0) {
                zA[a] = 1;
                todo.push_back(a);
            }
        while (!todo.empty()) {
            const int32_t a = todo.back();
            todo.pop_back();
            for (int32_t q = g.adj_ptr[bA[a]]; q < g.adj_ptr[bA[a] + 1]; ++q) {
                const int32_t v = g.adj[q];
                if (in_set[v] != markB) continue;
                const int32_t b = loc[v];
                if (zB[b] || matchA[a] == b) continue;
                zB[b] = 1;
                const int32_t a2 = matchB[b];
                if (a2 >= 0 && !zA[a2]) {
                    zA[a2] = 1;
                    todo.push_back(a2);
                }
            }
        }
        const int32_t markS = next_set++;
        for (size_t a = 0; a < bA.size(); ++a)
            if (!zA[a]) in_set[bA[a]] = markS;
        for (size_t b = 0; b < bB.size(); ++b)
            if (zB[b]) in_set[bB[b]] = markS;
        // hand back separator vertices that touch only one side
        for (int pass = 0; pass < 2; ++pass)
            for (int32_t u : set) {
                if (in_set[u] != markS) continue;
                bool hasA = false, hasB = false;
                for (int32_t q = g.adj_ptr[u]; q < g.adj_ptr[u + 1]; ++q) {
                    hasA = hasA || in_set[g.adj[q]] == markA;
                    hasB = hasB || in_set[g.adj[q]] == markB;
                }
                if (!hasB) in_set[u] = markA;
                else if (!hasA) in_set[u] = markB;
            }
        refine_separator(set, markA, markB, markS);
        for (int32_t u : set) {
            if (in_set[u] == markA) pa.push_back(u);
            else if (in_set[u] == markB) pb.push_back(u);
            else sep.push_back(u);
        }
    }

    // Fiduccia-Mattheyses passes on the vertex separator.  A move takes a separator vertex v into side X; its
    // neighbours on the other side Y then have to enter the separator, so the move changes the separator
    // weight by w(N(v) & Y) - w(v).  A pass applies the best admissible move again and again -- also when it
    // makes things worse, each vertex at most once -- and finally returns to the lightest separator it has
    // seen; that is how the search leaves the local optimum the vertex cover ends in.  Sides may not exceed
    // 58 % of the set: an unbalanced cut deepens the tree, and a tree level costs more than a few pivots.
    void refine_separator(const std::vector<int32_t>& set, int32_t markA, int32_t markB, int32_t markS) {
        // The decisions are those of the plain form -- every step weighs the unlocked separator vertices in the order
        // of the set and takes the first best move, every pass ends in the lightest state it has seen -- without a walk
        // over the whole set per step (round 6: 0.05 of the 0.07 s of the root cut of a 78 k-supervariable graph, all of
        // it on the critical path of the constructor): the separator is a bitmap over the positions of the set, the
        // weights of every vertex's neighbours on either side are kept up to date by the moves, and the way back to
        // the best state is a log of the changes since.
        const size_t ns = set.size();
        int64_t w[3] = {0, 0, 0};  // A, B, S
        auto side_of = [&](int32_t u) { return in_set[u] == markA ? 0 : (in_set[u] == markB ? 1 : (in_set[u] == markS ? 2 : 3)); };
        {
            int64_t pw[16][3] = {};
            over_pieces(ns, 1, [&](size_t i0, size_t i1, int t) {
                for (size_t i = i0; i < i1; ++i) pw[t][side_of(set[i])] += g.size(set[i]);
            });
            for (int t = 0; t < 16; ++t)
                for (int d = 0; d < 3; ++d) w[d] += pw[t][d];
        }
        const int64_t wTot = w[0] + w[1] + w[2];
        const int64_t wMax = std::max<int64_t>((int64_t)(0.58 * wTot), std::max(w[0], w[1]));
        std::vector<int32_t>& locked = stamp;  // stamp[v] == cur_stamp: moved in this pass
        std::vector<int32_t>& pos = dist;      // position of a member in the set
        std::vector<int32_t> nA(ns, 0), nB(ns, 0);
        std::vector<uint64_t> sep_bits((ns + 63) / 64, 0);
        over_pieces(ns, 64, [&](size_t i0, size_t i1, int) {
            for (size_t i = i0; i < i1; ++i) pos[set[i]] = (int32_t)i;
        });
        over_pieces(ns, 64, [&](size_t i0, size_t i1, int) {  // (pieces of whole bitmap words)
            for (size_t i = i0; i < i1; ++i) {
                const int32_t u = set[i];
                if (in_set[u] == markS) sep_bits[i >> 6] |= 1ull << (i & 63);
                for (int32_t q = g.adj_ptr[u]; q < g.adj_ptr[u + 1]; ++q) {
                    const int32_t v = g.adj[q];
                    if (in_set[v] == markA) nA[i] += g.size(v);
                    else if (in_set[v] == markB) nB[i] += g.size(v);
                }
            }
        });
        std::vector<std::pair<int32_t, int32_t>> undo;  // (vertex, mark it had) since the best state of the pass
        auto set_state = [&](int32_t u, int32_t to) {
            const int32_t from = in_set[u];
            const int32_t wu = g.size(u);
            const int32_t dA = (to == markA ? wu : 0) - (from == markA ? wu : 0);
            const int32_t dB = (to == markB ? wu : 0) - (from == markB ? wu : 0);
            for (int32_t q = g.adj_ptr[u]; q < g.adj_ptr[u + 1]; ++q) {
                const int32_t v = g.adj[q];
                const int32_t mv = in_set[v];
                if (mv != markA && mv != markB && mv != markS) continue;
                nA[pos[v]] += dA;
                nB[pos[v]] += dB;
            }
            const size_t i = pos[u];
            if (to == markS) sep_bits[i >> 6] |= 1ull << (i & 63);
            else if (from == markS) sep_bits[i >> 6] &= ~(1ull << (i & 63));
            in_set[u] = to;
        };
        for (int pass = 0; pass < 6; ++pass) {
            ++cur_stamp;
            int64_t best_w = w[2], cur_w = w[2];
            int64_t best_imb = std::llabs(w[0] - w[1]);
            undo.clear();
            int64_t wa = w[0], wb = w[1];
            int64_t best_wa = wa, best_wb = wb;  // (the sides' weights in the best state: exact integer bookkeeping)
            int since_best = 0;
            for (int step = 0; step < (int)ns && since_best < 60; ++step) {
                // best admissible move
                int32_t bv = -1, bside = 0;
                int64_t bgain = INT64_MIN;
                for (size_t wd = 0; wd < sep_bits.size(); ++wd)
                    for (uint64_t bits = sep_bits[wd]; bits; bits &= bits - 1) {
                        const size_t i = wd * 64 + (size_t)__builtin_ctzll(bits);
                        const int32_t v = set[i];
                        if (locked[v] == cur_stamp) continue;
                        const int64_t wv = g.size(v);
                        // to A: B-neighbours enter the separator (B shrinks, A grows by w(v))
                        if (wa + wv <= wMax) {
                            const int64_t gain = wv - nB[i];
                            if (gain > bgain || (gain == bgain && wa < wb)) {
                                bgain = gain;
                                bv = v;
                                bside = 0;
                            }
                        }
                        if (wb + wv <= wMax) {
                            const int64_t gain = wv - nA[i];
                            if (gain > bgain || (gain == bgain && bside == 0 && wb < wa)) {
                                bgain = gain;
                                bv = v;
                                bside = 1;
                            }
                        }
                    }
                if (bv < 0) break;
                const int32_t to = bside == 0 ? markA : markB, from = bside == 0 ? markB : markA;
                for (int32_t q = g.adj_ptr[bv]; q < g.adj_ptr[bv + 1]; ++q) {
                    const int32_t u = g.adj[q];
                    if (in_set[u] == from) {
                        undo.emplace_back(u, from);
                        set_state(u, markS);
                        (bside == 0 ? wb : wa) -= g.size(u);
                    }
                }
                undo.emplace_back(bv, markS);
                set_state(bv, to);
                locked[bv] = cur_stamp;
                (bside == 0 ? wa : wb) += g.size(bv);
                cur_w -= bgain;
                const int64_t imb = std::llabs(wa - wb);
                if (cur_w < best_w || (cur_w == best_w && imb < best_imb)) {
                    best_w = cur_w;
                    best_imb = imb;
                    best_wa = wa;
                    best_wb = wb;
                    undo.clear();
                    since_best = 0;
                } else {
                    ++since_best;
                }
            }
            for (size_t i = undo.size(); i-- > 0;) set_state(undo[i].first, undo[i].second);
            const int64_t before = w[2];
            w[0] = best_wa, w[1] = best_wb, w[2] = best_w;  // (what a count over the set would find)
            if (w[2] >= before) break;
        }
    }
};

// ---------------------------------------------------------------------------
// Amalgamation of tree levels.  Every level of the dissection tree costs the solve two launches per right-hand
// side (an ANM step solves `order` of them, one after the other) and the factorisation its fixed launches
// (extend-add rounds, first diagonal tile, panel finalisation, two GEMM passes), while the fronts of the lower
// levels are so small that their kernels are pure launch latency.  Merging the fronts of a level into their
// parents removes that level: the parent's pivot block takes the children's variables in front of its own (same
// elimination order as before), the grandchildren become its children, its boundary stays what it was.  The
// price is dense storage of the (zero) coupling between the merged children and of their rows over the whole
// parent boundary instead of their own -- a few per cent of the factor for the two merges below on a 3D mesh,
// against 1/5 of the solve launches.  SANM_MF_MERGE="h1,h2,..." overrides the heights whose fronts are merged
// into their parents ("" or "none": no merging); the default merges heights 1 and 3 when the tree is tall enough
// that those are small separator levels (measured on the BASELINE meshes: DESIGN.md section 5).
void amalgamate_levels(std::vector<NdNode>& nodes, const SvGraph& g) {
    const int32_t F = nodes.size();
    std::vector<int32_t> height(F, 0);
    // children appear after their parents in `nodes` (the dissection pushes a node before its parts)
    for (int32_t u = F - 1; u >= 0; --u)
        if (nodes[u].parent >= 0) height[nodes[u].parent] = std::max(height[nodes[u].parent], height[u] + 1);
    int32_t H = 0;
    for (int32_t u = 0; u < F; ++u) H = std::max(H, height[u] + 1);
    std::vector<int> merge;
    if (const char* e = std::getenv("SANM_MF_MERGE")) {
        for (const char* p = e; *p;) {
            if (*p >= '0' && *p <= '9') {
                merge.push_back(std::atoi(p));
                while (*p >= '0' && *p <= '9') ++p;
            } else {
                ++p;
            }
        }
    } else if (H >= 8 && g.nsv <= 50000) {
        // (small systems only: there a level is pure launch latency; on a 24^3-vertex block the two merges cost 7 %
        // more factor entries, on larger ones the factorisation is bound by arithmetic and they would only cost)
        merge = {1, 3};
    }
    if (merge.empty()) return;
    std::vector<char> at(H + 1, 0);
    for (int h : merge)
        if (h >= 1 && h < H - 1) at[h] = 1;  // never the leaves (they hold the bulk of the factor) nor the roots
    std::vector<char> dead(F, 0);
    // top-down over the nodes (parents first): a merged node hands its variables and children up
    for (int32_t u = 0; u < F; ++u) {
        const int32_t p = nodes[u].parent;
        if (p < 0 || !at[height[u]] || height[p] != height[u] + 1) continue;
        dead[u] = 1;
    }
    for (int32_t u = F - 1; u >= 0; --u) {  // children before parents: a dead node is complete when it is merged
        if (!dead[u]) continue;
        NdNode& c = nodes[u];
        int32_t p = c.parent;
        while (dead[p]) p = nodes[p].parent;  // (adjacent heights are never both merged below; kept general)
        NdNode& P = nodes[p];
        std::vector<int32_t> vars = std::move(c.vars);
        vars.insert(vars.end(), P.vars.begin(), P.vars.end());
        P.vars = std::move(vars);
        for (int32_t gc : c.children) {
            nodes[gc].parent = p;
            P.children.push_back(gc);
        }
        P.children.erase(std::remove(P.children.begin(), P.children.end(), u), P.children.end());
        c.children.clear();
    }
    // compact
    std::vector<int32_t> newid(F, -1);
    std::vector<NdNode> out;
    for (int32_t u = 0; u < F; ++u)
        if (!dead[u]) {
            newid[u] = out.size();
            out.push_back(std::move(nodes[u]));
        }
    for (auto& nd : out) {
        if (nd.parent >= 0) nd.parent = newid[nd.parent];
        for (auto& ch : nd.children) ch = newid[ch];
    }
    nodes.swap(out);
}

// ---------------------------------------------------------------------------
// Chains in place of big fronts.  A front of k pivots and b boundary rows pays, beside its LU (2/3 k^3 + 2 k^2 b +
// 2 k b^2), 2 k^3 + 2 k^2 b for the explicit inverses of its pivot block and the boundary blocks of the solve
// operators (mf_kernels.h: that is what makes every solve level one mat-vec launch) -- a third of all factor flops on a
// 48^3 block, nearly all of it in the dozen separators at the top.  Cutting the pivots of such a front into chunks
// of about W, each a front of its own whose boundary is the rest of the pivots plus the old boundary, leaves the
// elimination order, the permutation and the factor entries what they were and turns the inverse work into W k^2 + 2 W k b:
// the rest of the front's elimination becomes Schur products with K = W (the fast tall tiles).  The price is one
// more level per chunk (two launches per solve each, 5 us where a level of this size streams for 15-90) and the
// chunk's Schur complement handed on through an extend-add.  SANM_MF_SPLIT_K = W: fixed width (0: never cut).
bool split_big_fronts(std::vector<NdNode>& nodes, const SvGraph& g, const std::vector<int32_t>& node_k) {
    // chunk width W: fronts beyond 1.5 W pivots are cut into ceil(k / W) chunks.  One width for the whole tree (fronts
    // of a level are batched into one launch each: chunks of one size keep the levels homogeneous; a width per front
    // from a cost model -- c chunks leave (k^3 + 2 k^2 b) / c of the inverse work and cost c extend-adds of about
    // (b + k/2)^2 entries, smallest sum at 12.4 sqrt(b + k/2) pivots by the rates of one MI355X -- measured 1-4 %
    // slower on the 48^3 block for that reason).  The width that measured best grows with the tree: 896-1088 on the 48^3
    // block (largest front 6756 pivots; 1280+: 4 % slower), 1536-2048 on the 60^3 block (1024: 4 % slower), i.e.
    // about 1/6.75 of the largest front, kept inside [1024, 1536]: no front of fewer than 1537 pivots is ever cut, so
    // that every BASELINE mesh (fronts of <= 1032 rows: a level there is launch latency, not arithmetic) stays as it
    // was, bit for bit.  SANM_MF_SPLIT_K = W fixes the width (tests force the path on small fronts); 0: never cut.
    int W = -1;
    if (const char* e = std::getenv("SANM_MF_SPLIT_K")) W = std::atoi(e);
    if (W == 0) return false;
    const int32_t F = nodes.size();
    if (W < 0) {
        int32_t kmax = 0;
        for (int32_t u = 0; u < F; ++u) kmax = std::max(kmax, node_k[u]);
        W = std::min(1536, std::max(1024, (int)std::lround(kmax / 6.75 / 32) * 32));
    }
    bool any = false;
    for (int32_t u = 0; u < F; ++u) {
        const int64_t piv = node_k[u];
        const int nchunk = piv > W + W / 2 ? (int)((piv + W - 1) / W) : 1;
        if (nchunk < 2) continue;
        any = true;
        const int64_t target = (piv + nchunk - 1) / nchunk;
        // consecutive ranges of the front's variables; the node itself keeps the last one (and its place under its parent)
        std::vector<int32_t> vars = std::move(nodes[u].vars);
        std::vector<int32_t> below = std::move(nodes[u].children);
        size_t pos = 0;
        int32_t prev = -1;
        for (int c = 0; c < nchunk; ++c) {
            std::vector<int32_t> mine;
            int64_t got = 0;
            while (pos < vars.size() && (c == nchunk - 1 || got < target)) {
                got += g.size(vars[pos]);
                mine.push_back(vars[pos++]);
            }
            if (mine.empty()) continue;
            const bool last = pos == vars.size();
            int32_t id = u;
            if (!last) {
                id = nodes.size();
                nodes.emplace_back();
            }
            NdNode& nd = nodes[id];
            nd.vars = std::move(mine);
            nd.children.clear();
            if (prev < 0) {
                nd.children = below;
                for (int32_t ch : below) nodes[ch].parent = id;
            } else {
                nd.children.push_back(prev);
                nodes[prev].parent = id;
            }
            prev = id;
            if (last) break;
        }
    }
    return any;
}

}  // namespace

// ---------------------------------------------------------------------------
template <class T>
T* Multifrontal::upload(const std::vector<T>& v) {  // (on the spot: the merged top block, never deferred)
    void* p = m_be->alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
    if (!v.empty()) m_be->h2d(p, v.data(), v.size() * sizeof(T));
    m_bufs.push_back(p);
    return static_cast<T*>(p);
}
void Multifrontal::run_op(DeviceOp& op) {
    void* p = nullptr;
    if (op.detached && m_front_job.job.valid()) {
        p = m_front_job.job.get();
        if (p) m_be->adopt(p, op.bytes);
    }
    if (!p) p = m_be->alloc(std::max<size_t>(op.bytes, 1));
    if (op.src && op.bytes) m_be->h2d(p, op.src, op.bytes);
    if (op.zero) m_be->zero(p, op.bytes);
    m_bufs.push_back(p);
    op.set(p);
}
template <class P, class T>
void Multifrontal::upload_to(P& target, std::vector<T>&& v) {
    DeviceOp op;
    op.bytes = v.size() * sizeof(T);
    op.set = [&target](void* p) { target = static_cast<P>(p); };
    if (m_defer) {
        auto keep = std::make_shared<std::vector<T>>(std::move(v));
        op.src = keep->data();
        op.keep = keep;
        m_pending.push_back(std::move(op));
    } else {
        op.src = v.data();
        run_op(op);
    }
}
template <class P, class T>
void Multifrontal::upload_to(P& target, const std::vector<T>& v) {
    if (m_defer) {
        upload_to(target, std::vector<T>(v));
        return;
    }
    DeviceOp op;
    op.bytes = v.size() * sizeof(T);
    op.src = v.data();
    op.set = [&target](void* p) { target = static_cast<P>(p); };
    run_op(op);
}
template <class P, class T>
void Multifrontal::upload_kept(P& target, std::shared_ptr<void> keep, const T* src, size_t count) {
    DeviceOp op;
    op.bytes = count * sizeof(T);
    op.src = src;
    op.set = [&target](void* p) { target = static_cast<P>(p); };
    if (m_defer) {
        op.keep = std::move(keep);
        m_pending.push_back(std::move(op));
    } else {
        run_op(op);
    }
}
template <class P>
void Multifrontal::alloc_to(P& target, size_t bytes, bool zero) {
    DeviceOp op;
    op.bytes = bytes;
    op.zero = zero;
    op.set = [&target](void* p) { target = static_cast<P>(p); };
    if (m_defer) m_pending.push_back(std::move(op));
    else run_op(op);
}
void Multifrontal::finish_device() {
    for (DeviceOp& op : m_pending) run_op(op);
    m_pending.clear();
    m_pending.shrink_to_fit();
    m_defer = false;
}

Multifrontal::~Multifrontal() {
    m_be->forget_chains(m_dev.front_store);  // replayed launch chains captured on this solver's buffers
    for (void* p : m_bufs) m_be->free(p);
}

Multifrontal::Multifrontal(Backend* be, int64_t n, const std::vector<uint32_t>& rowptr,
                           const std::vector<uint32_t>& col, const double* coords, int rank, int world,
                           bool defer_device)
        : m_be{be},
          m_defer{defer_device} {
    sanm_check(n > 0 && (int64_t)rowptr.size() == n + 1, "bad CSR pattern");
    analyse(n, &rowptr, &col, nullptr, coords, rank, world);
}

Multifrontal::Multifrontal(Backend* be, int64_t n, const BlockPattern& blocks, const double* coords, int rank, int world,
                           bool defer_device)
        : m_be{be},
          m_defer{defer_device} {
    sanm_check(blocks.block >= 1 && n > 0 && n % blocks.block == 0 && blocks.qptr && blocks.qcol &&
                       (int64_t)blocks.qptr->size() == n / blocks.block + 1 && blocks.qptr->back() == blocks.qcol->size(),
               "bad block pattern");
    analyse(n, nullptr, nullptr, &blocks, coords, rank, world);
}

void Multifrontal::analyse(int64_t n, const std::vector<uint32_t>* rowptr_p, const std::vector<uint32_t>* col_p,
                           const BlockPattern* blocks, const double* coords, int rank, int world) {
    Backend* const be = m_be;
    const bool defer_device = m_defer;
    (void)be;
    const auto t_ctor = std::chrono::steady_clock::now();
    // (the merged top block multiplies device blocks out while it is built: it cannot be deferred, and a constructor
    // that touches the backend must run on the backend's owner thread -- the caller's job, anm.cpp: `beside`)
    sanm_check(!(defer_device && std::getenv("SANM_MF_TOP") && std::atoi(std::getenv("SANM_MF_TOP")) > 0),
               "multifrontal: SANM_MF_TOP needs a constructor that is not deferred (owner thread of the backend)");
    sanm_check(world >= 1 && rank >= 0 && rank < world, "multifrontal: rank %d of %d", rank, world);
    sanm_check(n < INT32_MAX / 2, "system too large for 32-bit indices");
    const bool dbg_clock = std::getenv("SANM_MF_DEBUG") != nullptr;
    auto t_clock = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg_clock) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "mf analysis: %-28s %.3f s\n", what, std::chrono::duration<double>(t1 - t_clock).count());
        t_clock = t1;
    };
    SvGraph g = blocks ? build_sv_graph_of_blocks(n, blocks->block, *blocks->qptr, *blocks->qcol, coords)
                       : build_sv_graph(n, *rowptr_p, *col_p, coords);
    lap("supervariable graph");
    nr_supervar = g.nsv;
    used_coords = coords != nullptr;
    NestedDissection nd{g};
    nd.run();
    lap("nested dissection");
    amalgamate_levels(nd.nodes, g);
    int32_t F = 0;
    std::vector<int32_t> post, fid, parent, height, sv_front, sv_start, own_start, kf, perm;
    std::vector<std::vector<int32_t>> children, bnd_sv;
    // numbering of the fronts and of the unknowns, boundaries (run once more after big fronts were cut into chains:
    // the cut needs every front's pivot and boundary counts, and leaves the numbering of the unknowns what it was)
    auto order_tree = [&]() {
    F = nd.nodes.size();

        // postorder: children before parents; front ids follow the postorder
        post.clear();
        post.reserve(F);
        {
            std::vector<std::pair<int32_t, size_t>> st;
            for (int32_t r = 0; r < F; ++r) {
                if (nd.nodes[r].parent >= 0) continue;
                st.emplace_back(r, 0);
                while (!st.empty()) {
                    auto& [u, ci] = st.back();
                    if (ci < nd.nodes[u].children.size()) {
                        int32_t c = nd.nodes[u].children[ci++];
                        st.emplace_back(c, 0);
                    } else {
                        post.push_back(u);
                        st.pop_back();
                    }
                }
            }
        }
        sanm_check((int32_t)post.size() == F, "postorder failed");
        fid.assign(F, 0);  // nd node -> front id
        for (int32_t i = 0; i < F; ++i) fid[post[i]] = i;

        parent.assign(F, -1);
        height.assign(F, 0);
        children.assign(F, {});
        for (int32_t u = 0; u < F; ++u) {
            if (nd.nodes[u].parent >= 0) {
                parent[fid[u]] = fid[nd.nodes[u].parent];
                children[fid[nd.nodes[u].parent]].push_back(fid[u]);
            }
        }
        for (int32_t f = 0; f < F; ++f) {
            std::sort(children[f].begin(), children[f].end());
            if (parent[f] >= 0) height[parent[f]] = std::max(height[parent[f]], height[f] + 1);
        }

        // new numbering of the unknowns; owner front of every supervariable
        sv_front.assign(g.nsv, -1);
        sv_start.assign(g.nsv, 0);
        own_start.assign(F, 0);
        kf.assign(F, 0);
        perm.assign(n, -1);
        {
            int32_t next = 0;
            for (int32_t f = 0; f < F; ++f) {
                own_start[f] = next;
                // (elimination order inside a front: as the dissection left it -- an amalgamated front lists the
                // variables of its former children first, i.e. keeps the order of the tree it came from)
                const auto& vars = nd.nodes[post[f]].vars;
                for (int32_t s : vars) {
                    sanm_check(sv_front[s] < 0, "supervariable assigned twice");
                    sv_front[s] = f;
                    sv_start[s] = next;
                    for (int32_t q = g.sv_ptr[s]; q < g.sv_ptr[s + 1]; ++q) perm[g.sv_members[q]] = next++;
                }
                kf[f] = next - own_start[f];
                sanm_check(kf[f] > 0, "empty front");
            }
            sanm_check(next == n, "ordering does not cover all unknowns");
        }

        // boundaries (in supervariables, then expanded)
        bnd_sv.assign(F, {});
        {
            // height by height (the fronts of one height need their children's lists only), each height's fronts on the
            // host's threads; a thread marks the supervariables a list already holds instead of sorting them out
            int32_t Hh = 0;
            for (int32_t f = 0; f < F; ++f) Hh = std::max(Hh, height[f] + 1);
            std::vector<int32_t> hptr(Hh + 1, 0), by_h(F);
            for (int32_t f = 0; f < F; ++f) hptr[height[f] + 1]++;
            for (int32_t h = 0; h < Hh; ++h) hptr[h + 1] += hptr[h];
            {
                std::vector<int32_t> fill(hptr.begin(), hptr.end() - 1);
                for (int32_t f = 0; f < F; ++f) by_h[fill[height[f]]++] = f;
            }
            std::vector<std::vector<int32_t>> seen(64);
            std::vector<std::string> errs(64);
            for (int32_t h = 0; h < Hh; ++h)
                parallel_ranges(hptr[h + 1] - hptr[h], 32, [&](int64_t i0, int64_t i1, int t) {
                    std::vector<int32_t>& mark = seen[t % 64];
                    if (mark.empty()) mark.assign(g.nsv, -1);
                    for (int64_t i = i0; i < i1; ++i) {
                        const int32_t f = by_h[hptr[h] + i];
                        std::vector<int32_t>& b = bnd_sv[f];
                        for (int32_t s : nd.nodes[post[f]].vars)
                            for (int32_t q = g.adj_ptr[s]; q < g.adj_ptr[s + 1]; ++q) {
                                const int32_t t2 = g.adj[q];
                                if (sv_front[t2] > f && mark[t2] != f) {
                                    mark[t2] = f;
                                    b.push_back(t2);
                                }
                            }
                        for (int32_t c : children[f])
                            for (int32_t t2 : bnd_sv[c])
                                if (sv_front[t2] != f && mark[t2] != f) {
                                    mark[t2] = f;
                                    b.push_back(t2);
                                }
                        // order by new index so that expanded lists are ascending
                        std::sort(b.begin(), b.end(), [&](int32_t a, int32_t c2) { return sv_start[a] < sv_start[c2]; });
                        for (int32_t t2 : b)
                            if (!(sv_front[t2] > f)) errs[t % 64] = "boundary variable is not in an ancestor";
                    }
                });
            for (const auto& e : errs) sanm_check(e.empty(), "%s", e.c_str());
        }

    };
    order_tree();
    {
        std::vector<int32_t> node_k(F);
        for (int32_t u = 0; u < F; ++u) node_k[u] = kf[fid[u]];
        const std::vector<int32_t> perm_before = perm;
        if (split_big_fronts(nd.nodes, g, node_k)) {
            order_tree();
            sanm_check(perm == perm_before, "cutting fronts into chains changed the elimination order");
        }
    }
    nr_front = F;
    lap("tree order, boundaries");
    if (dbg_clock) {  // fingerprint of the ordering: an analysis that got faster must print the same one
        uint64_t h = 1469598103934665603ull;
        for (int64_t i = 0; i < n; ++i) h = (h ^ (uint64_t)(uint32_t)perm[i]) * 1099511628211ull;
        for (int32_t f = 0; f < F; ++f) h = (h ^ (uint64_t)(uint32_t)(kf[f] * 31 + parent[f])) * 1099511628211ull;
        std::fprintf(stderr, "mf analysis: ordering fingerprint %016llx (%d fronts, %d supervariables)\n", (unsigned long long)h, F, g.nsv);
    }

    std::vector<MfFrontDev> fr(F);
    std::vector<double> front_flops(F, 0.0);
    // (bnd_idx, rel and perm are read again after their upload was queued: held by pointers the queue shares)
    auto bnd_idx_keep = std::make_shared<std::vector<int32_t>>();
    std::vector<int32_t>& bnd_idx = *bnd_idx_keep;
    int64_t off = 0;
    for (int32_t f = 0; f < F; ++f) {
        fr[f].bnd_off = bnd_idx.size();
        for (int32_t t : bnd_sv[f])
            for (int32_t q = 0; q < g.size(t); ++q) bnd_idx.push_back(sv_start[t] + q);
        int32_t b = bnd_idx.size() - fr[f].bnd_off;
        fr[f].k = kf[f];
        fr[f].m = kf[f] + b;
        fr[f].ld = fr[f].m + fr[f].k;
        fr[f].own_start = own_start[f];
        fr[f].parent = parent[f];
        fr[f].off = off;
        off += (int64_t)fr[f].ld * fr[f].ld;
        fr[f].rel_off = fr[f].bnd_off;  // rel is parallel to bnd_idx
        max_front = std::max(max_front, fr[f].m);
        if (parent[f] < 0) root_pivots = std::max(root_pivots, fr[f].k);
        double k = fr[f].k, bb = b;
        nnz_factors += (int64_t)(k * k + 2 * k * bb);
        // LU of the front plus the row / column operations on the augmentation
        front_flops[f] = 2.0 / 3 * k * k * k + 2 * k * k * bb + 2 * k * bb * bb + 2 * k * k * (k + bb);
        factor_flops += front_flops[f];
    }

    // ---- two-phase levels (mf_types.h, Level::two_phase) -----------------------------------------------------------
    // The boundary blocks of the solve operators, F[B,A] = -L21 L11^-1 and F[A,B] = -U11^-1 U12, cost k^2 b flops each;
    // without them a sweep over the front is two dependent mat-vecs instead of one.  That pays where the products of a
    // whole height of the tree (its fronts share their launches) outweigh two more launches in each of the ~2 x 20
    // sweeps of a step: 2e10 flops (rounds 4-5: 8e9, swept on the 32^3 / 40^3 / 48^3 blocks, 3e9 ... 5e10: flat between
    // 6e9 and 1.2e10, HISTORY.md section 5; round 6 below) -- the upper half of a 32^3-vertex block's tree and beyond, never a BASELINE mesh
    // (human ARAP, the largest: 11 GFLOP per factorisation in all).  SANM_MF_TWO_PHASE=1 / 0: every height / none; a
    // value above 1: the threshold in flops.
    std::vector<char> two_phase_h;
    {
        int32_t Hh = 0;
        for (int32_t f = 0; f < F; ++f) Hh = std::max(Hh, height[f] + 1);
        std::vector<double> h_k2b(Hh, 0.0);
        std::vector<int32_t> h_max_k(Hh, 0);
        for (int32_t f = 0; f < F; ++f) {
            const double k = fr[f].k, bb = fr[f].m - fr[f].k;
            h_k2b[height[f]] += 2 * k * k * bb;
            h_max_k[height[f]] = std::max(h_max_k[height[f]], fr[f].k);
        }
        const char* env = std::getenv("SANM_MF_TWO_PHASE");
        two_phase_h.assign(Hh, 0);
        const double env_v = env ? std::atof(env) : -1;  // (a value above 1: the threshold in flops, for sweeps)
        for (int32_t h = 0; h < Hh; ++h)
            // (round 6: 2e10, and never a height of small fronts.  The sum over a height crosses any threshold once the
            // height has fronts enough -- the leaf level of a 2.7 M-tet mesh, 14062 fronts of 57 pivots, sums to 15 GFLOP --
            // and a two-phase level loses the one-workgroup small-front kernel, the transposed forward operator and one
            // launch per sweep: 2.7 M tets, thresholds 8e9 / 2e10 / 4e10 / 8e10: solves 96.0 / 84.5 / 83.3 / 82.9, factor
            // 215.6 / 217.3 / 220.0 / 224.2 ms per step; block:48 172.1 / 172.0 / 174.7 / 179.3 ms per step)
            two_phase_h[h] = env ? (env_v > 1 ? h_k2b[h] >= env_v : (env_v != 0 && h_k2b[h] > 0))
                                 : (h_k2b[h] >= 2e10 && h_max_k[h] > 96);
        for (int32_t f = 0; f < F; ++f)
            if (two_phase_h[height[f]]) {
                const double k = fr[f].k, bb = fr[f].m - fr[f].k;
                front_flops[f] -= 2 * k * k * bb;
                factor_flops -= 2 * k * k * bb;
            }
    }

    // ---- tree-to-ranks distribution (MfSchedule::Dist, mf_types.h) -----------------------------------------------
    // Proportional mapping from the roots down: a node with the rank set R keeps R[0] as its owner and gives its
    // children disjoint slices of R in proportion to the factor work of their subtrees (heaviest child first, so that it
    // shares the node's owner and its Schur complement never travels); more children than ranks: each child to the rank
    // of the set with the least work so far, largest first.  A subtree whose set is one rank is that rank's (stage 0);
    // f_owner[f] = owner of every front, f_stage[f] >= 1 for the fronts above (the top).  One rank: no distribution.
    std::vector<int32_t> f_owner(F, -1), f_stage(F, 0);
    std::vector<int32_t> cut_roots;  // roots of the stage-0 subtrees, in front order (the same on every rank)
    auto& D = m_sched.dist;
    D.rank = rank;
    D.world = world;
    // When it pays: the exchanges add collectives to every solve, so small systems (a BASELINE-size mesh factors
    // 6 GFLOP in 1.9 ms of launch latency) stay replicated; SANM_DIST_SOLVER=1 / 0 forces it on / off, the default
    // takes it from 50 GFLOP per factorisation (block:24 and larger).
    const char* env_dist = std::getenv("SANM_DIST_SOLVER");
    const bool want_dist = world > 1 && (env_dist ? std::atoi(env_dist) != 0 : factor_flops >= 50e9);
    if (want_dist) {
        std::vector<double> sub_flops(front_flops);
        for (int32_t f = 0; f < F; ++f)
            if (parent[f] >= 0) sub_flops[parent[f]] += sub_flops[f];  // (postorder: children come first)
        // (a subtree lighter than a 64th of a rank's share is never spread over several ranks: latency, not work)
        const double min_split = factor_flops / (64.0 * world);
        struct Task {
            int32_t f;                  // -1: the virtual root above the tree's roots
            std::vector<int32_t> ranks;
        };
        std::vector<int32_t> tree_roots;
        for (int32_t f = 0; f < F; ++f)
            if (parent[f] < 0) tree_roots.push_back(f);
        // The mapping in two variants -- a child always gets at least one rank of its own (`ride` false), or the light side
        // of a lopsided node rides with the least loaded rank of the set (true) --; kept is the one with the shorter
        // critical path (a rank takes its fronts in elimination order, a front waits for its children).  Neither wins
        // everywhere: armadillo x64 at 8 ranks 32.7 % against 37.7 % of the flops on the path, the forest of a mesh with a
        // one-vertex component the other way round (one rank would own nothing but that vertex).
        auto map_tree = [&](bool ride, std::vector<int32_t>& owner, std::vector<int32_t>& top, std::vector<char>& is_cut_root) {
        owner.assign(F, -1);
        top.assign(F, 0);
        is_cut_root.assign(F, 0);
        std::vector<Task> stack;
        {
            Task t{-1, {}};
            for (int r = 0; r < world; ++r) t.ranks.push_back(r);
            stack.push_back(std::move(t));
        }
        while (!stack.empty()) {
            Task t = std::move(stack.back());
            stack.pop_back();
            const std::vector<int32_t>& ch = t.f < 0 ? tree_roots : children[t.f];
            if (t.f >= 0) {
                owner[t.f] = t.ranks[0];
                if (t.ranks.size() == 1 || ch.empty() || sub_flops[t.f] < min_split) {
                    is_cut_root[t.f] = 1;  // the whole subtree is this rank's
                    continue;
                }
                top[t.f] = 1;  // (a top front; its stage is computed below)
            }
            std::vector<int32_t> by_work(ch);
            std::stable_sort(by_work.begin(), by_work.end(), [&](int32_t a, int32_t b) { return sub_flops[a] > sub_flops[b]; });
            const int nr = (int)t.ranks.size(), nc = (int)by_work.size();
            // Every child gets the whole number of ranks in its share of the set (`ride`: possibly none; else at least one
            // while ranks last), the ranks left over go by largest remainder; the children left without a rank of their
            // own ride with the least loaded rank of the set, largest first.
            double W = 0;
            for (int32_t c : by_work) W += sub_flops[c];
            std::vector<int> cnt(nc, 0);
            std::vector<double> want(nc);
            int left = nr;
            for (int i = 0; i < nc; ++i) {
                want[i] = W > 0 ? sub_flops[by_work[i]] / W * nr : 0.0;
                cnt[i] = std::min(std::max((int)want[i], ride ? 0 : 1), left);
                left -= cnt[i];
            }
            while (left > 0) {
                int q = 0;
                for (int i = 1; i < nc; ++i)
                    if (want[i] - cnt[i] > want[q] - cnt[q]) q = i;
                ++cnt[q];
                --left;
            }
            std::vector<double> load(nr, 0.0);
            int at = 0;
            for (int i = 0; i < nc; ++i) {
                if (cnt[i] == 0) continue;
                Task u{by_work[i], {}};
                u.ranks.assign(t.ranks.begin() + at, t.ranks.begin() + at + cnt[i]);
                for (int r = at; r < at + cnt[i]; ++r) load[r] = sub_flops[by_work[i]] / cnt[i];
                at += cnt[i];
                stack.push_back(std::move(u));
            }
            for (int i = 0; i < nc; ++i) {
                if (cnt[i] != 0) continue;
                int q = 0;
                for (int r = 1; r < nr; ++r)
                    if (load[r] < load[q]) q = r;
                load[q] += sub_flops[by_work[i]];
                stack.push_back(Task{by_work[i], {t.ranks[q]}});
            }
        }
        // fronts below a cut root inherit its owner (parents have larger ids: walk down from the top)
        for (int32_t f = F - 1; f >= 0; --f)
            if (owner[f] < 0) {
                sanm_check(parent[f] >= 0 && owner[parent[f]] >= 0, "front %d has no owner", f);
                owner[f] = owner[parent[f]];
            }
        // the critical path in flops
        std::vector<double> finish(F, 0.0), rank_time(world, 0.0);
        double crit = 0;
        for (int32_t f = 0; f < F; ++f) {
            double t0 = rank_time[owner[f]];
            for (int32_t c : children[f]) t0 = std::max(t0, finish[c]);
            finish[f] = t0 + front_flops[f];
            rank_time[owner[f]] = finish[f];
            crit = std::max(crit, finish[f]);
        }
        return crit;
        };
        std::vector<char> is_cut_root;
        {
            std::vector<int32_t> o2, t2;
            std::vector<char> c2;
            const double crit_own = map_tree(false, f_owner, f_stage, is_cut_root);
            const double crit_ride = map_tree(true, o2, t2, c2);
            if (crit_ride < crit_own) {
                f_owner.swap(o2);
                f_stage.swap(t2);
                is_cut_root.swap(c2);
            }
            if (std::getenv("SANM_MF_DEBUG"))
                std::fprintf(stderr, "mf dist: critical path %.2f GF with a rank for every child, %.2f with riders\n",
                             crit_own / 1e9, crit_ride / 1e9);
        }
        for (int32_t f = 0; f < F; ++f)
            if (is_cut_root[f]) cut_roots.push_back(f);
        // stages of the top fronts (children first)
        int32_t nr_stage = 1;
        for (int32_t f = 0; f < F; ++f) {
            if (f_stage[f] == 0) continue;
            int32_t st = 1;
            for (int32_t c : children[f]) st = std::max(st, f_stage[c] + (f_owner[c] != f_owner[f] ? 1 : 0));
            f_stage[f] = st;
            nr_stage = std::max(nr_stage, st + 1);
        }
        D.enabled = true;
        D.nr_stage = nr_stage;
        D.nr_subtree = (int32_t)cut_roots.size();
        for (int32_t c : cut_roots) D.nr_subtree_own += f_owner[c] == rank;
        D.rank_flops.assign(world, 0.0);
        D.rank_top_flops.assign(world, 0.0);
        D.rank_nnz.assign(world, 0.0);
        D.stage_flops.assign((size_t)nr_stage * world, 0.0);
        D.stage_nnz.assign((size_t)nr_stage * world, 0.0);
        for (int32_t f = 0; f < F; ++f) {
            const double fk = fr[f].k, fb = fr[f].m - fr[f].k, fnnz = fk * fk + 2 * fk * fb;
            const int o = f_owner[f], st = f_stage[f];
            D.rank_nnz[o] += fnnz;
            D.stage_flops[(size_t)st * world + o] += front_flops[f];
            D.stage_nnz[(size_t)st * world + o] += fnnz;
            if (st > 0) {
                D.nnz_top += fnnz;
                D.flops_top += front_flops[f];
                D.rank_top_flops[o] += front_flops[f];
                ++D.nr_front_top;
                if (o == rank) D.flops_top_own += front_flops[f];
            } else {
                D.rank_flops[o] += front_flops[f];
                if (o == rank) {
                    D.flops_own += front_flops[f];
                    ++D.nr_front_own;
                }
            }
        }
        {
            double mx = 0, sum = 0;
            for (double l : D.rank_flops) {
                mx = std::max(mx, l);
                sum += l;
            }
            D.imbalance = sum > 0 ? mx * world / sum : 1.0;
        }
    } else {
        D.flops_top = factor_flops;
        D.flops_critical = factor_flops;
        D.nr_front_top = F;
    }
    // A rank of the distributed solver stores the fronts it factors and, of the children other ranks factor for its
    // fronts, the Schur block it receives -- at the address the child's descriptor gives it, F[B,B] with the child's own
    // row stride, so that the extend-add reads it like a child of its own (round 6; every rank used to hold the whole
    // store: four ranks of a 2.7 M-tet mesh did not fit one device, and capacity did not grow with the ranks).  Nothing
    // else of another rank's front is ever addressed (`front_here`: the scatter of A and the augmentation's identity
    // skip those fronts).  SANM_DIST_FULL_STORE=1: the whole store on every rank, as before.
    std::vector<uint8_t> front_here;
    if (D.enabled && !std::getenv("SANM_DIST_FULL_STORE")) {
        front_here.assign(F, 0);
        off = 0;
        for (int32_t f = 0; f < F; ++f) {
            const int64_t k = fr[f].k, ld = fr[f].ld, b = fr[f].m - fr[f].k;
            if (f_owner[f] == rank) {
                front_here[f] = 1;
                fr[f].off = off;
                off += ld * ld;
            } else if (parent[f] >= 0 && f_owner[parent[f]] == rank && b > 0) {
                fr[f].off = off - (2 * k * ld + 2 * k);  // (so that F[B,B] starts at `off`)
                off += (b - 1) * ld + b;
            } else {
                fr[f].off = 0;  // (never addressed)
            }
        }
    }
    front_doubles = off;
    // the front store's memory, asked for now on a thread of its own (deferred constructors only: the owner thread is
    // busy with the driver's tables, and mapping tens of GB takes the device 0.1-1 s)
    if (m_defer && off * sizeof(double) >= (size_t(1) << 28)) {
        m_front_job.be = m_be;
        m_front_job.job = std::async(std::launch::async, [be = m_be, bytes = (size_t)off * sizeof(double)] { return be->alloc_detached(bytes); });
    }

    // position of a (new-numbered) variable x inside front f
    auto pos_in_front = [&](int32_t f, int32_t x) -> int32_t {
        if (x >= fr[f].own_start && x < fr[f].own_start + fr[f].k) return x - fr[f].own_start;
        const int32_t* b = bnd_idx.data() + fr[f].bnd_off;
        int32_t nb = fr[f].m - fr[f].k;
        const int32_t* it = std::lower_bound(b, b + nb, x);
        sanm_check(it != b + nb && *it == x, "variable %d is not part of front %d", x, f);
        return 2 * fr[f].k + (it - b);  // [pivot | augmentation | boundary]
    };

    auto rel_keep = std::make_shared<std::vector<int32_t>>(bnd_idx.size(), -1);
    std::vector<int32_t>& rel = *rel_keep;
    {
        // a front's boundary and its parent's [pivots | boundary] both ascend in the new numbering: one walk along both
        std::vector<std::string> errs(64);
        parallel_ranges(F, 16, [&](int64_t f0, int64_t f1, int t) {
            for (int32_t f = (int32_t)f0; f < (int32_t)f1; ++f) {
                if (parent[f] < 0) {
                    if (fr[f].m != fr[f].k) errs[t % 64] = "root front has a boundary";
                    continue;
                }
                const MfFrontDev& P = fr[parent[f]];
                const int32_t* pb = bnd_idx.data() + P.bnd_off;
                const int32_t npb = P.m - P.k;
                int32_t at = 0;
                for (int32_t j = 0; j < fr[f].m - fr[f].k; ++j) {
                    const int32_t x = bnd_idx[fr[f].bnd_off + j];
                    if (x >= P.own_start && x < P.own_start + P.k) {
                        rel[fr[f].rel_off + j] = x - P.own_start;
                        continue;
                    }
                    while (at < npb && pb[at] < x) ++at;
                    if (at == npb || pb[at] != x) {
                        errs[t % 64] = "a boundary variable is not part of the parent front";
                        break;
                    }
                    rel[fr[f].rel_off + j] = 2 * P.k + at;  // [pivot | augmentation | boundary]
                }
            }
        });
        for (const auto& e : errs) sanm_check(e.empty(), "%s", e.c_str());
    }

    // inboxes of the solve: child c (slot j of its parent p) stores the update entry of its
    // boundary row i at inbox[p][j][logical row of i in p]; no gather lists are walked at solve time
    std::vector<int32_t> upd_dst(bnd_idx.size(), -1);
    int64_t inbox_doubles = 0;
    {
        auto logical = [&](int32_t f, int32_t pos) { return pos < fr[f].k ? pos : pos - fr[f].k; };
        for (int32_t f = 0; f < F; ++f) {
            fr[f].nch = children[f].size();
            fr[f].inbox_off = inbox_doubles;
            inbox_doubles += (int64_t)fr[f].nch * fr[f].m;
            sanm_check(inbox_doubles < (int64_t(1) << 31), "solve workspace exceeds 32-bit indexing");
        }
        for (int32_t f = 0; f < F; ++f)
            for (size_t j = 0; j < children[f].size(); ++j) {
                const int32_t c = children[f][j];
                for (int32_t i = 0; i < fr[c].m - fr[c].k; ++i)
                    upd_dst[fr[c].bnd_off + i] =
                            fr[f].inbox_off + (int32_t)j * fr[f].m + logical(f, rel[fr[c].rel_off + i]);
            }
    }

    lap("fronts, rel, inboxes");
    if (dbg_clock) {  // the positions once more, each by its own search
        for (int32_t f = 0; f < F; ++f)
            for (int32_t j = 0; parent[f] >= 0 && j < fr[f].m - fr[f].k; ++j)
                sanm_check(rel[fr[f].rel_off + j] == pos_in_front(parent[f], bnd_idx[fr[f].bnd_off + j]),
                           "rel: the walk and the search disagree (front %d, row %d)", f, j);
        std::fprintf(stderr, "mf analysis: rel checked entry by entry\n");
        lap("(debug: that check)");
    }
    // owner front of every new index
    std::vector<int32_t> owner(n);
    for (int32_t f = 0; f < F; ++f)
        for (int32_t q = 0; q < fr[f].k; ++q) owner[fr[f].own_start + q] = f;

    // The scatter map of A -- front_store[a_dst[p]] = val[p] -- is the backend's to fill when the first factorisation
    // runs (mf_types.h, mf_scatter_slot: a search of a boundary list per entry, nothing on a device and 8 bytes per entry
    // that are neither computed here nor uploaded; round 6).  Every entry has its place because the supervariable graph
    // was built from these very rows; SANM_MF_DEBUG looks every one of them up.
    const int64_t nnzA = blocks ? (int64_t)blocks->block * blocks->block * (int64_t)blocks->qcol->size() : (int64_t)col_p->size();
    if (dbg_clock && rowptr_p) {
        const std::vector<uint32_t>&rowptr = *rowptr_p, &col = *col_p;
        std::vector<std::string> errs(64);
        parallel_ranges(n, 2048, [&](int64_t r0, int64_t r1, int t) {
            for (int64_t i = r0; i < r1; ++i)
                for (uint32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
                    const int32_t pi = perm[i], pj = perm[col[p]];
                    const MfFrontDev& f = fr[owner[std::min(pi, pj)]];
                    for (int32_t x : {pi, pj}) {
                        if (x >= f.own_start && x < f.own_start + f.k) continue;
                        const int32_t* bb = bnd_idx.data() + f.bnd_off;
                        if (!std::binary_search(bb, bb + (f.m - f.k), x)) errs[t % 64] = "an entry of A has no place in its front";
                    }
                }
        });
        for (const auto& e : errs) sanm_check(e.empty(), "%s", e.c_str());
        std::fprintf(stderr, "mf analysis: every entry of A has its place\n");
        lap("(debug: places of A's entries)");
    }

    // levels: fronts by height, by decreasing k inside a level.  Distributed: this rank's own fronts only, stage after
    // stage (MfSchedule::Dist); fronts of other ranks are in no level of this schedule.
    int32_t H = 0;
    for (int32_t f = 0; f < F; ++f) H = std::max(H, height[f] + 1);
    std::vector<int32_t> level_fronts;
    std::vector<int32_t> ea_children;
    int64_t tmp_doubles = 1;
    std::vector<std::vector<int32_t>> level_lists;
    D.stage_level.assign(1, 0);
    for (int32_t st = 0; st < D.nr_stage; ++st) {
        std::vector<std::vector<int32_t>> by_h(H);
        for (int32_t f = 0; f < F; ++f)
            if (!D.enabled || (f_owner[f] == rank && f_stage[f] == st)) by_h[height[f]].push_back(f);
        for (auto& fs : by_h)
            if (!fs.empty()) level_lists.push_back(std::move(fs));
        D.stage_level.push_back((int32_t)level_lists.size());
    }
    H = (int32_t)level_lists.size();
    nr_level = H;
    m_sched.levels.resize(H);
    for (int32_t h = 0; h < H; ++h) {
        auto& L = m_sched.levels[h];
        L.front_begin = level_fronts.size();
        std::vector<int32_t> fs = level_lists[h];
        std::stable_sort(fs.begin(), fs.end(), [&](int32_t a, int32_t b) { return fr[a].k > fr[b].k; });
        level_fronts.insert(level_fronts.end(), fs.begin(), fs.end());
        L.front_end = level_fronts.size();
        L.max_m = L.max_k = L.max_b = 0;
        L.sum_m = L.sum_k = 0;
        L.two_phase = two_phase_h[height[fs[0]]] != 0;  // (a level's fronts share their height)
        size_t max_children = 0;
        for (int32_t f : fs) {
            L.max_m = std::max(L.max_m, fr[f].m);
            L.max_k = std::max(L.max_k, fr[f].k);
            L.max_b = std::max(L.max_b, fr[f].m - fr[f].k);
            L.sum_m += fr[f].m;
            L.sum_k += fr[f].k;
            max_children = std::max(max_children, children[f].size());
        }
        {
            // (SANM_MF_FWD_T=0: never; the merged top block reads F[B,A] itself: not together with SANM_MF_TOP)
            const char* env_ft = std::getenv("SANM_MF_FWD_T");
            const bool want = env_ft ? std::atoi(env_ft) != 0 : true;
            static const int fwd_t_max_k = std::getenv("SANM_MF_FWD_T_MAX_K") ? std::min(kFwdTMaxK, std::atoi(std::getenv("SANM_MF_FWD_T_MAX_K"))) : kFwdTMaxK;
            // (pivot blocks of 129 .. 256: a thread per boundary row walks a column of up to 64 entries per wavefront
            // quarter -- it pays where the level has rows enough to fill the chip with such threads: refine:armadillo_
            // small:1, levels of 152 / 80 / 47 fronts: solves 10.76 -> 10.12 ms per step; the 32-front level of
            // armadillo_small, 13 k rows, lost 0.11 ms)
            const bool rows_enough = L.max_k <= 128 || L.sum_m - L.sum_k >= 24576;
            L.fwd_t = want && !L.two_phase && L.max_k <= fwd_t_max_k && rows_enough && L.max_b > 0 && !std::getenv("SANM_MF_TOP");
        }
        {
            int64_t t = 0;
            for (int32_t f : fs) {
                fr[f].tmp_off = t;
                t += 2 * (int64_t)fr[f].k * (fr[f].m - fr[f].k);
            }
            tmp_doubles = std::max(tmp_doubles, t);
        }
        for (int32_t f : fs) L.front_k.push_back(fr[f].k);
        L.nr_panel = (L.max_k + MF_NB - 1) / MF_NB;
        L.panel_cnt.assign(L.nr_panel, 0);
        for (int32_t f : fs)
            for (int32_t p = 0; p * MF_NB < fr[f].k; ++p) L.panel_cnt[p]++;
        for (size_t r = 0; r < max_children; ++r) {
            int32_t b = ea_children.size(), mb = 0;
            for (int32_t f : fs)
                if (r < children[f].size()) {
                    int32_t c = children[f][r];
                    ea_children.push_back(c);
                    mb = std::max(mb, fr[c].m - fr[c].k);
                }
            L.ea_rounds.emplace_back(b, (int32_t)ea_children.size());
            L.ea_max_b.push_back(mb);
        }
        for (int32_t f : fs)
            if (!children[f].empty()) L.ea0_max_bp = std::max(L.ea0_max_bp, fr[f].m - fr[f].k);
    }

    // tile lists of the GEMM passes (mf_types.h, Level::g1_tiles / g2_tiles)
    for (int32_t h = 0; h < H; ++h) {
        auto& L = m_sched.levels[h];
        std::vector<uint32_t> t1, t2;
        std::vector<uint32_t> tall_q[8];
        size_t tall_next = 0;
        constexpr int GT = MF_GT;
        for (int32_t i = L.front_begin; i < L.front_end; ++i) {
            const auto& f = fr[level_fronts[i]];
            const uint32_t loc = (uint32_t)(i - L.front_begin);
            const int k = f.k, b = f.m - f.k;
            if (b == 0) continue;
            const int tk = (k + GT - 1) / GT, tb = (b + GT - 1) / GT;
            sanm_check(tk < 32768 && tb < 32768, "front too large for the tile lists");
            auto push = [](std::vector<uint32_t>& v, uint32_t front, int which, int ti, int tj) {
                v.push_back(front);
                v.push_back((uint32_t)which << 30 | (uint32_t)ti << 15 | (uint32_t)tj);
            };
            for (int ti = 0; ti < tk; ++ti)
                for (int tj = 0; tj < tb; ++tj) push(t1, loc, 0, ti, tj);  // tmpU (k x b)
            for (int ti = 0; ti < tb; ++ti)
                for (int tj = 0; tj < tk; ++tj) push(t1, loc, 1, ti, tj);  // tmpL (b x k)
            // (the interior of the Schur complement of a big front goes out in tall tiles: mf_types.h)
#ifdef SANM_MF_OLD_STAGING  // (A/B build without the tall tiles: every tile of the Schur complement in the GEMM-2 pass)
            const bool tall_tiles = false;
#else
            const bool tall_tiles = true;
#endif
            for (int ti = 0; ti < tb; ++ti)
                for (int tj = 0; tj < tb; ++tj)
                    if (!(tall_tiles && mf_gemm2_is_tall(k, b, ti, tj))) push(t2, loc, 0, ti, tj);
            if (tall_tiles && mf_gemm2_is_tall(k, b, 0, 0)) {
                // supertiles of tall tiles, dealt to the eight queues in turn
                const int TR = b / (2 * GT), TC = b / GT;
                for (int sr = 0; sr < TR; sr += MF_ST_R)
                    for (int sc = 0; sc < TC; sc += MF_ST_C) {
                        auto& q = tall_q[tall_next++ % 8];
                        for (int tp = sr; tp < std::min(sr + MF_ST_R, TR); ++tp)
                            for (int tj = sc; tj < std::min(sc + MF_ST_C, TC); ++tj) {
                                q.push_back(loc);
                                q.push_back((uint32_t)tp << 15 | (uint32_t)tj);
                            }
                    }
            }
            if (!L.two_phase) {
                for (int ti = 0; ti < tb; ++ti)
                    for (int tj = 0; tj < tk; ++tj) push(t2, loc, 1, ti, tj);
                for (int ti = 0; ti < tk; ++ti)
                    for (int tj = 0; tj < tb; ++tj) push(t2, loc, 2, ti, tj);
            }
        }
        L.n_g1 = (int32_t)(t1.size() / 2);
        L.n_g2 = (int32_t)(t2.size() / 2);
        upload_to(L.g1_tiles, std::move(t1));
        upload_to(L.g2_tiles, std::move(t2));
        {
            // the eight queues interleaved: entry n of the list is entry n / 8 of queue n % 8 (a padding entry where
            // that queue has run out)
            size_t longest = 0;
            for (const auto& q : tall_q) longest = std::max(longest, q.size() / 2);
            std::vector<uint32_t> t3;
            t3.reserve(longest * 16);
            for (size_t i = 0; i < longest; ++i)
                for (const auto& q : tall_q) {
                    const bool have = 2 * i < q.size();
                    t3.push_back(have ? q[2 * i] : ~0u);
                    t3.push_back(have ? q[2 * i + 1] : 0u);
                }
            L.n_gt = (int32_t)(t3.size() / 2);
            upload_to(L.gt_tiles, std::move(t3));
        }
    }

    if (std::getenv("SANM_MF_DEBUG")) {
        for (int32_t h = 0; h < H; ++h) {
            const auto& L = m_sched.levels[h];
            int64_t solve_elems = 0, fill = 0, sk = 0, sb = 0;
            double fl = 0, g_schur = 0, g_k2b = 0;
            for (int32_t i = L.front_begin; i < L.front_end; ++i) {
                const auto& f = fr[level_fronts[i]];
                solve_elems += (int64_t)(f.m + f.k) * f.k;
                const double k = f.k, bb = f.m - f.k;
                fill += (int64_t)(k * k + 2 * k * bb);
                fl += 2.0 / 3 * k * k * k + 2 * k * k * bb + 2 * k * bb * bb + 2 * k * k * (k + (L.two_phase ? 0 : bb));
                sk += f.k;
                sb += f.m - f.k;
                g_schur += 2 * k * bb * bb;
                g_k2b += k * k * bb;
            }
            const int nf = L.front_end - L.front_begin;
            std::fprintf(stderr, "mf level %d: fronts=%d max_k=%d max_m=%d max_b=%d panels=%d solve_MB=%.2f fill=%.2fM "
                         "GF=%.2f avg_k=%.0f avg_b=%.0f schurGF=%.2f k2bGF=%.2f%s\n", h, nf, L.max_k, L.max_m, L.max_b,
                         L.nr_panel, solve_elems * 8 / 1e6, fill / 1e6, fl / 1e9, (double)sk / nf, (double)sb / nf,
                         g_schur / 1e9, g_k2b / 1e9, L.two_phase ? " two-phase" : "");
        }
    }

    lap("levels");
    // device copies
    m_dev.n = n;
    m_dev.nnzA = nnzA;
    m_dev.nr_front = F;
    m_dev.nr_level = H;
    upload_to(m_dev.fronts, fr);
    upload_to(m_dev.level_fronts, level_fronts);
    {
        std::vector<MfFrontDev> lf(level_fronts.size());
        for (size_t i = 0; i < lf.size(); ++i) lf[i] = fr[level_fronts[i]];
        m_sched.h_lfronts = lf;
        upload_to(m_dev.lfronts, std::move(lf));
    }
    upload_to(m_dev.upd_dst, std::move(upd_dst));
    // (one more double at the end that nothing writes: the padding slot of the merged top block's lists)
    alloc_to(m_dev.inbox_store, (inbox_doubles + 1) * sizeof(double), true);
    upload_kept(m_dev.bnd_idx, bnd_idx_keep, bnd_idx.data(), bnd_idx.size());
    upload_kept(m_dev.rel, rel_keep, rel.data(), rel.size());
    upload_to(m_dev.perm, perm);
    upload_to(m_dev.own_front, std::move(owner));
    if (!front_here.empty()) upload_to(m_dev.front_here, front_here);
    alloc_to(m_dev.a_dst, (size_t)std::max<int64_t>(nnzA, 1) * sizeof(int64_t), false);
    upload_to(m_sched.ea_children, ea_children);
    {
        // the parent-side map of round 0 (mf_types.h, MfSchedule::ea_inv)
        std::vector<int32_t> ea_inv(bnd_idx.size(), -1);
        for (int32_t f = 0; f < F; ++f) {
            if (children[f].empty()) continue;
            const int32_t c = children[f][0], bc = fr[c].m - fr[c].k;
            for (int32_t i = 0; i < bc; ++i) {
                const int32_t pos = rel[fr[c].rel_off + i];
                if (pos >= 2 * fr[f].k) ea_inv[fr[f].bnd_off + pos - 2 * fr[f].k] = i;
            }
        }
        upload_to(m_sched.ea_inv, std::move(ea_inv));
        // blocks of rows the prologue zeroes (this rank's fronts when the tree is distributed)
        std::vector<int32_t> zb;
        int64_t zd = 0;
        for (int32_t f = 0; f < F; ++f) {
            if (D.enabled && f_owner[f] != rank) continue;
            for (int32_t r0 = 0; r0 < fr[f].ld; r0 += MF_ZERO_ROWS) {
                zb.push_back(f);
                zb.push_back(r0);
            }
            const int64_t k = fr[f].k, b = fr[f].m - fr[f].k;
            zd += 4 * k * k + 2 * k * b;
        }
        m_sched.n_zero_blocks = (int32_t)(zb.size() / 2);
        m_sched.zero_doubles = zd;
        upload_to(m_sched.zero_blocks, std::move(zb));
    }
    lap("device tables: uploads");
    m_dev.front_store_size = off;
    {
        DeviceOp op;
        op.bytes = off * sizeof(double);
        op.detached = true;
        op.set = [this](void* p) { m_dev.front_store = static_cast<double*>(p); };
        if (m_defer) m_pending.push_back(std::move(op));
        else run_op(op);
    }
    lap("device tables: front store");
    // (n doubles, and room behind them for the solution of the merged top block, MfSchedule::Top)
    const char* env_top = std::getenv("SANM_MF_TOP");
    const int32_t top_max_n = env_top ? std::atoi(env_top) : 0;
    alloc_to(m_dev.work, (n + std::max(top_max_n, 0)) * sizeof(double), false);
    alloc_to(m_dev.work2, n * sizeof(double), false);
    alloc_to(m_dev.tmp_store, tmp_doubles * sizeof(double), false);
    {
        // status word, and the largest |a_ij| behind it
        DeviceOp op;
        op.bytes = 64;
        op.zero = true;
        op.set = [this](void* p) {
            m_dev.status = static_cast<int32_t*>(p);
            m_dev.piv_amax = reinterpret_cast<double*>(m_dev.status) + 1;
        };
        if (m_defer) m_pending.push_back(std::move(op));
        else run_op(op);
    }

    lap("device tables");
    // ---- exchange tables of the distributed schedule (MfSchedule::Dist) -----------------------------------------
    if (D.enabled) {
        const int32_t S = D.nr_stage;
        D.schur.resize(S);
        D.inbox.resize(S);
        D.sol.resize(S);
        std::vector<std::vector<MfCopy2D>> sp(S), su(S), ip(S), iu(S), lp(S), lu(S);
        // edges of the tree whose ends have different owners, by the parent's stage
        for (int32_t c = 0; c < F; ++c) {
            const int32_t p = parent[c];
            if (p < 0 || f_owner[p] == f_owner[c]) continue;
            const int32_t st = f_stage[p], b = fr[c].m - fr[c].k;
            sanm_check(st >= 1 && f_stage[c] < st, "distributed schedule: edge %d -> %d does not cross a stage", c, p);
            if (b == 0) continue;
            const int src = f_owner[c], dst = f_owner[p];
            auto& E = D.schur[st];
            const int64_t blk = fr[c].off + (int64_t)2 * fr[c].k * fr[c].ld + 2 * fr[c].k;  // F[B,B]
            E.xfers.push_back({src, dst, E.doubles, (int64_t)b * b, f_stage[c]});
            if (rank == src) sp[st].push_back({blk, E.doubles, b, b, fr[c].ld, b});
            if (rank == dst) su[st].push_back({E.doubles, blk, b, b, b, fr[c].ld});
            E.doubles += (int64_t)b * b;
            E.max_rows = E.max_cols = std::max(E.max_rows, b);
            // the child's slot row in its parent's inbox
            const auto& ch = children[p];
            const int32_t j = (int32_t)(std::find(ch.begin(), ch.end(), c) - ch.begin());
            const int64_t row = fr[p].inbox_off + (int64_t)j * fr[p].m;
            auto& I = D.inbox[st];
            I.xfers.push_back({src, dst, I.doubles, fr[p].m, f_stage[c]});
            if (rank == src) ip[st].push_back({row, I.doubles, 1, fr[p].m, fr[p].m, fr[p].m});
            if (rank == dst) iu[st].push_back({I.doubles, row, 1, fr[p].m, fr[p].m, fr[p].m});
            I.doubles += fr[p].m;
            I.max_rows = 1;
            I.max_cols = std::max(I.max_cols, fr[p].m);
        }
        // solution entries: the pivots of every stage's fronts, owner -> everyone; runs of fronts with one owner are
        // one range of the permuted vector (front order = order of the unknowns)
        for (int32_t f = 0; f < F; ++f) {
            auto& X = D.sol[f_stage[f]].xfers;
            const int64_t b0 = fr[f].own_start, k = fr[f].k;
            if (!X.empty() && X.back().src == f_owner[f] && X.back().off + X.back().cnt == b0) X.back().cnt += k;
            else X.push_back({f_owner[f], -1, b0, k, f_stage[f]});
        }
        for (int32_t st = 0; st < S; ++st) {
            // (through the all-reduce callback the ranges of the top stages travel packed; stage 0's -- most of the
            // vector -- are summed in place, the ranges of others zeroed first: zero_ranges)
            auto& X = D.sol[st];
            for (const auto& x : X.xfers) {
                if (st > 0) {
                    if (rank == x.src) lp[st].push_back({x.off, X.doubles, 1, (int32_t)x.cnt, (int32_t)x.cnt, (int32_t)x.cnt});
                    else lu[st].push_back({X.doubles, x.off, 1, (int32_t)x.cnt, (int32_t)x.cnt, (int32_t)x.cnt});
                    X.max_rows = 1;
                    X.max_cols = std::max<int32_t>(X.max_cols, (int32_t)x.cnt);
                }
                X.doubles += x.cnt;
            }
        }
        int64_t stage_doubles = 1;
        for (int32_t st = 0; st < S; ++st) {
            auto fin = [&](MfSchedule::Exchange& E, std::vector<MfCopy2D>& pk, std::vector<MfCopy2D>& up, bool staged) {
                E.n_pack = (int32_t)pk.size();
                E.n_unpack = (int32_t)up.size();
                upload_to(E.pack, std::move(pk));
                upload_to(E.unpack, std::move(up));
                if (staged) stage_doubles = std::max(stage_doubles, E.doubles);
            };
            fin(D.schur[st], sp[st], su[st], true);
            fin(D.inbox[st], ip[st], iu[st], true);
            fin(D.sol[st], lp[st], lu[st], st > 0);
            D.schur_doubles += D.schur[st].doubles;
            D.inbox_doubles += D.inbox[st].doubles;
        }
        D.stage_doubles = stage_doubles;
        alloc_to(D.stage, stage_doubles * sizeof(double), false);
        {
            // critical path in flops: end(r, s) = max(end(r, s - 1), end of the stages that send to (r, s)) + flops(r, s)
            std::vector<double> end((size_t)S * world, 0.0);
            for (int32_t st = 0; st < S; ++st)
                for (int r = 0; r < world; ++r) {
                    double t0 = st > 0 ? end[(size_t)(st - 1) * world + r] : 0.0;
                    for (const auto& x : D.schur[st].xfers)
                        if (x.dst == r) t0 = std::max(t0, end[(size_t)x.src_stage * world + x.src]);
                    end[(size_t)st * world + r] = t0 + D.stage_flops[(size_t)st * world + r];
                    D.flops_critical = std::max(D.flops_critical, end[(size_t)st * world + r]);
                }
        }
        // entries of the permuted solution this rank does not speak for in the last exchange (callback form: a sum in
        // place over the whole vector): the pivots of every front it does not own
        for (int32_t f = 0; f < F; ++f) {
            if (f_owner[f] == rank) continue;
            const int32_t b0 = fr[f].own_start, e0 = b0 + fr[f].k;
            if (!D.zero_ranges.empty() && D.zero_ranges.back().second == b0) D.zero_ranges.back().second = e0;
            else D.zero_ranges.emplace_back(b0, e0);
        }
        // the part of the front storage this rank factors in (and the children of its top fronts, whose Schur
        // complements arrive whole): what the factorisation's prologue zeroes
        for (int32_t f = 0; f < F; ++f) {
            if (f_owner[f] != rank) continue;
            const int64_t b0 = fr[f].off, e0 = b0 + (int64_t)fr[f].ld * fr[f].ld;
            if (!D.own_store.empty() && D.own_store.back().second == b0) D.own_store.back().second = e0;
            else D.own_store.emplace_back(b0, e0);
        }
        if (std::getenv("SANM_MF_DEBUG")) {
            std::fprintf(stderr, "mf dist: rank %d of %d: %d subtrees (%d own), %d stages, fronts own %d top %d, GF own %.2f "
                         "top %.2f (own %.2f) of %.2f, critical path %.2f GF, levels %d, schur exchange %.2f MB, inbox %.1f KB, "
                         "%zu zero ranges, %zu own store ranges\n", rank, world, D.nr_subtree, D.nr_subtree_own, S,
                         D.nr_front_own, D.nr_front_top, D.flops_own / 1e9, D.flops_top / 1e9, D.flops_top_own / 1e9,
                         factor_flops / 1e9, D.flops_critical / 1e9, H, D.schur_doubles * 8 / 1e6, D.inbox_doubles * 8 / 1e3,
                         D.zero_ranges.size(), D.own_store.size());
            for (int32_t st = 0; st < S; ++st) {
                std::fprintf(stderr, "mf dist:   stage %d: levels [%d, %d), GF per rank", st, D.stage_level[st], D.stage_level[st + 1]);
                for (int r = 0; r < world; ++r) std::fprintf(stderr, " %.1f", D.stage_flops[(size_t)st * world + r] / 1e9);
                std::fprintf(stderr, "; schur %.1f MB in %zu transfers\n", D.schur[st].doubles * 8 / 1e6, D.schur[st].xfers.size());
            }
        }
    }

    // ---- merged top block (mf_types.h, MfSchedule::Top): the root and the level below it ----------------------
    // Opt-in (SANM_MF_TOP = largest n_T to merge, e.g. 2048: 32 MB; default 0 = off).  Measured on armadillo_small
    // (n_T = 1071): 3 launches fewer per solve, 20 solves: -0.13 ms; two batched GEMM launches of 47 us per
    // factorisation: +0.095 ms; the step time does not move.
    {
        auto& T = m_sched.top;
        const int32_t max_n = top_max_n;
        bool ok = max_n > 0 && !D.enabled && H >= 3 && m_sched.levels[H - 1].front_end - m_sched.levels[H - 1].front_begin == 1;
        ok = ok && !m_sched.levels[H - 1].two_phase && !m_sched.levels[H - 2].two_phase;  // (needs the one-launch operators)
        std::vector<int32_t> tf;
        if (ok) {
            const auto& L = m_sched.levels[H - 2];
            for (int32_t i = L.front_begin; i < L.front_end; ++i) tf.push_back(level_fronts[i]);
            const int32_t root = level_fronts[m_sched.levels[H - 1].front_begin];
            for (int32_t f : tf) ok = ok && parent[f] == root;
            tf.push_back(root);
            ok = ok && (int)tf.size() <= MF_TOP_MAXF && fr[root].m == fr[root].k;
        }
        if (ok) {
            T.nf = tf.size();
            T.off[0] = 0;
            for (int i = 0; i < T.nf; ++i) T.off[i + 1] = T.off[i] + fr[tf[i]].k;
            T.n = T.off[T.nf];
            ok = T.n <= max_n;
        }
        if (ok) {
            std::vector<MfFrontDev> tfd;
            for (int32_t f : tf) tfd.push_back(fr[f]);
            const int64_t n_t = T.n;
            T.M = static_cast<double*>(be->alloc(n_t * n_t * sizeof(double)));
            m_bufs.push_back(T.M);
            const int R = T.nf - 1;  // the root's place in the block
            auto Fp = [&](int i) { return m_dev.front_store + tfd[i].off; };
            auto blockM = [&](int bi, int bj) { return T.M + (int64_t)T.off[bi] * n_t + T.off[bj]; };
            auto relp = [&](int i) { return m_dev.rel + tfd[i].rel_off; };
            // temporaries: P_c = E_c U_R^-1[rel_c, :] (k_c x k_R) and Q_c = L_R^-1[:, rel_c] G_c (k_R x k_c)
            const int kR = tfd[R].k;
            int64_t tmp_doubles = 0;
            std::vector<int64_t> p_off(T.nf), q_off(T.nf);
            for (int c = 0; c < R; ++c) {
                p_off[c] = tmp_doubles;
                tmp_doubles += (int64_t)tfd[c].k * kR;
                q_off[c] = tmp_doubles;
                tmp_doubles += (int64_t)tfd[c].k * kR;
            }
            double* tmp = static_cast<double*>(be->alloc(std::max<int64_t>(tmp_doubles, 1) * sizeof(double)));
            m_bufs.push_back(tmp);
            // a boundary that is the whole root in the root's own order needs no index map
            auto rel_map = [&](int c) -> const int32_t* {
                const int b = tfd[c].m - tfd[c].k;
                bool ident = b == kR;
                for (int j = 0; j < b && ident; ++j) ident = rel[tfd[c].rel_off + j] == j;
                return ident ? nullptr : relp(c);
            };
            std::vector<TopGemm> g;
            int dim[2] = {0, 0};
            bool indexed = false;
            auto push = [&](int stage, const TopGemm& t) {
                g.push_back(t);
                dim[stage] = std::max(dim[stage], std::max(t.M, t.N));
                indexed = indexed || t.A.cidx || t.B.ridx;
            };
            const double* UinvR = Fp(R) + (int64_t)kR * tfd[R].ld;  // F[A,P] = U11^-1 of the root
            const double* LinvR = Fp(R) + kR;                        // F[P,A] = L11^-1
            const int ldR = tfd[R].ld;
            // stage 0: U^-1 L^-1 of every front on its diagonal block of M; P_c and Q_c
            T.stage_begin[0] = 0;
            for (int i = 0; i < T.nf; ++i) {
                const int k = tfd[i].k, ld = tfd[i].ld;
                TopGemm t{};
                t.A = MatView{Fp(i) + (int64_t)k * ld, ld, k, k};
                t.B = MatView{Fp(i) + k, ld, k, k};
                t.C = blockM(i, i);
                t.ldc = n_t, t.M = k, t.N = k, t.K = k, t.acc = 0;
                push(0, t);
            }
            for (int c = 0; c < R; ++c) {
                const int k = tfd[c].k, b = tfd[c].m - k, ld = tfd[c].ld;
                TopGemm t{};
                t.A = MatView{Fp(c) + (int64_t)k * ld + 2 * k, ld, k, b};  // E_c = F[A,B] = -U11^-1 U12
                t.B = MatView{UinvR, ldR, b, kR, rel_map(c), nullptr};    // U_R^-1[rel_c, :]
                t.C = tmp + p_off[c];
                t.ldc = kR, t.M = k, t.N = kR, t.K = b, t.acc = 0;
                push(0, t);
                TopGemm u{};
                u.A = MatView{LinvR, ldR, kR, b, nullptr, rel_map(c)};            // L_R^-1[:, rel_c]
                u.B = MatView{Fp(c) + (int64_t)2 * k * ld + k, ld, b, k};          // G_c = F[B,A] = -L21 L11^-1
                u.C = tmp + q_off[c];
                u.ldc = k, u.M = kR, u.N = k, u.K = b, u.acc = 0;
                push(0, u);
            }
            // stage 1: M_cR = P_c L_R^-1, M_Rc = U_R^-1 Q_c, M_cc' = P_c Q_c' (+ the diagonal block of stage 0)
            T.stage_begin[1] = g.size();
            for (int c = 0; c < R; ++c) {
                const int k = tfd[c].k;
                TopGemm t{};
                t.A = MatView{tmp + p_off[c], kR, k, kR};
                t.B = MatView{LinvR, ldR, kR, kR};
                t.C = blockM(c, R);
                t.ldc = n_t, t.M = k, t.N = kR, t.K = kR, t.acc = 0;
                push(1, t);
                TopGemm u{};
                u.A = MatView{UinvR, ldR, kR, kR};
                u.B = MatView{tmp + q_off[c], k, kR, k};
                u.C = blockM(R, c);
                u.ldc = n_t, u.M = kR, u.N = k, u.K = kR, u.acc = 0;
                push(1, u);
                for (int c2 = 0; c2 < R; ++c2) {
                    const int k2 = tfd[c2].k;
                    TopGemm v{};
                    v.A = MatView{tmp + p_off[c], kR, k, kR};
                    v.B = MatView{tmp + q_off[c2], k2, kR, k2};
                    v.C = blockM(c, c2);
                    v.ldc = n_t, v.M = k, v.N = k2, v.K = kR, v.acc = c == c2;
                    push(1, v);
                }
            }
            T.stage_begin[2] = g.size();
            T.stage_dim[0] = dim[0];
            T.stage_dim[1] = dim[1];
            T.indexed = indexed;
            T.gemms = upload(g);
            // the right-hand side lists
            std::vector<std::vector<int32_t>> lists(T.n);
            std::vector<int32_t> wsrc(T.n);
            std::vector<char> in_top(F, 0);
            for (int32_t f : tf) in_top[f] = 1;
            for (int i = 0; i < T.nf; ++i) {
                const MfFrontDev& fd = tfd[i];
                for (int q = 0; q < fd.k; ++q) {
                    wsrc[T.off[i] + q] = fd.own_start + q;
                    for (int j = 0; j < fd.nch; ++j)
                        if (!in_top[children[tf[i]][j]])  // (the block's own fronts no longer write their parent's inbox)
                            lists[T.off[i] + q].push_back(fd.inbox_off + j * fd.m + q);
                }
            }
            for (int c = 0; c < R; ++c) {
                const MfFrontDev& fd = tfd[c];
                for (int j = 0; j < fd.m - fd.k; ++j)
                    for (int sl = 0; sl < fd.nch; ++sl)
                        lists[T.off[R] + rel[fd.rel_off + j]].push_back(fd.inbox_off + sl * fd.m + fd.k + j);
            }
            size_t W = 0;
            for (const auto& l : lists) W = std::max(W, l.size());
            W = (W + 1) / 2 * 2;
            if (W == 0) W = 2;
            if (W > 8) {
                T.enabled = false;  // (no kernel instance for lists this long)
            } else {
                std::vector<int32_t> ell(W * (size_t)T.n, (int32_t)inbox_doubles);
                for (int i = 0; i < T.n; ++i)
                    for (size_t sl = 0; sl < lists[i].size(); ++sl) ell[sl * T.n + i] = lists[i][sl];
                T.wsrc = upload(wsrc);
                T.ell = upload(ell);
                T.W = W;
                std::vector<int32_t> redirect(n);
                for (int64_t q = 0; q < n; ++q) redirect[q] = q;
                for (int i = 0; i < T.nf; ++i)
                    for (int q = 0; q < tfd[i].k; ++q) redirect[tfd[i].own_start + q] = n + T.off[i] + q;
                std::vector<int32_t> bnd_x(bnd_idx.size()), perm_x(n);
                for (size_t q = 0; q < bnd_idx.size(); ++q) bnd_x[q] = redirect[bnd_idx[q]];
                for (int64_t q = 0; q < n; ++q) perm_x[q] = redirect[perm[q]];
                T.bnd_x = upload(bnd_x);
                T.perm_x = upload(perm_x);
            }
            T.enabled = T.W > 0;
            if (std::getenv("SANM_MF_DEBUG"))
                std::fprintf(stderr, "mf top block: %d fronts, %d pivots, %zu products\n", T.nf, T.n, g.size());
        }
    }
    analysis_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ctor).count();
}

}  // namespace sanm_hip
