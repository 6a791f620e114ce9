// Contiguous ranges of [0, n) on a few host threads.  The solver's construction (tet order, remap tables, Jacobian
// pattern, analysis) is part of the reference's time_solve (fea/main.cpp:382: the clock starts before the solver
// is constructed), so its loops over the rows of a big mesh are worth the threads.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <system_error>
#include <thread>
#include <vector>

namespace sanm_hip {
inline int host_thread_cap() {
    // (SANM_MF_ND_THREADS is the dissection's own budget, multifrontal.cpp: it no longer caps every setup loop)
    const char* env_thr = std::getenv("SANM_HOST_THREADS");
    if (env_thr) return std::min(64, std::max(1, std::atoi(env_thr)));
    // the host's threads are shared by the ranks of a node (torchrun's LOCAL_WORLD_SIZE, else WORLD_SIZE)
    unsigned ranks = 1;
    for (const char* name : {"LOCAL_WORLD_SIZE", "WORLD_SIZE"})
        if (const char* v = std::getenv(name)) {
            ranks = (unsigned)std::max(1, std::atoi(v));
            break;
        }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (int)std::min(16u, std::max(1u, hw / ranks));
}

//! threads that are joined when the scope ends, whatever ends it (a std::thread that is still joinable when it is
//! destroyed terminates the process: a throwing emplace_back -- thread limits of a container -- or a throwing piece of
//! work on the calling thread would otherwise do that inside a library constructor)
struct JoinedThreads {
    std::vector<std::thread> th;
    ~JoinedThreads() {
        for (auto& x : th)
            if (x.joinable()) x.join();
    }
    //! start fn on a thread of its own; when the system refuses one, run it here
    template <class F>
    void run(F&& fn) {
        try {
            th.emplace_back(fn);
        } catch (const std::system_error&) {
            fn();
        }
    }
};

//! fn(begin, end, thread) over nt <= cap ranges of at least min_per_thread elements; fn must not throw
template <class F>
void parallel_ranges(int64_t n, int64_t min_per_thread, F&& fn) {
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_thread_cap(), n / std::max<int64_t>(min_per_thread, 1)));
    if (nt <= 1) {
        fn(0, n, 0);
        return;
    }
    JoinedThreads jt;
    for (int t = 1; t < nt; ++t) jt.run([&, t] { fn(n * t / nt, n * (t + 1) / nt, t); });
    fn(0, n / nt, 0);
}

//! std::sort by pieces on the threads of parallel_ranges, merged pairwise.  cmp must be a strict TOTAL order (break
//! ties by index): then the result is the sequential sort's whatever the number of pieces.
template <class It, class Cmp>
void parallel_sort(It first, It last, Cmp cmp, int64_t min_per_thread = 4096) {
    const int64_t n = last - first;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_thread_cap(), n / std::max<int64_t>(min_per_thread, 1)));
    if (nt <= 1) {
        std::sort(first, last, cmp);
        return;
    }
    std::vector<int64_t> cut(nt + 1);
    for (int t = 0; t <= nt; ++t) cut[t] = n * t / nt;
    {
        JoinedThreads jt;
        for (int t = 1; t < nt; ++t) jt.run([&, t] { std::sort(first + cut[t], first + cut[t + 1], cmp); });
        std::sort(first + cut[0], first + cut[1], cmp);
    }
    while (cut.size() > 2) {
        const int64_t pairs = (int64_t)(cut.size() - 1) / 2;
        {
            JoinedThreads jt;
            for (int64_t p = 1; p < pairs; ++p)
                jt.run([&, p] { std::inplace_merge(first + cut[2 * p], first + cut[2 * p + 1], first + cut[2 * p + 2], cmp); });
            std::inplace_merge(first + cut[0], first + cut[1], first + cut[2], cmp);
        }
        std::vector<int64_t> next;
        for (size_t i = 0; i < cut.size(); i += 2) next.push_back(cut[i]);
        if (next.back() != n) next.push_back(n);
        cut.swap(next);
    }
}

//! n elements left uninitialised (a std::vector would zero them on one thread before the workers fill them)
template <class T>
std::unique_ptr<T[]> raw_array(size_t n) {
    return std::unique_ptr<T[]>(new T[std::max<size_t>(n, 1)]);
}

//! wall-clock laps of a setup phase on stderr when SANM_DEBUG_SETUP is set
class SetupLaps {
    const char* m_phase;
    bool m_on;
    std::chrono::steady_clock::time_point m_t;

public:
    explicit SetupLaps(const char* phase)
            : m_phase{phase}, m_on{std::getenv("SANM_DEBUG_SETUP") != nullptr}, m_t{std::chrono::steady_clock::now()} {}
    void lap(const char* what) {
        if (!m_on) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[setup] %s.%s %.4f\n", m_phase, what, std::chrono::duration<double>(t - m_t).count());
        m_t = t;
    }
};
}  // namespace sanm_hip
