// Contiguous ranges of [0, n) on a few host threads.  The solver's construction (tet order, remap tables, Jacobian
// pattern, analysis) is part of the reference's time_solve (fea/main.cpp:382: the clock starts before the solver
// is constructed), so its loops over the rows of a big mesh are worth the threads.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <thread>
#include <vector>

namespace sanm_hip {
inline int host_thread_cap() {
    const char* env_thr = std::getenv("SANM_HOST_THREADS");
    if (!env_thr) env_thr = std::getenv("SANM_MF_ND_THREADS");  // the name of round 5's first version
    if (env_thr) return std::max(1, std::atoi(env_thr));
    return (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
}

//! fn(begin, end, thread) over nt <= cap ranges of at least min_per_thread elements; fn must not throw
template <class F>
void parallel_ranges(int64_t n, int64_t min_per_thread, F&& fn) {
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_thread_cap(), n / std::max<int64_t>(min_per_thread, 1)));
    if (nt <= 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back([&, t] { fn(n * t / nt, n * (t + 1) / nt, t); });
    fn(0, n / nt, 0);
    for (auto& x : th) x.join();
}
}  // namespace sanm_hip
