// Device-side reduction helpers shared by the ahead-of-time kernels (backend_hip.hip) and the pass kernels
// compiled at run time (graph.cpp: the rider workgroups of spec_pass4): wavefront reductions, the grid-wide
// deterministic reduction `grid_commit`, and the scaling phase of a Gram-Schmidt step.  HIP device code only.
#pragma once

namespace sanm_hip {
namespace {
__device__ __forceinline__ double wave_reduce_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_reduce_max(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}


// Grid-wide reduction of nv values whose result the HOST reads: every workgroup stores its partials,
// the last one to arrive (device-scope ticket) combines them in workgroup order -- deterministic, unlike
// an atomic accumulation -- and writes straight into pinned host memory, so a reduction costs one launch
// and one stream synchronisation (no accumulator memset, no read-back copy kernel).
constexpr unsigned RED_MAX_GRID = 512;  // one same-address atomic per workgroup (~12 ns each) bounds the useful grid
struct GridRed {
    double* partials;  // [MAX_RED][RED_MAX_GRID]
    unsigned* ticket;
    double* host;      // pinned, device-accessible
};
// (bid of nb: the workgroup's place among the workgroups that take part -- all of a launch, or the extra ones a
// launch carries for a deferred Gram-Schmidt phase, see GsRider)
template <int NV>
__device__ __forceinline__ void grid_commit_at(const double (&v)[NV], int nv, unsigned maxmask, GridRed g,
                                               unsigned bid, unsigned nb) {
    // A wavefront reduction is 6 cross-lane steps of ~100 cycles; with many values per thread (the Gram-Schmidt
    // projections: up to 24) they are spread over the 4 wavefronts through LDS instead of every wavefront
    // reducing every value (multi_dot_kernel: 17.6 -> see DESIGN.md for 20 vectors).
    constexpr bool kViaLds = NV > 4;
    __shared__ double sh[NV][4];
    __shared__ double stage[kViaLds ? NV : 1][kViaLds ? 256 : 1];
    __shared__ bool last;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // (workgroups of 256 threads everywhere but in the pass kernels compiled at run time, whose rider workgroups
    // have the 64 or 256 threads of the pass: only the small-NV path is taken there)
    const int nw = kViaLds ? 4 : (int)(blockDim.x >> 6);
    if constexpr (kViaLds) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (j < nv) stage[j][threadIdx.x] = v[j];
        __syncthreads();
        for (int j = w; j < nv; j += 4) {
            const bool mx = (maxmask >> j) & 1;
            const double a = stage[j][lane], b = stage[j][lane + 64], c = stage[j][lane + 128],
                         d = stage[j][lane + 192];
            const double r = mx ? wave_reduce_max(fmax(fmax(a, b), fmax(c, d))) : wave_reduce_sum((a + b) + (c + d));
            if (lane == 0) {
                sh[j][0] = r;
                sh[j][1] = sh[j][2] = sh[j][3] = mx ? -1e300 : 0.0;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (j < nv) {
                const double r = ((maxmask >> j) & 1) ? wave_reduce_max(v[j]) : wave_reduce_sum(v[j]);
                if (lane == 0) sh[j][w] = r;
            }
    }
    __syncthreads();
    // Hand-off without cache-wide fences (MI355X_MICROARCH.md, inter-workgroup visibility): the partials are
    // written through (agent-scope atomic stores), the storing wavefront drains them, one lane signals with
    // an agent-scope add behind the workgroup barrier, and the workgroup whose add came last reads them
    // with agent-scope loads.
    if ((int)threadIdx.x < nv) {
        const int j = threadIdx.x;
        const bool mx = (maxmask >> j) & 1;
        double r = sh[j][0];
        for (int i = 1; i < nw; ++i) r = mx ? fmax(r, sh[j][i]) : r + sh[j][i];
        __hip_atomic_store(&g.partials[j * RED_MAX_GRID + bid], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        last = __hip_atomic_fetch_add(g.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1;
    __syncthreads();
    if (!last) return;
    // one wavefront per value; the (up to KV) values of a wavefront are read in lock step, so that their
    // partials -- agent-scope loads that go all the way to memory -- share the round trips
    // (with fewer than 4 wavefronts only the values j = w + 4 q, w < nw, are combined: NV = 1 is the only
    // instantiation launched with 64 threads)
    constexpr int KV = (NV + 3) / 4;
    double r[KV];
#pragma unroll
    for (int q = 0; q < KV; ++q) r[q] = ((maxmask >> (w + 4 * q)) & 1) ? -1e300 : 0.0;
    for (unsigned b0 = 0; b0 < nb; b0 += 64) {
        const unsigned b = b0 + lane;
#pragma unroll
        for (int q = 0; q < KV; ++q) {
            const int j = w + 4 * q;
            if (j < nv && b < nb) {
                const double pv = __hip_atomic_load(&g.partials[j * RED_MAX_GRID + b], __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
                r[q] = ((maxmask >> j) & 1) ? fmax(r[q], pv) : r[q] + pv;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < KV; ++q) {
        const int j = w + 4 * q;
        if (j < nv) {
            const double t = ((maxmask >> j) & 1) ? wave_reduce_max(r[q]) : wave_reduce_sum(r[q]);
            if (lane == 0) g.host[j] = t;
        }
    }
    if (threadIdx.x == 0) *g.ticket = 0;  // launches on the stream are serialised
}
template <int NV>
__device__ __forceinline__ void grid_commit(const double (&v)[NV], int nv, unsigned maxmask, GridRed g) {
    grid_commit_at<NV>(v, nv, maxmask, g, blockIdx.x, gridDim.x);
}


// v *= 1 / max(sqrt(*norm2), eps), and the squared norm of the result for the (rare) second normalisation
__device__ __forceinline__ void scale_rsqrt_body(size_t n, double* v, const double* __restrict__ norm2, double eps,
                                                 GridRed g, unsigned bid, unsigned nb) {
    const double f = 1.0 / fmax(sqrt(*norm2), eps);
    double s[1] = {0};
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < n; i += (size_t)nb * blockDim.x) {
        const double w = v[i] * f;
        v[i] = w;
        s[0] += w * w;
    }
    grid_commit_at<1>(s, 1, 0u, g, bid, nb);
}
}  // namespace
}  // namespace sanm_hip
