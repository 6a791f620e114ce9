// Per-kernel time of a chain of dependent small kernels: stream launches against a hipGraph replay of the same chain.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_chain scripts/micro/graph_chain.hip && /tmp/graph_chain
// The level launches of the direct solver's sweeps (16 per solve, 20 solves per ANM step, fixed arguments) are such
// a chain; this measures what a graph replay would save on them.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256) work(const double* __restrict__ in, double* __restrict__ out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 1.0000001 + 1e-9;
}
int main() {
    const int n = 256 * 1024;  // 1024 workgroups, 2 MB in + 2 MB out: about the size of an upper tree level
    double *a, *b;
    CK(hipMalloc(&a, n * 8));
    CK(hipMalloc(&b, n * 8));
    CK(hipMemset(a, 0, n * 8));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int chain : {16, 64, 340}) {
        for (int grid : {8, 1024}) {
            const int nn = grid * 256;
            auto launch_chain = [&]() {
                for (int k = 0; k < chain; ++k)
                    hipLaunchKernelGGL(work, dim3(grid), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, nn);
            };
            // stream
            for (int w = 0; w < 5; ++w) launch_chain();
            CK(hipStreamSynchronize(s));
            const int reps = 40;
            auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) launch_chain();
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("chain %3d grid %4d  stream: %.2f us per kernel (device), %.2f us wall\n", chain, grid,
                   ms * 1e3 / (reps * chain), wall * 1e6 / (reps * chain));
            // graph
            hipGraph_t g;
            hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            launch_chain();
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("chain %3d grid %4d  graph : %.2f us per kernel (device), %.2f us wall\n", chain, grid,
                   ms * 1e3 / (reps * chain), wall * 1e6 / (reps * chain));
            CK(hipGraphExecDestroy(ge));
            CK(hipGraphDestroy(g));
        }
    }
    return 0;
}
