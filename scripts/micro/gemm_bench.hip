// fp64-MFMA GEMM tile variants for the multifrontal factorisation (mf_kernels.h: gemm_tile), standalone:
//   C (M x N, row-major, ld) -= A (M x K) * B (K x N)
// at the shapes the factorisation produces: long K (Schur complement of a big front: gemm2) and K = 128 (rank-128
// trailing update of the two-level panel scheme: block_gemm).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-mllvm -amdgpu-mfma-vgpr-form] -o gemm_bench gemm_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double mfma_f64x4 __attribute__((ext_vector_type(4)));
struct MatView {
    const double* p;
    int ld, rows, cols;
};
constexpr int GT = 64, GK = 16;

// ---------------------------------------------------------------- V0: as in mf_kernels.h (round 2) -----------
namespace v0 {
struct GemmStage { double a[4], b[4]; };
__device__ __forceinline__ void stage_load(GemmStage& st, const MatView& A, const MatView& B, int ti, int tj, int kk, int k1) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 256 * s;
        const int ar = idx / GK, ae = idx % GK;
        int gr = ti * GT + ar, gc = kk + ae;
        st.a[s] = (gr < A.rows && gc < A.cols && gc < k1) ? A.p[(int64_t)gr * A.ld + gc] : 0.0;
        const int be = idx / GT, bc = idx % GT;
        gr = kk + be;
        gc = tj * GT + bc;
        st.b[s] = (gr < B.rows && gr < k1 && gc < B.cols) ? B.p[(int64_t)gr * B.ld + gc] : 0.0;
    }
}
__device__ __forceinline__ void stage_store(const GemmStage& st, double (*As)[GT + 1], double (*Bs)[GT + 4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 256 * s;
        As[idx % GK][idx / GK] = st.a[s];
        Bs[idx / GT][idx % GT] = st.b[s];
    }
}
__global__ void __launch_bounds__(256) kernel(MatView A, MatView B, double* C, int ldc, int K) {
    __shared__ double As[GK][GT + 1], Bs[GK][GT + 4];
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = 32 * (wv >> 1) + (lane & 15), c0 = 32 * (wv & 1) + (lane & 15), kq = lane >> 4;
    mfma_f64x4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f64x4{0, 0, 0, 0};
    GemmStage st;
    stage_load(st, A, B, ti, tj, 0, K);
    for (int kk = 0; kk < K; kk += GK) {
        __syncthreads();
        stage_store(st, As, Bs);
        __syncthreads();
        if (kk + GK < K) stage_load(st, A, B, ti, tj, kk + GK, K);
#pragma unroll
        for (int e = 0; e < GK; e += 4) {
            const double a0 = As[e + kq][r0], a1 = As[e + kq][r0 + 16];
            const double b0 = Bs[e + kq][c0], b1 = Bs[e + kq][c0 + 16];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int g = 0; g < 4; ++g) {
        const int r = ti * GT + 32 * (wv >> 1) + 16 * mi + (lane >> 4) + 4 * g, c = tj * GT + 32 * (wv & 1) + 16 * ni + (lane & 15);
        if (r < A.rows && c < B.cols) C[(int64_t)r * ldc + c] -= acc[mi][ni][g];
    }
}
}  // namespace v0

// ---------------------------------------------------------------- V1: branch-free staging ---------------------
// loads with clamped indices and a select instead of a guarded load (no exec-mask branches in the K loop: the
// accumulators can stay where the matrix cores write them), 16-byte loads along the contiguous direction
namespace v1 {
template <int KS>
struct Stage { double2 a[KS / 8], b[KS / 8]; };  // 64 x KS doubles of A and KS x 64 of B over 256 threads, 2 at a time
template <int KS>
__device__ __forceinline__ void stage_load(Stage<KS>& st, const MatView& A, const MatView& B, int ti, int tj, int kk, int k1) {
    const int tid = threadIdx.x;
    constexpr int APR = KS / 2;  // double2 per row of the A tile
#pragma unroll
    for (int s = 0; s < KS / 8; ++s) {
        const int idx = tid + 256 * s;
        const int ar = idx / APR, ae = (idx % APR) * 2;
        const int gr = ti * GT + ar, gc = kk + ae;
        const bool ok0 = gr < A.rows && gc < k1, ok1 = gr < A.rows && gc + 1 < k1;
        const int cr = min(gr, A.rows - 1), cc = min(gc, A.cols - 2);
        const double2 v = *reinterpret_cast<const double2*>(A.p + (int64_t)cr * A.ld + (cc & ~1));
        st.a[s] = double2{ok0 ? v.x : 0.0, ok1 ? v.y : 0.0};
        const int be = idx / 32, bc = (idx % 32) * 2;
        const int br = kk + be, bcol = tj * GT + bc;
        const bool okb0 = br < k1 && bcol < B.cols, okb1 = br < k1 && bcol + 1 < B.cols;
        const int crb = min(br, B.rows - 1), ccb = min(bcol, B.cols - 2);
        const double2 w = *reinterpret_cast<const double2*>(B.p + (int64_t)crb * B.ld + (ccb & ~1));
        st.b[s] = double2{okb0 ? w.x : 0.0, okb1 ? w.y : 0.0};
    }
}
template <int KS>
__device__ __forceinline__ void stage_store(const Stage<KS>& st, double (*As)[GT + 1], double (*Bs)[GT + 4]) {
    const int tid = threadIdx.x;
    constexpr int APR = KS / 2;
#pragma unroll
    for (int s = 0; s < KS / 8; ++s) {
        const int idx = tid + 256 * s;
        const int ar = idx / APR, ae = (idx % APR) * 2;
        As[ae][ar] = st.a[s].x;
        As[ae + 1][ar] = st.a[s].y;
        const int be = idx / 32, bc = (idx % 32) * 2;
        *reinterpret_cast<double2*>(&Bs[be][bc]) = st.b[s];
    }
}
template <int KS>
__global__ void __launch_bounds__(256) kernel(MatView A, MatView B, double* C, int ldc, int K) {
    __shared__ double As[KS][GT + 1], Bs[KS][GT + 4];
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = 32 * (wv >> 1) + (lane & 15), c0 = 32 * (wv & 1) + (lane & 15), kq = lane >> 4;
    mfma_f64x4 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f64x4{0, 0, 0, 0};
    Stage<KS> st;
    stage_load<KS>(st, A, B, ti, tj, 0, K);
    for (int kk = 0; kk < K; kk += KS) {
        __syncthreads();
        stage_store<KS>(st, As, Bs);
        __syncthreads();
        stage_load<KS>(st, A, B, ti, tj, min(kk + KS, K), K);  // (past the end: all zero, clamped addresses)
#pragma unroll
        for (int e = 0; e < KS; e += 4) {
            const double a0 = As[e + kq][r0], a1 = As[e + kq][r0 + 16];
            const double b0 = Bs[e + kq][c0], b1 = Bs[e + kq][c0 + 16];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) for (int g = 0; g < 4; ++g) {
        const int r = ti * GT + 32 * (wv >> 1) + 16 * mi + (lane >> 4) + 4 * g, c = tj * GT + 32 * (wv & 1) + 16 * ni + (lane & 15);
        if (r < A.rows && c < B.cols) C[(int64_t)r * ldc + c] -= acc[mi][ni][g];
    }
}
}  // namespace v1


// ---------------------------------------------------------------- V2: 128 x 64 tile, each wave 64 x 32 ---------
namespace v2 {
constexpr int TM = 128, TN = 64, KS = 16;
struct Stage { double2 a[4], b[2]; };
__device__ __forceinline__ void stage_load(Stage& st, const MatView& A, const MatView& B, int ti, int tj, int kk, int k1) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {  // A tile 128 x 16: 1024 double2
        const int idx = tid + 256 * s;
        const int ar = idx / 8, ae = (idx % 8) * 2;
        const int gr = ti * TM + ar, gc = kk + ae;
        const bool ok0 = gr < A.rows && gc < k1, ok1 = gr < A.rows && gc + 1 < k1;
        const int cr = min(gr, A.rows - 1), cc = min(gc, A.cols - 2);
        const double2 v = *reinterpret_cast<const double2*>(A.p + (int64_t)cr * A.ld + (cc & ~1));
        st.a[s] = double2{ok0 ? v.x : 0.0, ok1 ? v.y : 0.0};
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // B tile 16 x 64: 512 double2
        const int idx = tid + 256 * s;
        const int be = idx / 32, bc = (idx % 32) * 2;
        const int br = kk + be, bcol = tj * TN + bc;
        const bool okb0 = br < k1 && bcol < B.cols, okb1 = br < k1 && bcol + 1 < B.cols;
        const int crb = min(br, B.rows - 1), ccb = min(bcol, B.cols - 2);
        const double2 w = *reinterpret_cast<const double2*>(B.p + (int64_t)crb * B.ld + (ccb & ~1));
        st.b[s] = double2{okb0 ? w.x : 0.0, okb1 ? w.y : 0.0};
    }
}
__device__ __forceinline__ void stage_store(const Stage& st, double (*As)[TM + 1], double (*Bs)[TN + 4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 256 * s;
        const int ar = idx / 8, ae = (idx % 8) * 2;
        As[ae][ar] = st.a[s].x;
        As[ae + 1][ar] = st.a[s].y;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int idx = tid + 256 * s;
        const int be = idx / 32, bc = (idx % 32) * 2;
        *reinterpret_cast<double2*>(&Bs[be][bc]) = st.b[s];
    }
}
__global__ void __launch_bounds__(256) kernel(MatView A, MatView B, double* C, int ldc, int K) {
    __shared__ double As[KS][TM + 1], Bs[KS][TN + 4];
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = 64 * (wv >> 1) + (lane & 15), c0 = 32 * (wv & 1) + (lane & 15), kq = lane >> 4;
    mfma_f64x4 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f64x4{0, 0, 0, 0};
    Stage st;
    stage_load(st, A, B, ti, tj, 0, K);
    for (int kk = 0; kk < K; kk += KS) {
        __syncthreads();
        stage_store(st, As, Bs);
        __syncthreads();
        stage_load(st, A, B, ti, tj, min(kk + KS, K), K);
#pragma unroll
        for (int e = 0; e < KS; e += 4) {
            double a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[e + kq][r0 + 16 * i];
            b[0] = Bs[e + kq][c0];
            b[1] = Bs[e + kq][c0 + 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 2; ++ni) for (int g = 0; g < 4; ++g) {
        const int r = ti * TM + 64 * (wv >> 1) + 16 * mi + (lane >> 4) + 4 * g, c = tj * TN + 32 * (wv & 1) + 16 * ni + (lane & 15);
        if (r < A.rows && c < B.cols) C[(int64_t)r * ldc + c] -= acc[mi][ni][g];
    }
}
}  // namespace v2

// ---------------------------------------------------------------- V3: 128 x 128 tile, each wave 64 x 64 --------
namespace v3 {
constexpr int TM = 128, TN = 128, KS = 16;
struct Stage { double2 a[4], b[4]; };
__device__ __forceinline__ void stage_load(Stage& st, const MatView& A, const MatView& B, int ti, int tj, int kk, int k1) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 256 * s;
        const int ar = idx / 8, ae = (idx % 8) * 2;
        const int gr = ti * TM + ar, gc = kk + ae;
        const bool ok0 = gr < A.rows && gc < k1, ok1 = gr < A.rows && gc + 1 < k1;
        const int cr = min(gr, A.rows - 1), cc = min(gc, A.cols - 2);
        const double2 v = *reinterpret_cast<const double2*>(A.p + (int64_t)cr * A.ld + (cc & ~1));
        st.a[s] = double2{ok0 ? v.x : 0.0, ok1 ? v.y : 0.0};
        const int be = idx / 64, bc = (idx % 64) * 2;
        const int br = kk + be, bcol = tj * TN + bc;
        const bool okb0 = br < k1 && bcol < B.cols, okb1 = br < k1 && bcol + 1 < B.cols;
        const int crb = min(br, B.rows - 1), ccb = min(bcol, B.cols - 2);
        const double2 w = *reinterpret_cast<const double2*>(B.p + (int64_t)crb * B.ld + (ccb & ~1));
        st.b[s] = double2{okb0 ? w.x : 0.0, okb1 ? w.y : 0.0};
    }
}
__device__ __forceinline__ void stage_store(const Stage& st, double (*As)[TM + 1], double (*Bs)[TN + 4]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 256 * s;
        const int ar = idx / 8, ae = (idx % 8) * 2;
        As[ae][ar] = st.a[s].x;
        As[ae + 1][ar] = st.a[s].y;
        const int be = idx / 64, bc = (idx % 64) * 2;
        *reinterpret_cast<double2*>(&Bs[be][bc]) = st.b[s];
    }
}
__global__ void __launch_bounds__(256) kernel(MatView A, MatView B, double* C, int ldc, int K) {
    __shared__ double As[KS][TM + 1], Bs[KS][TN + 4];
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = 64 * (wv >> 1) + (lane & 15), c0 = 64 * (wv & 1) + (lane & 15), kq = lane >> 4;
    mfma_f64x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = mfma_f64x4{0, 0, 0, 0};
    Stage st;
    stage_load(st, A, B, ti, tj, 0, K);
    for (int kk = 0; kk < K; kk += KS) {
        __syncthreads();
        stage_store(st, As, Bs);
        __syncthreads();
        stage_load(st, A, B, ti, tj, min(kk + KS, K), K);
#pragma unroll
        for (int e = 0; e < KS; e += 4) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[e + kq][r0 + 16 * i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[e + kq][c0 + 16 * j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 4; ++ni) for (int g = 0; g < 4; ++g) {
        const int r = ti * TM + 64 * (wv >> 1) + 16 * mi + (lane >> 4) + 4 * g, c = tj * TN + 64 * (wv & 1) + 16 * ni + (lane & 15);
        if (r < A.rows && c < B.cols) C[(int64_t)r * ldc + c] -= acc[mi][ni][g];
    }
}
}  // namespace v3

// ---------------------------------------------------------------- V4: 128 x 64 tile, K step KS (16 / 32 / 64) -----
// half / a quarter of the barriers per flop; interior tiles only (no guards), as the product's fast path (= KS 16).
// Round 4, 8192^2, K = 128 / 1024 / 4096: KS 16 41.7 / 67.1 / 69.4 TFLOP/s, KS 32 32.3 / 60.6 / 64.1 (fewer workgroups
// per CU), double-buffered LDS with one barrier per step 41.7 / 64.4 / 65.9 (KS 32: 20.0 / 42.6 / 48.3): the fast path
// as shipped is the best of them; 128 x 128 tiles 28.7 / 61.1 / 65.6.
namespace v4 {
constexpr int TM = 128, TN = 64;
typedef double2 __attribute__((aligned(8))) double2_u;
template <int KS>
struct Stage { double2 a[KS / 4], b[KS / 8]; };
template <int KS>
__device__ __forceinline__ void stage_load(Stage<KS>& st, const MatView& A, const MatView& B, int ti, int tj, int kk) {
    const int tid = threadIdx.x;
    constexpr int APR = KS / 2;  // double2 per row of the A tile
#pragma unroll
    for (int s = 0; s < KS / 4; ++s) {  // A tile 128 x KS
        const int idx = tid + 256 * s;
        const int ar = idx / APR, ae = (idx % APR) * 2;
        st.a[s] = *reinterpret_cast<const double2_u*>(A.p + (int64_t)(ti * TM + ar) * A.ld + kk + ae);
    }
#pragma unroll
    for (int s = 0; s < KS / 8; ++s) {  // B tile KS x 64
        const int idx = tid + 256 * s;
        const int be = idx / 32, bc = (idx % 32) * 2;
        st.b[s] = *reinterpret_cast<const double2_u*>(B.p + (int64_t)(kk + be) * B.ld + tj * TN + bc);
    }
}
template <int KS>
__device__ __forceinline__ void stage_store(const Stage<KS>& st, double (*As)[TM + 1], double (*Bs)[TN + 4]) {
    const int tid = threadIdx.x;
    constexpr int APR = KS / 2;
#pragma unroll
    for (int s = 0; s < KS / 4; ++s) {
        const int idx = tid + 256 * s;
        const int ar = idx / APR, ae = (idx % APR) * 2;
        As[ae][ar] = st.a[s].x;
        As[ae + 1][ar] = st.a[s].y;
    }
#pragma unroll
    for (int s = 0; s < KS / 8; ++s) {
        const int idx = tid + 256 * s;
        const int be = idx / 32, bc = (idx % 32) * 2;
        *reinterpret_cast<double2*>(&Bs[be][bc]) = st.b[s];
    }
}
template <int KS>
__global__ void __launch_bounds__(256) kernel(MatView A, MatView B, double* C, int ldc, int K) {
    __shared__ double As[KS][TM + 1], Bs[KS][TN + 4];
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = 64 * (wv >> 1) + (lane & 15), c0 = 32 * (wv & 1) + (lane & 15), kq = lane >> 4;
    mfma_f64x4 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f64x4{0, 0, 0, 0};
    Stage<KS> st;
    stage_load<KS>(st, A, B, ti, tj, 0);
    for (int kk = 0; kk < K; kk += KS) {
        __syncthreads();
        stage_store<KS>(st, As, Bs);
        __syncthreads();
        if (kk + KS < K) stage_load<KS>(st, A, B, ti, tj, kk + KS);
#pragma unroll
        for (int e = 0; e < KS; e += 4) {
            double a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[e + kq][r0 + 16 * i];
            b[0] = Bs[e + kq][c0];
            b[1] = Bs[e + kq][c0 + 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 2; ++ni) for (int g = 0; g < 4; ++g) {
        const int r = ti * TM + 64 * (wv >> 1) + 16 * mi + (lane >> 4) + 4 * g, c = tj * TN + 32 * (wv & 1) + 16 * ni + (lane & 15);
        C[(int64_t)r * ldc + c] -= acc[mi][ni][g];
    }
}
// double-buffered LDS: one barrier per K step, the next step's tile stored while this one's MFMAs run
template <int KS>
__global__ void __launch_bounds__(256) kernel_db(MatView A, MatView B, double* C, int ldc, int K) {
    __shared__ double As[2][KS][TM + 1], Bs[2][KS][TN + 4];
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = 64 * (wv >> 1) + (lane & 15), c0 = 32 * (wv & 1) + (lane & 15), kq = lane >> 4;
    mfma_f64x4 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = mfma_f64x4{0, 0, 0, 0};
    Stage<KS> st;
    stage_load<KS>(st, A, B, ti, tj, 0);
    stage_store<KS>(st, As[0], Bs[0]);
    if (KS < K) stage_load<KS>(st, A, B, ti, tj, KS);
    int cur = 0;
    for (int kk = 0; kk < K; kk += KS) {
        __syncthreads();  // buffer `cur` is complete; buffer `cur ^ 1` has been read by everyone (previous step)
        if (kk + KS < K) stage_store<KS>(st, As[cur ^ 1], Bs[cur ^ 1]);
        if (kk + 2 * KS < K) stage_load<KS>(st, A, B, ti, tj, kk + 2 * KS);
#pragma unroll
        for (int e = 0; e < KS; e += 4) {
            double a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[cur][e + kq][r0 + 16 * i];
            b[0] = Bs[cur][e + kq][c0];
            b[1] = Bs[cur][e + kq][c0 + 16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        cur ^= 1;
    }
    for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 2; ++ni) for (int g = 0; g < 4; ++g) {
        const int r = ti * TM + 64 * (wv >> 1) + 16 * mi + (lane >> 4) + 4 * g, c = tj * TN + 32 * (wv & 1) + 16 * ni + (lane & 15);
        C[(int64_t)r * ldc + c] -= acc[mi][ni][g];
    }
}
}  // namespace v4

template <class L>
double bench(L&& launch, int reps, double flops) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return flops * reps / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 8192, N = M;
    if (argc > 5) {
        // the product's operand layout: gemm_bench M K lda ldb ldc  (tmpL: lda = K; tmpU: ldb = N; C inside a front:
        // ldc = N + 2 K) -- the 128 x 64 fast path only
        const int K = atoi(argv[2]), lda = atoi(argv[3]), ldb = atoi(argv[4]), ldc = atoi(argv[5]);
        double *dA, *dB, *dC;
        CK(hipMalloc(&dA, (size_t)M * lda * 8));
        CK(hipMalloc(&dB, (size_t)K * ldb * 8));
        CK(hipMalloc(&dC, (size_t)M * ldc * 8));
        CK(hipMemset(dA, 0, (size_t)M * lda * 8));
        CK(hipMemset(dB, 0, (size_t)K * ldb * 8));
        CK(hipMemset(dC, 0, (size_t)M * ldc * 8));
        MatView A{dA, lda, M, K}, B{dB, ldb, K, N};
        const dim3 grid2(N / 64, M / 128);
        printf("M=N=%d K=%d lda=%d ldb=%d ldc=%d: 128x64 fast path %.1f TFLOP/s\n", M, K, lda, ldb, ldc,
               bench([&] { hipLaunchKernelGGL(v4::kernel<16>, grid2, dim3(256), 0, 0, A, B, dC, ldc, K); }, 5, 2.0 * M * N * K));
        return 0;
    }
    for (int K : {128, 1024, 4096}) {
        const int ld = 8192 + 192;  // like a front: rows far apart
        std::vector<double> hA((size_t)M * K), hB((size_t)K * N), hC((size_t)M * N, 1.0);
        for (size_t i = 0; i < hA.size(); ++i) hA[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
        for (size_t i = 0; i < hB.size(); ++i) hB[i] = ((i * 40503u) % 1000) / 1000.0 - 0.5;
        double *dA, *dB, *dC, *dC2;
        CK(hipMalloc(&dA, (size_t)M * ld * 8));
        CK(hipMalloc(&dB, (size_t)K * ld * 8));
        CK(hipMalloc(&dC, (size_t)M * ld * 8));
        CK(hipMalloc(&dC2, (size_t)M * ld * 8));
        CK(hipMemcpy2D(dA, (size_t)ld * 8, hA.data(), (size_t)K * 8, (size_t)K * 8, M, hipMemcpyHostToDevice));
        CK(hipMemcpy2D(dB, (size_t)ld * 8, hB.data(), (size_t)N * 8, (size_t)N * 8, K, hipMemcpyHostToDevice));
        MatView A{dA, ld, M, K}, B{dB, ld, K, N};
        const dim3 grid(N / GT, M / GT);
        const double flops = 2.0 * M * N * K;
        auto reset = [&](double* c) { hipMemset(c, 0, (size_t)M * ld * 8); };
        // correctness of every variant against V0 on one launch
        reset(dC);
        hipLaunchKernelGGL(v0::kernel, grid, dim3(256), 0, 0, A, B, dC, ld, K);
        auto check = [&](const char* name) {
            std::vector<double> a((size_t)1024), b((size_t)1024);
            double md = 0;
            for (int r : {0, 63, 64, 4097, M - 1}) {
                hipMemcpy(a.data(), dC + (size_t)r * ld, 1024 * 8, hipMemcpyDeviceToHost);
                hipMemcpy(b.data(), dC2 + (size_t)r * ld, 1024 * 8, hipMemcpyDeviceToHost);
                for (int i = 0; i < 1024; ++i) md = fmax(md, fabs(a[i] - b[i]));
            }
            printf("   %s max |diff| vs v0 = %.3g\n", name, md);
        };
        printf("M=N=%d K=%d\n", M, K);
        const dim3 grid2(N / 64, M / 128);
        reset(dC2); hipLaunchKernelGGL(v1::kernel<16>, grid, dim3(256), 0, 0, A, B, dC2, ld, K); check("v1<16>");
        reset(dC2); hipLaunchKernelGGL(v1::kernel<32>, grid, dim3(256), 0, 0, A, B, dC2, ld, K); check("v1<32>");
        reset(dC2); hipLaunchKernelGGL(v2::kernel, grid2, dim3(256), 0, 0, A, B, dC2, ld, K); check("v2");
        const dim3 grid3(N / 128, M / 128);
        reset(dC2); hipLaunchKernelGGL(v3::kernel, grid3, dim3(256), 0, 0, A, B, dC2, ld, K); check("v3");
        reset(dC2); hipLaunchKernelGGL(v4::kernel<16>, grid2, dim3(256), 0, 0, A, B, dC2, ld, K); check("v4<16>");
        reset(dC2); hipLaunchKernelGGL(v4::kernel<32>, grid2, dim3(256), 0, 0, A, B, dC2, ld, K); check("v4<32>");
        reset(dC2); hipLaunchKernelGGL(v4::kernel_db<16>, grid2, dim3(256), 0, 0, A, B, dC2, ld, K); check("v4db<16>");
        reset(dC2); hipLaunchKernelGGL(v4::kernel_db<32>, grid2, dim3(256), 0, 0, A, B, dC2, ld, K); check("v4db<32>");
        printf("  v0 (round 2)            %.1f TFLOP/s\n", bench([&] { hipLaunchKernelGGL(v0::kernel, grid, dim3(256), 0, 0, A, B, dC, ld, K); }, 5, flops));
        printf("  v1 branch-free, KS=16   %.1f TFLOP/s\n", bench([&] { hipLaunchKernelGGL(v1::kernel<16>, grid, dim3(256), 0, 0, A, B, dC2, ld, K); }, 5, flops));
        printf("  v1 branch-free, KS=32   %.1f TFLOP/s\n", bench([&] { hipLaunchKernelGGL(v1::kernel<32>, grid, dim3(256), 0, 0, A, B, dC2, ld, K); }, 5, flops));
        printf("  v3 128x128 tile         %.1f TFLOP/s\n", bench([&] { hipLaunchKernelGGL(v3::kernel, grid3, dim3(256), 0, 0, A, B, dC2, ld, K); }, 5, flops));
        printf("  v2 128x64 tile          %.1f TFLOP/s\n", bench([&] { hipLaunchKernelGGL(v2::kernel, grid2, dim3(256), 0, 0, A, B, dC2, ld, K); }, 5, flops));
        printf("  v4 128x64 unguarded KS=16   %.1f TFLOP/s\n", bench([&] { hipLaunchKernelGGL(v4::kernel<16>, grid2, dim3(256), 0, 0, A, B, dC2, ld, K); }, 5, flops));
        printf("  v4 128x64 unguarded KS=32   %.1f TFLOP/s\n", bench([&] { hipLaunchKernelGGL(v4::kernel<32>, grid2, dim3(256), 0, 0, A, B, dC2, ld, K); }, 5, flops));
        printf("  v4 128x64 dbl-buf   KS=16   %.1f TFLOP/s\n", bench([&] { hipLaunchKernelGGL(v4::kernel_db<16>, grid2, dim3(256), 0, 0, A, B, dC2, ld, K); }, 5, flops));
        printf("  v4 128x64 dbl-buf   KS=32   %.1f TFLOP/s\n", bench([&] { hipLaunchKernelGGL(v4::kernel_db<32>, grid2, dim3(256), 0, 0, A, B, dC2, ld, K); }, 5, flops));
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dC2);
    }
    return 0;
}
