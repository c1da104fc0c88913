// Y (M, N) = A (M, K; row stride lda) * W (N, K)^T for the edge-wise MLPs of RandLA-Net: millions of rows, K and N <= 32
// (torch_points3d/modules/RandLANet/modules.py:20-22: point_pos_nn [10, 8, F], attention_nn [2F, 8, 2F]).
// Library GEMMs run these through 16x256 macro-tiles (measured 166 us for 4 M rows of 12 -> 8 on MI355X); the work is a
// pure stream -- 4 * M * (K + N) bytes -- so: one row per lane, the row in registers, the N*K weights broadcast from
// LDS, four outputs per step.  fp32 FMA order: k ascending per output (fixed, independent of the launch shape).
#include "tp3d_common.h"

namespace tp3d {

constexpr int SK_MAX = 32;

// EPI: the output goes through a per-column affine + LeakyReLU before it is stored -- Linear -> BatchNorm (running
// statistics) -> activation of an eval-mode edge MLP in ONE pass over the rows: out = act((y - mean[n]) * scale[n] + beta[n])
template <int KV, bool VEC, bool EPI>  // KV = ceil(K / 4) register quads per row
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const float *__restrict__ A, const float *__restrict__ W,
                                                          int64_t M, int N, int K, int lda, float *__restrict__ Y,
                                                          const float *__restrict__ mean, const float *__restrict__ scale,
                                                          const float *__restrict__ beta, float slope)
{
    __shared__ __attribute__((aligned(16))) float sw[SK_MAX * SK_MAX];  // [n][KV*4], zero padded
    __shared__ float se[EPI ? 3 * SK_MAX : 1];
    for (int i = threadIdx.x; i < N * KV * 4; i += 256) {
        const int n = i / (KV * 4), k = i % (KV * 4);
        sw[i] = k < K ? W[n * K + k] : 0.0f;
    }
    if (EPI && threadIdx.x < N) {
        se[threadIdx.x] = mean[threadIdx.x];
        se[SK_MAX + threadIdx.x] = scale[threadIdx.x];
        se[2 * SK_MAX + threadIdx.x] = beta[threadIdx.x];
    }
    __syncthreads();
    auto finish = [&](float y, int n) __attribute__((always_inline)) -> float {
        if (!EPI) return y;
        const float z = (y - se[n]) * se[SK_MAX + n] + se[2 * SK_MAX + n];
        return z > 0.0f ? z : z * slope;
    };
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < M; r += stride) {
        float a[KV * 4];
        const float *row = A + r * lda;
        if (VEC) {
#pragma unroll
            for (int v = 0; v < KV; ++v) {
                // lda % 4 == 0 and lda >= KV*4 (checked by the host): whole quads, padding columns hit zero weights
                const float4 q = *reinterpret_cast<const float4 *>(row + v * 4);
                a[v * 4 + 0] = q.x, a[v * 4 + 1] = q.y, a[v * 4 + 2] = q.z, a[v * 4 + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < KV * 4; ++k) a[k] = k < K ? row[k] : 0.0f;
        }
        float *out = Y + r * N;
        int n = 0;
        if ((N & 3) == 0) {
            for (; n < N; n += 4) {
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < KV * 4; ++k) acc[j] = __builtin_fmaf(a[k], sw[(n + j) * KV * 4 + k], acc[j]);
                *reinterpret_cast<float4 *>(out + n) =
                    make_float4(finish(acc[0], n), finish(acc[1], n + 1), finish(acc[2], n + 2), finish(acc[3], n + 3));
            }
        } else {
            for (; n < N; ++n) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < KV * 4; ++k) acc = __builtin_fmaf(a[k], sw[n * KV * 4 + k], acc);
                out[n] = finish(acc, n);
            }
        }
    }
}

}  // namespace tp3d

using namespace tp3d;

namespace {
int launch_skinny(const float *A, const float *W, int64_t M, int N, int K, int lda, float *Y, const float *mean,
                  const float *scale, const float *beta, float slope, hipStream_t s)
{
    if (M < 0 || N <= 0 || K <= 0 || lda < K) return TP3D_E_BADARG;
    if (N > SK_MAX || K > SK_MAX) return TP3D_E_TOOBIG;
    if (M == 0) return TP3D_OK;
    if (!A || !W || !Y) return TP3D_E_BADARG;
    const int KV = (K + 3) / 4;
    const bool vec = (lda & 3) == 0 && lda >= KV * 4 && ((uintptr_t)A & 15) == 0;
    int64_t blocks = (M + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 workgroups per CU
    dim3 grid((unsigned)blocks);
#define TP3D_SK_ONE(KV_, VEC_, EPI_)                                                                                  \
    hipLaunchKernelGGL((gemm_skinny_kernel<KV_, VEC_, EPI_>), grid, dim3(256), 0, s, A, W, M, N, K, lda, Y, mean, scale, \
                       beta, slope)
#define TP3D_SK(KV_)                                                                                                 \
    case KV_:                                                                                                        \
        if (mean) {                                                                                                  \
            if (vec) TP3D_SK_ONE(KV_, true, true);                                                                   \
            else TP3D_SK_ONE(KV_, false, true);                                                                      \
        } else {                                                                                                     \
            if (vec) TP3D_SK_ONE(KV_, true, false);                                                                  \
            else TP3D_SK_ONE(KV_, false, false);                                                                     \
        }                                                                                                            \
        break;
    switch (KV) {
        TP3D_SK(1) TP3D_SK(2) TP3D_SK(3) TP3D_SK(4) TP3D_SK(5) TP3D_SK(6) TP3D_SK(7) TP3D_SK(8)
    default: return TP3D_E_TOOBIG;
    }
#undef TP3D_SK
#undef TP3D_SK_ONE
    return check_launch();
}
}  // namespace

TP3D_EXPORT int tp3d_gemm_skinny_f32(const float *A, const float *W, int64_t M, int N, int K, int lda, float *Y,
                                     void *stream)
{
    return launch_skinny(A, W, M, N, K, lda, Y, nullptr, nullptr, nullptr, 1.0f, (hipStream_t)stream);
}

// out = LeakyReLU_slope((A W^T - mean) * scale + beta): Linear -> BatchNorm (given statistics, i.e. eval mode) ->
// activation of an edge MLP layer in one pass over its rows
TP3D_EXPORT int tp3d_gemm_skinny_bnact_f32(const float *A, const float *W, int64_t M, int N, int K, int lda,
                                           const float *mean, const float *scale, const float *beta, float slope,
                                           float *out, void *stream)
{
    if (!mean || !scale || !beta) return TP3D_E_BADARG;
    return launch_skinny(A, W, M, N, K, lda, out, mean, scale, beta, slope, (hipStream_t)stream);
}
