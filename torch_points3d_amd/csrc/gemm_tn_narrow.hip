// Weight gradient of the FIRST layer of a shared MLP on grouped rows -- a handful of input channels (relative position +
// features: 6 -> 8 padded) against 64 output channels over a million rows:
//   dW[n][k] = sum_r dY[r][n] * A[r][k],  K <= 16,
//   dY = scale * ((dA * act'(z) - c1) - (Y - mean) * c2),  z = (Y - mean) * scale + beta      (BatchNorm + LeakyReLU backward)
// formed on the fly from the layer's pre-BatchNorm output Y and the gradient dA of its activated output
// (core/common_modules/dense_modules.py:25-29, autograd backward).
//
// 2 * M * N * K flops is nothing (1 GFLOP at 1 M x 64 x 8) while Y and dA are 268 MB each: the MFMA tile kernel pads K to
// its 64-column tile and spends 100 us on zeros, after a separate pass wrote dY (another 268 MB out and in).  Here: four
// output columns per lane (float4 loads of Y and dA; a wave covers 64 / (N/4) rows per load), their 4 x K partial sums in
// registers, the rows of Y and dA read exactly once, coalesced, four row groups in flight per wave.  Row groups of a wave
// are folded by shuffles, the four waves through LDS, the row splits by tn_reduce_splits -- a fixed order: reproducible.
#include "tp3d_common.h"

namespace tp3d {

constexpr int TNN_BLOCK = 256, TNN_WAVES = 4, TNN_U = 4;  // row groups in flight per wave
constexpr int TNN_KMAX = 16, TNN_NMAX = 256;

// (four waves per SIMD asked for: left to itself the scheduler chases occupancy, sinks every load next to its use behind a
// full wait -- one request in flight per wave -- and the kernel runs at a third of the memory rate)
template <int KQ>  // K = 4 * KQ
__global__ __launch_bounds__(TNN_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_tn_narrow_bn_kernel(
    const float *__restrict__ Y, const float *__restrict__ dA, const float *__restrict__ A, const float *__restrict__ mean,
    const float *__restrict__ scale, const float *__restrict__ beta, const float *__restrict__ c1,
    const float *__restrict__ c2, float slope, int64_t M, int N, int nv_shift, int64_t rows_per_split,
    float *__restrict__ partial, int reverse)
{
    constexpr int K = 4 * KQ;
    __shared__ float red[TNN_WAVES - 1][TNN_NMAX * K];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int NV = 1 << nv_shift;        // float4 columns of a row (N = 4 * NV, a power of two)
    const int RW = 64 >> nv_shift;       // rows one wave covers per load
    const int cq = lane & (NV - 1), rsub = lane >> nv_shift;
    const float4 mu = *reinterpret_cast<const float4 *>(mean + 4 * cq), sc = *reinterpret_cast<const float4 *>(scale + 4 * cq);
    const float4 be = *reinterpret_cast<const float4 *>(beta + 4 * cq), k1 = *reinterpret_cast<const float4 *>(c1 + 4 * cq);
    const float4 k2 = *reinterpret_cast<const float4 *>(c2 + 4 * cq);
    const int split = reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;  // (last rows first: include/tp3d_hip.h)
    const int64_t r0 = (int64_t)split * rows_per_split;
    const int64_t r1 = min(r0 + rows_per_split, M);
    float acc[4][K];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[c][k] = 0.0f;
    auto one = [&](float y, float d, float m_, float s_, float b_, float a1, float a2, bool in) __attribute__((always_inline)) -> float {
        const float yc = y - m_;
        const float z = yc * s_ + b_;
        const float dz = d * (z > 0.0f ? 1.0f : slope);
        const float dy = s_ * ((dz - a1) - yc * a2);
        return in ? dy : 0.0f;
    };
    const int step = TNN_WAVES * RW;  // rows the workgroup covers per load round
    for (int64_t r = r0 + w * RW + rsub; r < r1; r += (int64_t)TNN_U * step) {
        float4 y[TNN_U], d[TNN_U], a[TNN_U][KQ];
#pragma unroll
        for (int u = 0; u < TNN_U; ++u) {
            const int64_t rr = min(r + (int64_t)u * step, r1 - 1);  // a valid row; masked below
            y[u] = *reinterpret_cast<const float4 *>(Y + rr * N + 4 * cq);
            d[u] = *reinterpret_cast<const float4 *>(dA + rr * N + 4 * cq);
#pragma unroll
            for (int q = 0; q < KQ; ++q) a[u][q] = *reinterpret_cast<const float4 *>(A + rr * K + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < TNN_U; ++u) {
            const bool in = r + (int64_t)u * step < r1;
            const float dy[4] = {one(y[u].x, d[u].x, mu.x, sc.x, be.x, k1.x, k2.x, in), one(y[u].y, d[u].y, mu.y, sc.y, be.y, k1.y, k2.y, in),
                                 one(y[u].z, d[u].z, mu.z, sc.z, be.z, k1.z, k2.z, in), one(y[u].w, d[u].w, mu.w, sc.w, be.w, k1.w, k2.w, in)};
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int q = 0; q < KQ; ++q) {
                    acc[c][4 * q + 0] = __builtin_fmaf(dy[c], a[u][q].x, acc[c][4 * q + 0]);
                    acc[c][4 * q + 1] = __builtin_fmaf(dy[c], a[u][q].y, acc[c][4 * q + 1]);
                    acc[c][4 * q + 2] = __builtin_fmaf(dy[c], a[u][q].z, acc[c][4 * q + 2]);
                    acc[c][4 * q + 3] = __builtin_fmaf(dy[c], a[u][q].w, acc[c][4 * q + 3]);
                }
        }
    }
    // the wave's row groups (lanes cq, cq + NV, ...) into lanes 0 .. NV-1
    for (int off = 32; off >= NV; off >>= 1)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < K; ++k) acc[c][k] += __shfl_down(acc[c][k], off);
    if (w > 0 && rsub == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < K; ++k) red[w - 1][(4 * cq + c) * K + k] = acc[c][k];
    }
    __syncthreads();
    if (w == 0 && rsub == 0) {
        float *out = partial + ((size_t)split * N + 4 * cq) * K;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[c][4 * q + j];
#pragma unroll
                    for (int x = 0; x < TNN_WAVES - 1; ++x) t += red[x][(4 * cq + c) * K + 4 * q + j];
                    v[j] = t;
                }
                *reinterpret_cast<float4 *>(out + c * K + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
            }
    }
}

// The forward contraction of the same layer, Y[r][n] = sum_k A[r][k] W[n][k] (K <= 16), with the shifted BatchNorm sums of
// its output: the MFMA tile kernel pads K to its 32-wide step and runs this 1 : 8 write-dominated stream at 3 TB/s
// (4194304 x 64 x 8: 403 us for 1.2 GB).  Here: the lane's four output columns' weights in registers (4 x K), the A row
// of its row by float4 loads (the lanes of a row share the address), 4 K FMAs in ascending k, one float4 store; four row
// groups in flight per wave.  Statistics: one chunk per workgroup in tp3d_bn_finalize_f32's layout -- sum d, sum d^2,
// shift (the output of the workgroup's first row), rows -- lanes folded by shuffles, waves through LDS, fixed order.
template <int KQ, bool STATS>
__global__ __launch_bounds__(TNN_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_rows_narrow_kernel(
    const float *__restrict__ A, const float *__restrict__ W, int64_t M, int N, int nv_shift, int64_t rows_per_split,
    float *__restrict__ Y, float *__restrict__ partial, int reverse)
{
    constexpr int K = 4 * KQ;
    constexpr int U = KQ >= 3 ? 2 : TNN_U;  // (the 4 x K weights leave room for two rows' operands at K = 12, 16)
    __shared__ float red[TNN_WAVES - 1][2][TNN_NMAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int NV = 1 << nv_shift, RW = 64 >> nv_shift;
    const int cq = lane & (NV - 1), rsub = lane >> nv_shift;
    const int split = reverse ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    const int64_t r0 = (int64_t)split * rows_per_split;
    const int64_t r1 = min(r0 + rows_per_split, M);
    float4 wq[4][KQ];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < KQ; ++q) wq[c][q] = *reinterpret_cast<const float4 *>(W + (size_t)(4 * cq + c) * K + 4 * q);
    auto row_out = [&](const float4 (&a)[KQ]) __attribute__((always_inline)) -> float4 {
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float t = 0.0f;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                t = __builtin_fmaf(a[q].x, wq[c][q].x, t);
                t = __builtin_fmaf(a[q].y, wq[c][q].y, t);
                t = __builtin_fmaf(a[q].z, wq[c][q].z, t);
                t = __builtin_fmaf(a[q].w, wq[c][q].w, t);
            }
            o[c] = t;
        }
        return make_float4(o[0], o[1], o[2], o[3]);
    };
    float4 shift = make_float4(0.f, 0.f, 0.f, 0.f), s1 = shift, s2 = shift;
    if (STATS) {  // the shift of this chunk: the output of its first row (every lane forms it for its own columns)
        float4 a[KQ];
#pragma unroll
        for (int q = 0; q < KQ; ++q) a[q] = *reinterpret_cast<const float4 *>(A + r0 * K + 4 * q);
        shift = row_out(a);
    }
    const int step = TNN_WAVES * RW;
    for (int64_t r = r0 + w * RW + rsub; r < r1; r += (int64_t)U * step) {
        float4 a[U][KQ];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = min(r + (int64_t)u * step, r1 - 1);
#pragma unroll
            for (int q = 0; q < KQ; ++q) a[u][q] = *reinterpret_cast<const float4 *>(A + rr * K + 4 * q);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t rr = r + (int64_t)u * step;
            if (rr < r1) {
                const float4 y = row_out(a[u]);
                *reinterpret_cast<float4 *>(Y + rr * N + 4 * cq) = y;
                if (STATS) {
                    const float dx = y.x - shift.x, dy = y.y - shift.y, dz = y.z - shift.z, dw = y.w - shift.w;
                    s1.x += dx, s1.y += dy, s1.z += dz, s1.w += dw;
                    s2.x += dx * dx, s2.y += dy * dy, s2.z += dz * dz, s2.w += dw * dw;
                }
            }
        }
    }
    if (!STATS) return;
    for (int off = 32; off >= NV; off >>= 1) {
        s1.x += __shfl_down(s1.x, off), s1.y += __shfl_down(s1.y, off), s1.z += __shfl_down(s1.z, off), s1.w += __shfl_down(s1.w, off);
        s2.x += __shfl_down(s2.x, off), s2.y += __shfl_down(s2.y, off), s2.z += __shfl_down(s2.z, off), s2.w += __shfl_down(s2.w, off);
    }
    if (w > 0 && rsub == 0) {
        *reinterpret_cast<float4 *>(&red[w - 1][0][4 * cq]) = s1;
        *reinterpret_cast<float4 *>(&red[w - 1][1][4 * cq]) = s2;
    }
    __syncthreads();
    if (w == 0 && rsub == 0) {
#pragma unroll
        for (int x = 0; x < TNN_WAVES - 1; ++x) {
            const float4 t1 = *reinterpret_cast<const float4 *>(&red[x][0][4 * cq]);
            const float4 t2 = *reinterpret_cast<const float4 *>(&red[x][1][4 * cq]);
            s1.x += t1.x, s1.y += t1.y, s1.z += t1.z, s1.w += t1.w;
            s2.x += t2.x, s2.y += t2.y, s2.z += t2.z, s2.w += t2.w;
        }
        float *pr = partial + (size_t)split * 4 * N + 4 * cq;
        const float rows = (float)(r1 - r0);
        *reinterpret_cast<float4 *>(pr) = s1;
        *reinterpret_cast<float4 *>(pr + N) = s2;
        *reinterpret_cast<float4 *>(pr + 2 * (size_t)N) = shift;
        *reinterpret_cast<float4 *>(pr + 3 * (size_t)N) = make_float4(rows, rows, rows, rows);
    }
}

static int64_t tnn_rows_per_split(int64_t M)
{
    int64_t rps = (M + 1023) / 1024;  // at most 1024 splits (all resident at once: 4 workgroups per CU), at least 256 rows each
    if (rps < 256) rps = 256;
    return (rps + 15) / 16 * 16;
}

}  // namespace tp3d

using namespace tp3d;

// 1: (M, N, K) is this kernel's case -- a contraction of at most 16 channels (whole float4 quads) into 4, 8, 16 ... 256
// output channels (a power of two: the lanes of a wave tile whole rows)
TP3D_EXPORT int tp3d_gemm_tn_bn_narrow_serves(int64_t M, int N, int K)
{
    return (M > 0 && N >= 4 && N <= TNN_NMAX && (N & (N - 1)) == 0 && K >= 4 && K <= TNN_KMAX && (K & 3) == 0) ? 1 : 0;
}

TP3D_EXPORT size_t tp3d_gemm_tn_bn_narrow_workspace_floats(int64_t M, int N, int K)
{
    if (!tp3d_gemm_tn_bn_narrow_serves(M, N, K)) return 0;
    const int64_t rps = tnn_rows_per_split(M);
    return (size_t)((M + rps - 1) / rps) * (size_t)N * (size_t)K;
}

// Y, dA (M,N); mean_n, scale_n, beta_n, c1_n, c2_n (N) -- c1 / c2 from tp3d_bn_bwd_reduce_f32; A (M,K) plain rows ->
// out (N,K).  workspace: tp3d_gemm_tn_bn_narrow_workspace_floats.
TP3D_EXPORT int tp3d_gemm_tn_bn_narrow_f32(const float *Y, const float *dA, const float *mean_n, const float *scale_n,
                                           const float *beta_n, const float *c1_n, const float *c2_n, float slope_n,
                                           const float *A, int64_t M, int N, int K, float *out, float *workspace,
                                           int reverse, void *stream)
{
    if (!tp3d_gemm_tn_bn_narrow_serves(M, N, K) || !out) return TP3D_E_BADARG;
    if (!Y || !dA || !mean_n || !scale_n || !beta_n || !c1_n || !c2_n || !A || !workspace) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rps = tnn_rows_per_split(M);
    const int splits = (int)((M + rps - 1) / rps);
    dim3 grid(splits);
    int nv_shift = 0;
    while ((4 << nv_shift) < N) ++nv_shift;
#define TP3D_TNN(KQ_)                                                                                                  \
    hipLaunchKernelGGL((gemm_tn_narrow_bn_kernel<KQ_>), grid, dim3(TNN_BLOCK), 0, s, Y, dA, A, mean_n, scale_n, beta_n, c1_n, \
                       c2_n, slope_n, M, N, nv_shift, rps, workspace, reverse)
    switch (K / 4) {
    case 1: TP3D_TNN(1); break;
    case 2: TP3D_TNN(2); break;
    case 3: TP3D_TNN(3); break;
    default: TP3D_TNN(4); break;
    }
#undef TP3D_TNN
    if (int rc = check_launch()) return rc;
    return tn_reduce_splits(workspace, splits, (int64_t)N * K, out, s);
}

// Forward form: Y (M,N) = A (M,K) W (N,K)^T for the same shapes (tp3d_gemm_tn_bn_narrow_serves); with stat_partial != NULL
// also tp3d_gemm_rows_narrow_chunks(M) statistics chunks of [4][N] floats in tp3d_bn_finalize_f32's layout.
TP3D_EXPORT int tp3d_gemm_rows_narrow_chunks(int64_t M)
{
    if (M <= 0) return 0;
    const int64_t rps = tnn_rows_per_split(M);
    return (int)((M + rps - 1) / rps);
}

TP3D_EXPORT int tp3d_gemm_rows_narrow_f32(const float *A, const float *W, int64_t M, int N, int K, float *Y,
                                          float *stat_partial, int reverse, void *stream)
{
    if (!tp3d_gemm_tn_bn_narrow_serves(M, N, K) || !A || !W || !Y) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rps = tnn_rows_per_split(M);
    const int splits = (int)((M + rps - 1) / rps);
    int nv_shift = 0;
    while ((4 << nv_shift) < N) ++nv_shift;
#define TP3D_RN(KQ_)                                                                                                   \
    do {                                                                                                               \
        if (stat_partial)                                                                                              \
            hipLaunchKernelGGL((gemm_rows_narrow_kernel<KQ_, true>), dim3(splits), dim3(TNN_BLOCK), 0, s, A, W, M, N, nv_shift, rps, \
                               Y, stat_partial, reverse);                                                              \
        else                                                                                                           \
            hipLaunchKernelGGL((gemm_rows_narrow_kernel<KQ_, false>), dim3(splits), dim3(TNN_BLOCK), 0, s, A, W, M, N, nv_shift, rps, \
                               Y, stat_partial, reverse);                                                              \
    } while (0)
    switch (K / 4) {
    case 1: TP3D_RN(1); break;
    case 2: TP3D_RN(2); break;
    case 3: TP3D_RN(3); break;
    default: TP3D_RN(4); break;
    }
#undef TP3D_RN
    return check_launch();
}
