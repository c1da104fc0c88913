// Tall-skinny fp32 GEMM of the shared MLPs with the BatchNorm statistics fused into its epilogue:
//     C[M,N] = A[M,K] * B[K,N]          M = B*npoint*nsample rows (up to ~1e6), N, K <= ~1300
//     stats (optional): per 128-row block, per column: sum and sum of squares of C  -> bn_finalize
// Forward pass:  Y = rows @ W^T   (B = W^T, K = Cin, N = Cout)  + column statistics of Y (saves a full read of Y)
// Input grad:    dA = dY @ W      (B = W,   K = Cout, N = Cin)
// Reference semantics: Conv2d 1x1 (bias=False) followed by BatchNorm2d in training mode
// (torch_points3d/core/common_modules/dense_modules.py:5-12,25-29).
//
// 4 waves per workgroup as 2x2, each wave 2x2 MFMA tiles of 32x32 (v_mfma_f32_32x32x2_f32, exact fp32):
// a 128 x 128 output tile per workgroup, K walked in steps of 32 through LDS with the global loads of step i+1
// in flight during the MFMAs of step i.  A is staged row-major with a 33-float pitch so that the MFMA operand
// fetch (32 consecutive ROWS at one k) is bank-conflict free; B rows are contiguous in n already.
#include "tp3d_common.h"

namespace tp3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GR_BLOCK_T = 256;
constexpr int GR_BM = 128, GR_BN = 128, GR_BK = 32;
constexpr int GR_LDA = GR_BK + 1;   // 33: odd pitch -> lanes (consecutive rows) hit distinct banks
constexpr int GR_LDB = GR_BN + 4;   // 132: keeps float4 stores aligned

template <bool STATS>
__global__ __launch_bounds__(GR_BLOCK_T) void gemm_rows_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                                int64_t M, int N, int K, int tiles_n,
                                                                float *__restrict__ C, float *__restrict__ partial)
{
    __shared__ float sA[GR_BM * GR_LDA];
    __shared__ __attribute__((aligned(16))) float sB[GR_BK * GR_LDB];
    __shared__ float s_st[2][2][GR_BN];  // [sum|sumsq][wave row][column]

    // XCD-aware order: the column tiles of one row block are 8 ids apart, i.e. on the same XCD (shared L2 for A)
    const int id = blockIdx.x;
    const int grp = id / (8 * tiles_n), rem = id % (8 * tiles_n);
    const int64_t rb = (int64_t)grp * 8 + (rem & 7);
    const int ct = rem >> 3;
    const int64_t m0 = rb * GR_BM;
    const int n0 = ct * GR_BN;
    if (m0 >= M) return;  // uniform for the workgroup (grid is rounded up to a multiple of 8 row blocks)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // staging registers: A tile 128 x 32 = 1024 float4 (4 per thread), B tile 32 x 128 = 1024 float4 (4 per thread)
    float4 ra[4], rbv[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * GR_BLOCK_T;
            const int row = e >> 3, k4 = (e & 7) * 4;  // 8 float4 per A row
            const int64_t m = m0 + row;
            ra[i] = (m < M && k0 + k4 < K) ? *reinterpret_cast<const float4 *>(A + m * K + k0 + k4)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * GR_BLOCK_T;
            const int kr = e >> 5, c4 = (e & 31) * 4;  // 32 float4 per B row
            rbv[i] = (k0 + kr < K && n0 + c4 < N) ? *reinterpret_cast<const float4 *>(B + (size_t)(k0 + kr) * N + n0 + c4)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += GR_BK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * GR_BLOCK_T;
            float *d = &sA[(e >> 3) * GR_LDA + (e & 7) * 4];
            d[0] = ra[i].x;
            d[1] = ra[i].y;
            d[2] = ra[i].z;
            d[3] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * GR_BLOCK_T;
            *reinterpret_cast<float4 *>(&sB[(e >> 5) * GR_LDB + (e & 31) * 4]) = rbv[i];
        }
        __syncthreads();
        if (k0 + GR_BK < K) fetch(k0 + GR_BK);
#pragma unroll
        for (int kk = 0; kk < GR_BK; kk += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = sA[((wr * 2 + i) * 32 + l31) * GR_LDA + kk + lh];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = sB[(kk + lh) * GR_LDB + (wc * 2 + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + (wc * 2 + j) * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + (wr * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (m < M && n < N) C[m * N + n] = acc[i][j][e];
            }
        }
    if (STATS) {
        // rows past M were staged as zeros, so they add nothing to either sum
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = acc[i][j][e];
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32);  // the two half-waves hold different rows of the same column
            s2 += __shfl_xor(s2, 32);
            if (lh == 0) {
                s_st[0][wr][(wc * 2 + j) * 32 + l31] = s1;
                s_st[1][wr][(wc * 2 + j) * 32 + l31] = s2;
            }
        }
        __syncthreads();
        if (tid < GR_BN && n0 + tid < N) {
            partial[((size_t)rb * 2 + 0) * N + n0 + tid] = s_st[0][0][tid] + s_st[0][1][tid];
            partial[((size_t)rb * 2 + 1) * N + n0 + tid] = s_st[1][0][tid] + s_st[1][1][tid];
        }
    }
}

}  // namespace tp3d

using namespace tp3d;

// number of 128-row blocks = number of statistic chunks the epilogue writes
TP3D_EXPORT size_t tp3d_gemm_rows_stat_floats(int64_t M, int N)
{
    if (M <= 0 || N <= 0) return 0;
    return (size_t)((M + GR_BM - 1) / GR_BM) * 2 * (size_t)N;
}

TP3D_EXPORT int tp3d_gemm_rows_f32(const float *A, const float *B, int64_t M, int N, int K, float *C,
                                   float *stat_partial, void *stream)
{
    if (M < 0 || N <= 0 || K <= 0 || (N & 3) || (K & 3)) return TP3D_E_BADARG;  // rows must be 16-byte aligned
    if (M == 0) return TP3D_OK;
    if (!A || !B || !C) return TP3D_E_BADARG;
    const int tiles_n = (N + GR_BN - 1) / GR_BN;
    const int64_t row_blocks = (M + GR_BM - 1) / GR_BM;
    const int64_t groups = (row_blocks + 7) / 8;
    const int64_t blocks = groups * 8 * tiles_n;
    if (blocks > 0x7fffffff) return TP3D_E_TOOBIG;
    hipStream_t s = (hipStream_t)stream;
    if (stat_partial)
        hipLaunchKernelGGL(gemm_rows_kernel<true>, dim3((unsigned)blocks), dim3(GR_BLOCK_T), 0, s, A, B, M, N, K, tiles_n,
                           C, stat_partial);
    else
        hipLaunchKernelGGL(gemm_rows_kernel<false>, dim3((unsigned)blocks), dim3(GR_BLOCK_T), 0, s, A, B, M, N, K, tiles_n,
                           C, stat_partial);
    return check_launch();
}
