// Forward contraction of a hidden layer of a shared MLP on the bf16 matrix pipe:
//
//   C[M,N] = leaky((Y - mean) * scale + beta)[M,K] * Bt[N,K]^T
//
// the split-role form of gemm_rows_sp.hip (previous layer's BatchNorm + LeakyReLU in the loader waves, statistics chunks in
// the epilogue, optional activated side output) with every fp32 operand value split EXACTLY into three bf16 terms
// (x3_split.h) and the product taken as the six term pairs of weight >= 2^-15, each a v_mfma_f32_32x32x16_bf16 with fp32
// accumulation, smallest pairs first (the contraction is at most 512 deep: one accumulator per block, unlike the
// million-row sums of gemm_tn_x3.hip).  24 MFMAs of 32 cycles per 16-deep K-step instead of 32 of 64 for the fp32
// MFMA: the layer leaves the matrix pipe's critical path (524288 x 128 x 128: 207 -> 162 us in round 2's probe).  Two 8-wave
// workgroups per CU (72 KB of planes + the constants).
// Reference semantics: Conv2d 1x1 (bias=False) of MLP2D, core/common_modules/dense_modules.py:5-12,25-29.
#include <hip/hip_runtime.h>

#include "tp3d_common.h"
#include "x3_split.h"

namespace tp3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int B3_BLOCK = 512;
constexpr int B3_BM = 128, B3_BN = 128;
constexpr int B3_BK = 16;                     // 16-deep K-steps: 72 KB of planes, two workgroups per CU
#define B3_PITCH (B3_BK + 8)                  /* halfwords per tile row: the k's + 8 pad (16-byte aligned rows) */
#define B3_PLANE (B3_BM * B3_PITCH)           /* halfwords per plane */
#define B3_LDS_BYTES (2 /*buffers*/ * 2 /*operands*/ * 3 /*planes*/ * B3_PLANE * 2)

__device__ __forceinline__ float4 b3_keep(bool c, float4 v)
{
    return make_float4(c ? v.x : 0.0f, c ? v.y : 0.0f, c ? v.z : 0.0f, c ? v.w : 0.0f);
}

// STATS: 0 none, 2 one statistics chunk per (workgroup, wave row) -- the layout of gemm_rows_sp.hip
template <int STATS>
__global__ __launch_bounds__(B3_BLOCK, 4) void gemm_rows_x3_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                                                                int64_t M, int N, int K, int tiles_n, int64_t items,
                                                                float *__restrict__ C, float *__restrict__ partial,
                                                                float *__restrict__ act_out, const float *__restrict__ mean,
                                                                const float *__restrict__ scale, const float *__restrict__ beta,
                                                                float slope, int reverse)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    // the prologue's per-channel constants (K <= 512)
    __shared__ __attribute__((aligned(16))) float sK[3 * 512];
    constexpr bool pro = true;
    {
        for (int k = threadIdx.x; k < 512; k += B3_BLOCK) {
            sK[k] = k < K ? mean[k] : 0.0f;
            sK[512 + k] = k < K ? scale[k] : 0.0f;
            sK[1024 + k] = k < K ? beta[k] : 0.0f;
        }
        __syncthreads();
    }
    // plane p of operand o (0 = A, 1 = B) in buffer u: smem + ((u * 2 + o) * 3 + p) * B3_PLANE
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksteps = (K + B3_BK - 1) / B3_BK;
    if ((int64_t)blockIdx.x >= items) return;
    const int64_t my_items = (items - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t total = my_items * ksteps;

    const int64_t last_grp = items / (8 * tiles_n) - 1;
    auto decode = [&](int64_t item, int64_t &m0, int &n0) {
        int64_t grp = item / (8 * tiles_n);
        const int rem = (int)(item % (8 * tiles_n));
        if (reverse) grp = max(last_grp - grp, (int64_t)0);  // row blocks last to first (the loaders' look-ahead past the
                                                             // last item lands on group 0: valid addresses, never consumed)
        m0 = (grp * 8 + (rem & 7)) * B3_BM;
        n0 = (rem >> 3) * B3_BN;
    };

    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int lt = tid - 256;
        constexpr int TPR = B3_BK / 4, RPP = 256 / TPR;  // threads per tile row, rows per pass (32 or 64)
        constexpr int NS = 128 / RPP;                    // float4 slots per thread and operand (4 or 2)
        const int frow = lt / TPR, fk4 = (lt % TPR) * 4;
        int64_t f_item = blockIdx.x, f_m0;
        int f_n0, f_ks = 0;
        decode(f_item, f_m0, f_n0);
        auto act4 = [&](const float4 raw, int kk) __attribute__((always_inline)) -> float4 {
            if (!pro) return raw;
            const int kc = min(kk, 512 - 4);
            const float4 mu = *reinterpret_cast<const float4 *>(&sK[kc]);
            const float4 sc = *reinterpret_cast<const float4 *>(&sK[512 + kc]);
            const float4 be = *reinterpret_cast<const float4 *>(&sK[1024 + kc]);
            const float z0 = (raw.x - mu.x) * sc.x + be.x, z1 = (raw.y - mu.y) * sc.y + be.y;
            const float z2 = (raw.z - mu.z) * sc.z + be.z, z3 = (raw.w - mu.w) * sc.w + be.w;
            return make_float4(z0 > 0.0f ? z0 : z0 * slope, z1 > 0.0f ? z1 : z1 * slope, z2 > 0.0f ? z2 : z2 * slope,
                               z3 > 0.0f ? z3 : z3 * slope);
        };
#define B3_FETCH(S)                                                                                                   \
    do {                                                                                                              \
        m0_##S = f_m0, n0_##S = f_n0, k0_##S = f_ks * B3_BK;                                                          \
        const int kk = min(k0_##S + fk4, K - 4);                                                                      \
        a0_##S = *reinterpret_cast<const float4 *>(A + min(f_m0 + frow + 0, M - 1) * K + kk);                         \
        a1_##S = *reinterpret_cast<const float4 *>(A + min(f_m0 + frow + RPP, M - 1) * K + kk);                       \
        b0_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 0, N - 1) * K + kk);                \
        b1_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + RPP, N - 1) * K + kk);              \
        if constexpr (NS == 4) {                                                                                      \
            a2_##S = *reinterpret_cast<const float4 *>(A + min(f_m0 + frow + 2 * RPP, M - 1) * K + kk);               \
            a3_##S = *reinterpret_cast<const float4 *>(A + min(f_m0 + frow + 3 * RPP, M - 1) * K + kk);               \
            b2_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 2 * RPP, N - 1) * K + kk);      \
            b3_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 3 * RPP, N - 1) * K + kk);      \
        }                                                                                                             \
        if (++f_ks == ksteps) {                                                                                       \
            f_ks = 0;                                                                                                 \
            f_item += gridDim.x;                                                                                      \
            decode(f_item, f_m0, f_n0);                                                                               \
        }                                                                                                             \
    } while (0)
#define B3_PUT(BUF, O, I, V, IN)                                                                                      \
    do {                                                                                                              \
        uint2 hi, mid, lo;                                                                                            \
        const float4 kept = b3_keep(IN, V);                                                                           \
        if (O == 0 && side && (IN))                                                                                   \
            *reinterpret_cast<float4 *>(act_out + (m0_cur + frow + RPP * I) * K + k0_cur + fk4) = kept;               \
        x3_split(kept, hi, mid, lo);                                                                                  \
        unsigned short *d = smem + ((BUF * 2 + O) * 3) * B3_PLANE + (frow + RPP * I) * B3_PITCH + fk4;                 \
        *reinterpret_cast<uint2 *>(d) = hi;                                                                           \
        *reinterpret_cast<uint2 *>(d + B3_PLANE) = mid;                                                               \
        *reinterpret_cast<uint2 *>(d + 2 * B3_PLANE) = lo;                                                            \
    } while (0)
#define B3_STASH(S, BUF)                                                                                              \
    do {                                                                                                              \
        const bool kin = k0_##S + fk4 < K;                                                                            \
        const bool side = act_out != nullptr && n0_##S == 0; /* one column tile of a row block writes the side rows */ \
        const int64_t m0_cur = m0_##S;                                                                                \
        const int k0_cur = k0_##S;                                                                                    \
        B3_PUT(BUF, 0, 0, act4(a0_##S, k0_##S + fk4), kin && m0_##S + frow + 0 < M);                                  \
        B3_PUT(BUF, 0, 1, act4(a1_##S, k0_##S + fk4), kin && m0_##S + frow + RPP < M);                                \
        B3_PUT(BUF, 1, 0, b0_##S, kin && n0_##S + frow + 0 < N);                                                      \
        B3_PUT(BUF, 1, 1, b1_##S, kin && n0_##S + frow + RPP < N);                                                    \
        if constexpr (NS == 4) {                                                                                      \
            B3_PUT(BUF, 0, 2, act4(a2_##S, k0_##S + fk4), kin && m0_##S + frow + 2 * RPP < M);                        \
            B3_PUT(BUF, 0, 3, act4(a3_##S, k0_##S + fk4), kin && m0_##S + frow + 3 * RPP < M);                        \
            B3_PUT(BUF, 1, 2, b2_##S, kin && n0_##S + frow + 2 * RPP < N);                                            \
            B3_PUT(BUF, 1, 3, b3_##S, kin && n0_##S + frow + 3 * RPP < N);                                            \
        }                                                                                                             \
    } while (0)
        float4 a0_0, a1_0, a2_0, a3_0, b0_0, b1_0, b2_0, b3_0, a0_1, a1_1, a2_1, a3_1, b0_1, b1_1, b2_1, b3_1;
        int64_t m0_0, m0_1;
        int n0_0, n0_1, k0_0, k0_1;
        B3_FETCH(0);
        B3_FETCH(1);
        B3_STASH(0, 0);
        B3_FETCH(0);
        __syncthreads();
#pragma unroll 1
        for (int64_t s = 0; s < total; s += 2) {
            B3_STASH(1, 1);
            B3_FETCH(1);
            __syncthreads();
            if (s + 1 >= total) break;
            B3_STASH(0, 0);
            B3_FETCH(0);
            __syncthreads();
        }
#undef B3_FETCH
#undef B3_PUT
#undef B3_STASH
        return;
    }

    // ---------------------------------------------------------------------- MFMA waves
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    int64_t item = blockIdx.x, m0;
    int n0, ks = 0;
    decode(item, m0, n0);
    const int n0_first = n0;
    float run1[2] = {0.0f, 0.0f}, run2[2] = {0.0f, 0.0f}, kshift[2] = {0.0f, 0.0f};
    bool have_shift = false;
    int run_rows = 0;
    f32x16 acc[2][2];  // one accumulator per block: the contraction is K <= 512 deep, the running sum is rounded K/16 * 6 times
    __syncthreads();
    for (int64_t s = 0; s < total; ++s) {
        const unsigned short *pa = smem + (((s & 1) * 2 + 0) * 3) * B3_PLANE;
        const unsigned short *pb = smem + (((s & 1) * 2 + 1) * 3) * B3_PLANE;
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        }
#pragma unroll
        for (int kb = 0; kb < B3_BK / 16; ++kb) {  // 16-deep blocks of the K-step: lane (r, h) holds k = 16 kb + 8 h + 0..7
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    a[i][p] = *reinterpret_cast<const bf16x8 *>(pa + p * B3_PLANE + ((wr * 2 + i) * 32 + l31) * B3_PITCH + kb * 16 + lh * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[j][p] = *reinterpret_cast<const bf16x8 *>(pb + p * B3_PLANE + ((wc * 2 + j) * 32 + l31) * B3_PITCH + kb * 16 + lh * 8);
            // small terms first; four accumulators interleaved under every term pair
#define B3_TERM(ACC, PA, PB)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                       \
        ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[j][PB], ACC[i][j], 0, 0, 0)
            B3_TERM(acc, 2, 0);
            B3_TERM(acc, 0, 2);
            B3_TERM(acc, 1, 1);
            B3_TERM(acc, 1, 0);
            B3_TERM(acc, 0, 1);
            B3_TERM(acc, 0, 0);
#undef B3_TERM
        }
        if (++ks == ksteps) {
            ks = 0;
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
            if (STATS != 0) {
                const int valid = (int)min((int64_t)64, max((int64_t)0, M - (m0 + wr * 64)));  // wave-uniform
                const float pad = (float)(64 - valid);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!have_shift) kshift[j] = __shfl(acc[0][j][0], l31);
                    const float k = kshift[j];
                    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float d = acc[i][j][e] - k;
                            s1 += d;
                            s2 += d * d;
                        }
                    s1 += __shfl_xor(s1, 32);
                    s2 += __shfl_xor(s2, 32);
                    s1 += pad * k;
                    s2 -= pad * (k * k);
                    run1[j] += s1;
                    run2[j] += s2;
                }
                if (valid > 0) have_shift = true;
                run_rows += valid;
            }
            item += gridDim.x;
            decode(item, m0, n0);
        }
        __syncthreads();
    }
    if (STATS != 0) {
        const int per = 8 * tiles_n;
        const int64_t slot = (int64_t)(blockIdx.x / per) * 8 + (blockIdx.x & 7);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0_first + (wc * 2 + j) * 32 + l31;
            if (lh == 0 && n < N) {
                float *pr = partial + ((size_t)(slot * 2 + wr) * 4) * N + n;
                pr[0] = run1[j];
                pr[(size_t)N] = run2[j];
                pr[(size_t)2 * N] = kshift[j];
                pr[(size_t)3 * N] = (float)run_rows;
            }
        }
    }
}

}  // namespace tp3d

// Shapes: as tp3d_gemm_rows_bnact_sp_f32 with more than 64 output columns.  Chunk count as tp3d_gemm_rows_sp_chunks.
static int b3_tiles_n(int64_t M, int N, int K)
{
    if (M <= 0 || N <= 64 || K < 4 || (K & 3) || K > 512) return 0;
    const int rem = N % tp3d::B3_BN;
    if (rem > 0 && rem <= 64) return 0;
    const int tiles_n = (N + tp3d::B3_BN - 1) / tp3d::B3_BN;
    if (512 % (8 * tiles_n)) return 0;
    const int64_t row_blocks = (M + tp3d::B3_BM - 1) / tp3d::B3_BM;
    if ((row_blocks + 7) / 8 * 8 * tiles_n < 512) return 0;
    return tiles_n;
}
static int64_t b3_items(int64_t M, int tiles_n) { return ((M + tp3d::B3_BM - 1) / tp3d::B3_BM + 7) / 8 * 8 * tiles_n; }
static int b3_grid(int64_t items, bool side) { return (side && items >= 2048) ? 1024 : 512; }

TP3D_EXPORT int tp3d_gemm_rows_x3_chunks(int64_t M, int N, int K, int with_act_out)
{
    const int tiles_n = b3_tiles_n(M, N, K);
    return tiles_n ? 2 * (b3_grid(b3_items(M, tiles_n), with_act_out != 0) / tiles_n) : 0;
}

TP3D_EXPORT int tp3d_gemm_rows_bnact_x3_f32(const float *Y, const float *mean, const float *scale, const float *beta,
                                            float slope, const float *Bt, int64_t M, int N, int K, float *C,
                                            float *stat_partial, float *act_out, int reverse, void *stream)
{
    using namespace tp3d;
    const int tiles_n = b3_tiles_n(M, N, K);
    if (!tiles_n || !Y || !mean || !scale || !beta || !Bt || !C) return TP3D_E_BADARG;
    const int64_t items = b3_items(M, tiles_n);
    const int grid = b3_grid(items, act_out != nullptr);
    hipStream_t s = (hipStream_t)stream;
    static bool set0[64] = {false}, set2[64] = {false};
    if (stat_partial) {
        allow_large_dynamic_lds(reinterpret_cast<const void *>(&gemm_rows_x3_kernel<2>), B3_LDS_BYTES, set2);
        hipLaunchKernelGGL(gemm_rows_x3_kernel<2>, dim3(grid), dim3(B3_BLOCK), B3_LDS_BYTES, s, Y, Bt, M, N, K, tiles_n, items, C,
                           stat_partial, act_out, mean, scale, beta, slope, reverse);
    } else {
        allow_large_dynamic_lds(reinterpret_cast<const void *>(&gemm_rows_x3_kernel<0>), B3_LDS_BYTES, set0);
        hipLaunchKernelGGL(gemm_rows_x3_kernel<0>, dim3(grid), dim3(B3_BLOCK), B3_LDS_BYTES, s, Y, Bt, M, N, K, tiles_n, items, C,
                           (float *)nullptr, act_out, mean, scale, beta, slope, reverse);
    }
    return check_launch();
}
