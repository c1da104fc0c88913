// EXPERIMENT (tools/probes/gemm_sp.py builds it into /tmp on the GPU box; the product form is csrc/gemm_rows_sp.hip).
// Rows GEMM with SPLIT ROLES: C[M,N] = A[M,K] * Bt[N,K]^T (fp32 MFMA), the same contraction as gemm_rows.hip.
//
// A workgroup is 8 waves: waves 0-3 only issue MFMAs (2 x 2 waves, each 2 x 2 tiles of 32 x 32: a 128 x 128 output tile),
// waves 4-7 only move data -- global loads two K-steps ahead into registers, then LDS writes into the buffer the compute
// waves will read NEXT step.  Two LDS buffers, ONE barrier per K-step, persistent workgroups over (row block, column tile)
// items.  Measured as a plain GEMM it is on par with gemm_rows.hip (where every wave does both jobs in turn); probe bit 0
// serves the input from cache and bit 1 skips the stores, to see what bounds it; PRO 1 puts a BatchNorm + LeakyReLU
// prologue into the loader waves (free there -- DESIGN.md section 5).
#include "tp3d_common.h"

namespace tp3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SP_BLOCK = 512;
constexpr int SP_BM = 128, SP_BN = 128, SP_BK = 32;
constexpr int SP_LD = SP_BK + 4;  // 36-float pitch: aligned float4 stores, conflict-free lane = row fragment reads
constexpr int SP_TILE = SP_BM * SP_LD;

// value select per component: `c ? v : zero` on two float4 lvalues selects an address and keeps both in scratch memory
__device__ __forceinline__ float4 sp_keep(bool c, float4 v) {
    return make_float4(c ? v.x : 0.0f, c ? v.y : 0.0f, c ? v.z : 0.0f, c ? v.w : 0.0f);
}

// STATS: 0 none, 2 one statistics chunk per (workgroup, wave row) -- shifted sums as in gemm_rows.hip
// PRO 1: the loader waves turn the staged pre-BatchNorm rows into activated ones, leaky((y - mean) * scale + beta) per
// contraction channel, on their way into LDS (K <= SP_PRO_KMAX; constants in LDS)
constexpr int SP_PRO_KMAX = 512;
struct SpPro {
    const float *mean, *scale, *beta;
    float slope;
};

template <int STATS, int PRO>
__global__ __launch_bounds__(SP_BLOCK, 4) void gemm_rows_sp_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                                                                int64_t M, int N, int K, int tiles_n, int64_t items,
                                                                float *__restrict__ C, float *__restrict__ partial, int probe,
                                                                SpPro pro)
{
    __shared__ __attribute__((aligned(16))) float sK[PRO ? 3 * SP_PRO_KMAX : 4];
    if (PRO) {
        for (int k = threadIdx.x; k < SP_PRO_KMAX; k += SP_BLOCK) {
            sK[k] = k < K ? pro.mean[k] : 0.0f;
            sK[SP_PRO_KMAX + k] = k < K ? pro.scale[k] : 0.0f;
            sK[2 * SP_PRO_KMAX + k] = k < K ? pro.beta[k] : 0.0f;
        }
        __syncthreads();  // the loader waves read the table for their first LDS write, before the first barrier of the K walk
    }
    // probe (experiments only): bit 0 reads A rows modulo 8192 (cache-resident input), bit 1 skips the stores of C
    const int64_t a_wrap = (probe & 1) ? 8191 : ~(int64_t)0;
    const bool put = !(probe & 2);
    __shared__ __attribute__((aligned(16))) float sA[2][SP_TILE];
    __shared__ __attribute__((aligned(16))) float sB[2][SP_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksteps = (K + SP_BK - 1) / SP_BK;
    const int tail_groups = (K - (ksteps - 1) * SP_BK + 7) / 8;
    if ((int64_t)blockIdx.x >= items) return;
    const int64_t my_items = (items - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t total = my_items * ksteps;  // K-steps this workgroup walks, as one flat sequence

    // item -> (row block, column tile): the column tiles of one row block are 8 ids apart (same XCD: shared L2 for A)
    auto decode = [&](int64_t item, int64_t &m0, int &n0) {
        const int64_t grp = item / (8 * tiles_n);
        const int rem = (int)(item % (8 * tiles_n));
        m0 = (grp * 8 + (rem & 7)) * SP_BM;
        n0 = (rem >> 3) * SP_BN;
    };

    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int lt = tid - 256;
        const int frow = lt >> 3, fk4 = (lt & 7) * 4;  // slot i of this thread: tile row frow + 32 i, floats fk4..fk4+3
        int64_t f_item = blockIdx.x, f_m0;
        int f_n0, f_ks = 0;
        decode(f_item, f_m0, f_n0);
        auto act4 = [&](const float4 raw, int kk) __attribute__((always_inline)) -> float4 {
            const int kc = min(kk, SP_PRO_KMAX - 4);
            const float4 mu = *reinterpret_cast<const float4 *>(&sK[kc]);
            const float4 sc = *reinterpret_cast<const float4 *>(&sK[SP_PRO_KMAX + kc]);
            const float4 be = *reinterpret_cast<const float4 *>(&sK[2 * SP_PRO_KMAX + kc]);
            const float z0 = (raw.x - mu.x) * sc.x + be.x, z1 = (raw.y - mu.y) * sc.y + be.y;
            const float z2 = (raw.z - mu.z) * sc.z + be.z, z3 = (raw.w - mu.w) * sc.w + be.w;
            return make_float4(z0 > 0.0f ? z0 : z0 * pro.slope, z1 > 0.0f ? z1 : z1 * pro.slope,
                               z2 > 0.0f ? z2 : z2 * pro.slope, z3 > 0.0f ? z3 : z3 * pro.slope);
        };
        // every fetch issues exactly eight unconditional loads (rows / columns past the matrix read a valid address and
        // are zeroed when they are written to LDS), so the compiler can count them and waits for the older stage only.
        // The two stages are plain named variables filled by macros: arrays or structs handed to lambdas ended up in
        // scratch memory.
#define SP_FETCH(S)                                                                                                   \
    do {                                                                                                              \
        m0_##S = f_m0, n0_##S = f_n0, k0_##S = f_ks * SP_BK;                                                          \
        const int kk = min(k0_##S + fk4, K - 4);                                                                      \
        a0_##S = *reinterpret_cast<const float4 *>(A + (min(f_m0 + frow + 0, M - 1) & a_wrap) * K + kk);                         \
        a1_##S = *reinterpret_cast<const float4 *>(A + (min(f_m0 + frow + 32, M - 1) & a_wrap) * K + kk);                        \
        a2_##S = *reinterpret_cast<const float4 *>(A + (min(f_m0 + frow + 64, M - 1) & a_wrap) * K + kk);                        \
        a3_##S = *reinterpret_cast<const float4 *>(A + (min(f_m0 + frow + 96, M - 1) & a_wrap) * K + kk);                        \
        b0_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 0, N - 1) * K + kk);                \
        b1_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 32, N - 1) * K + kk);               \
        b2_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 64, N - 1) * K + kk);               \
        b3_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 96, N - 1) * K + kk);               \
        if (++f_ks == ksteps) {                                                                                       \
            f_ks = 0;                                                                                                 \
            f_item += gridDim.x;                                                                                      \
            const int64_t grp = f_item / (8 * tiles_n);                                                               \
            const int rem = (int)(f_item % (8 * tiles_n));                                                            \
            f_m0 = (grp * 8 + (rem & 7)) * SP_BM;                                                                     \
            f_n0 = (rem >> 3) * SP_BN;                                                                                \
        }                                                                                                             \
    } while (0)
#define SP_STASH(S, BUF)                                                                                              \
    do {                                                                                                              \
        const bool kin = k0_##S + fk4 < K;                                                                            \
        float *da = &sA[BUF][frow * SP_LD + fk4], *db = &sB[BUF][frow * SP_LD + fk4];                                 \
        *reinterpret_cast<float4 *>(da + 0 * 32 * SP_LD) = sp_keep(kin && m0_##S + frow + 0 < M, PRO ? act4(a0_##S, k0_##S + fk4) : a0_##S);           \
        *reinterpret_cast<float4 *>(da + 1 * 32 * SP_LD) = sp_keep(kin && m0_##S + frow + 32 < M, PRO ? act4(a1_##S, k0_##S + fk4) : a1_##S);          \
        *reinterpret_cast<float4 *>(da + 2 * 32 * SP_LD) = sp_keep(kin && m0_##S + frow + 64 < M, PRO ? act4(a2_##S, k0_##S + fk4) : a2_##S);          \
        *reinterpret_cast<float4 *>(da + 3 * 32 * SP_LD) = sp_keep(kin && m0_##S + frow + 96 < M, PRO ? act4(a3_##S, k0_##S + fk4) : a3_##S);          \
        *reinterpret_cast<float4 *>(db + 0 * 32 * SP_LD) = sp_keep(kin && n0_##S + frow + 0 < N, b0_##S);           \
        *reinterpret_cast<float4 *>(db + 1 * 32 * SP_LD) = sp_keep(kin && n0_##S + frow + 32 < N, b1_##S);          \
        *reinterpret_cast<float4 *>(db + 2 * 32 * SP_LD) = sp_keep(kin && n0_##S + frow + 64 < N, b2_##S);          \
        *reinterpret_cast<float4 *>(db + 3 * 32 * SP_LD) = sp_keep(kin && n0_##S + frow + 96 < N, b3_##S);          \
    } while (0)
        float4 a0_0, a1_0, a2_0, a3_0, b0_0, b1_0, b2_0, b3_0, a0_1, a1_1, a2_1, a3_1, b0_1, b1_1, b2_1, b3_1;
        int64_t m0_0, m0_1;
        int n0_0, n0_1, k0_0, k0_1;
        SP_FETCH(0);  // step 0
        SP_FETCH(1);  // step 1
        SP_STASH(0, 0);
        SP_FETCH(0);  // step 2
        __syncthreads();  // B0: buffer 0 holds step 0
#pragma unroll 1
        for (int64_t s = 0; s < total; s += 2) {
            // during compute step s: write step s+1 (stage 1) into buffer 1, refill stage 1 with step s+3
            SP_STASH(1, 1);
            SP_FETCH(1);
            __syncthreads();
            if (s + 1 >= total) break;
            // during compute step s+1: write step s+2 (stage 0) into buffer 0, refill stage 0 with step s+4
            SP_STASH(0, 0);
            SP_FETCH(0);
            __syncthreads();
        }
#undef SP_FETCH
#undef SP_STASH
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    int64_t item = blockIdx.x, m0;
    int n0, ks = 0;
    decode(item, m0, n0);
    const int n0_first = n0;
    float run1[2] = {0.0f, 0.0f}, run2[2] = {0.0f, 0.0f}, kshift[2] = {0.0f, 0.0f};
    bool have_shift = false;
    int run_rows = 0;
    f32x16 acc[2][2];
    __syncthreads();  // B0
    for (int64_t s = 0; s < total; ++s) {
        const float *sa = sA[s & 1], *sb = sB[s & 1];
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        }
        const int ng = (ks + 1 < ksteps) ? SP_BK / 8 : tail_groups;
#pragma unroll
        for (int g = 0; g < SP_BK / 8; ++g) {
            if (g < ng) {
                float4 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i] = *reinterpret_cast<const float4 *>(&sa[((wr * 2 + i) * 32 + l31) * SP_LD + g * 8 + lh * 4]);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    b[j] = *reinterpret_cast<const float4 *>(&sb[((wc * 2 + j) * 32 + l31) * SP_LD + g * 8 + lh * 4]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                    }
            }
        }
        if (++ks == ksteps) {
            ks = 0;
            // ---- epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + (wc * 2 + j) * 32 + l31;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int64_t m = m0 + (wr * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        if (put && m < M && n < N) C[m * N + n] = acc[i][j][e];
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

// Experimental entry (tools/microbench.py): same contract as tp3d_gemm_rows_f32 without statistics; grid workgroups.
TP3D_EXPORT int tp3d_gemm_rows_sp_f32(const float *A, const float *Bt, int64_t M, int N, int K, float *C, int grid,
                                      int probe, void *stream)
{
    using namespace tp3d;
    if (M <= 0 || N <= 0 || K < 4 || (K & 3) || !A || !Bt || !C || grid <= 0 || (grid & 7)) return TP3D_E_BADARG;
    const int tiles_n = (N + SP_BN - 1) / SP_BN;
    const int64_t row_blocks = (M + SP_BM - 1) / SP_BM;
    const int64_t items = (row_blocks + 7) / 8 * 8 * tiles_n;
    hipLaunchKernelGGL((gemm_rows_sp_kernel<0, 0>), dim3(grid), dim3(SP_BLOCK), 0, (hipStream_t)stream, A, Bt, M, N, K, tiles_n,
                       items, C, (float *)nullptr, probe, SpPro{});
    return check_launch();
}

// The same with the BatchNorm + LeakyReLU prologue of tp3d_gemm_rows_bnact_f32 in the loader waves (no statistics).
TP3D_EXPORT int tp3d_gemm_rows_sp_bnact_f32(const float *Y, const float *mean, const float *scale, const float *beta,
                                            float slope, const float *Bt, int64_t M, int N, int K, float *C, int grid,
                                            void *stream)
{
    using namespace tp3d;
    if (M <= 0 || N <= 0 || K < 4 || (K & 3) || K > SP_PRO_KMAX || !Y || !Bt || !C || grid <= 0 || (grid & 7)) return TP3D_E_BADARG;
    const int tiles_n = (N + SP_BN - 1) / SP_BN;
    const int64_t row_blocks = (M + SP_BM - 1) / SP_BM;
    const int64_t items = (row_blocks + 7) / 8 * 8 * tiles_n;
    SpPro pro{mean, scale, beta, slope};
    hipLaunchKernelGGL((gemm_rows_sp_kernel<0, 1>), dim3(grid), dim3(SP_BLOCK), 0, (hipStream_t)stream, Y, Bt, M, N, K, tiles_n,
                       items, C, (float *)nullptr, 0, pro);
    return check_launch();
}
