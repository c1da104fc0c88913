// Rows GEMM with SPLIT ROLES and the BatchNorm + LeakyReLU of the PREVIOUS layer applied on the way in:
//   C[M,N] = leaky((Y[M,K] - mean) * scale + beta) * Bt[N,K]^T      (fp32 MFMA; the contraction of gemm_rows.hip)
//
// A workgroup is 8 waves: waves 0-3 only issue MFMAs (2 x 2 waves, each 2 x 2 tiles of 32 x 32: a 128 x 128 output tile),
// waves 4-7 only move data -- global loads two K-steps ahead into registers, the per-channel affine + activation on those
// registers, LDS writes into the buffer the compute waves read NEXT step, and (training) the activated rows as a side
// output for the weight-gradient pass.  Two LDS buffers, ONE barrier per K-step, persistent workgroups over
// (row block, column tile) items.  As a plain GEMM this layout is on par with gemm_rows.hip (tools/probes/gemm_sp.py);
// what it buys is that the prologue costs the MFMA waves nothing: in gemm_rows.hip's prologue variants every wave does
// both jobs, the constants and the second operand push it to 230-256 VGPRs and the MFMA rate drops to 50-67 TFLOP/s,
// while here the activated layer input is formed in registers the MFMA waves never see (524288 x 128 x 128: separate
// BatchNorm pass + GEMM 306 us, prologue in the MFMA waves 283 us, in the loader waves 203 us; bit-identical results).
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
// PRO 2: the loader waves form dY, the BatchNorm + LeakyReLU BACKWARD of (dA, Y) with the reduction constants c1, c2 of
// tp3d_bn_bwd_reduce_f32 -- dY = scale * ((dA * act'(z) - c1) - (y - mean) * c2), z = (y - mean) * scale + beta -- stage
// it as the A operand and write it out as the side output (K <= SP_BWD_KMAX; five constants per channel in LDS)
constexpr int SP_PRO_KMAX = 512;
constexpr int SP_BWD_KMAX = 256;
struct SpPro {
    const float *mean, *scale, *beta;
    float slope;
    const float *dA, *c1, *c2;  // PRO 2, 3
    const int *arg;             // PRO 3: winning row (0 .. ns-1) of every (group, channel)
    int ns_shift;               //        ns = 1 << ns_shift rows per group
    int pad_lo, pad_hi;         // PRO 2, 3: columns in front of / behind C's N columns in its rows, to be zeroed
};

template <int STATS, int PRO, int BN>
__global__ __launch_bounds__(SP_BLOCK, 4) void gemm_rows_sp_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                                                                int64_t M, int N, int K, int tiles_n, int64_t items,
                                                                float *__restrict__ C, int64_t ldc,
                                                                float *__restrict__ partial, float *__restrict__ act_out,
                                                                SpPro pro, int reverse)
{
    constexpr int KMAX = PRO >= 2 ? SP_BWD_KMAX : SP_PRO_KMAX;
    __shared__ __attribute__((aligned(16))) float sK[PRO >= 2 ? 5 * KMAX : (PRO ? 3 * KMAX : 4)];
    if (PRO) {
        for (int k = threadIdx.x; k < KMAX; k += SP_BLOCK) {
            sK[k] = k < K ? pro.mean[k] : 0.0f;
            sK[KMAX + k] = k < K ? pro.scale[k] : 0.0f;
            sK[2 * KMAX + k] = k < K ? pro.beta[k] : 0.0f;
            if (PRO >= 2) {
                sK[3 * KMAX + k] = k < K ? pro.c1[k] : 0.0f;
                sK[4 * KMAX + k] = k < K ? pro.c2[k] : 0.0f;
            }
        }
        __syncthreads();  // the loader waves read the table for their first LDS write, before the first barrier of the K walk
    }
    __shared__ __attribute__((aligned(16))) float sA[2][SP_TILE];
    __shared__ __attribute__((aligned(16))) float sB[2][BN * SP_LD];
    constexpr int TJ = BN / 64;  // 32-column MFMA tiles per compute wave (the wave grid stays 2 x 2)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ksteps = (K + SP_BK - 1) / SP_BK;
    const int tail_groups = (K - (ksteps - 1) * SP_BK + 7) / 8;
    if ((int64_t)blockIdx.x >= items) return;
    const int64_t my_items = (items - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t total = my_items * ksteps;  // K-steps this workgroup walks, as one flat sequence

    // item -> (row block, column tile): the column tiles of one row block are 8 ids apart (same XCD: shared L2 for A)
    const int64_t last_grp = items / (8 * tiles_n) - 1;
    auto decode = [&](int64_t item, int64_t &m0, int &n0) {
        int64_t grp = item / (8 * tiles_n);
        const int rem = (int)(item % (8 * tiles_n));
        if (reverse) grp = max(last_grp - grp, (int64_t)0);  // row blocks last to first (look-ahead past the last item: group 0)
        m0 = (grp * 8 + (rem & 7)) * SP_BM;
        n0 = (rem >> 3) * BN;
    };

    if (wave >= 4) {
        // ------------------------------------------------------------------ loader waves
        const int lt = tid - 256;
        const int frow = lt >> 3, fk4 = (lt & 7) * 4;  // slot i of this thread: tile row frow + 32 i, floats fk4..fk4+3
        int64_t f_item = blockIdx.x, f_m0;
        int f_n0, f_ks = 0;
        decode(f_item, f_m0, f_n0);
        auto act4 = [&](const float4 raw, int kk) __attribute__((always_inline)) -> float4 {
            const int kc = min(kk, KMAX - 4);
            const float4 mu = *reinterpret_cast<const float4 *>(&sK[kc]);
            const float4 sc = *reinterpret_cast<const float4 *>(&sK[KMAX + kc]);
            const float4 be = *reinterpret_cast<const float4 *>(&sK[2 * KMAX + kc]);
            const float z0 = (raw.x - mu.x) * sc.x + be.x, z1 = (raw.y - mu.y) * sc.y + be.y;
            const float z2 = (raw.z - mu.z) * sc.z + be.z, z3 = (raw.w - mu.w) * sc.w + be.w;
            return make_float4(z0 > 0.0f ? z0 : z0 * pro.slope, z1 > 0.0f ? z1 : z1 * pro.slope,
                               z2 > 0.0f ? z2 : z2 * pro.slope, z3 > 0.0f ? z3 : z3 * pro.slope);
        };
        if constexpr (PRO >= 2) {
            // PRO 3: dA is the gradient of the max-POOLED output (M / ns rows) with the winning row of every group: it is
            // small and L2-resident like the weights, so it travels with the B operand, one step ahead
            constexpr bool POOL = PRO == 3;
            // ---- backward prologue: two A streams (Y, dA) two K-steps ahead, the small B operand (L2-resident weights) one
            // step ahead and issued BEFORE the A loads of its iteration, so that waiting for it leaves the newest A stage
            // in flight (loads complete in order)
            auto bwd4 = [&](const float4 raw, const float4 d, int kk) __attribute__((always_inline)) -> float4 {
                const int kc = min(kk, KMAX - 4);
                const float4 mu = *reinterpret_cast<const float4 *>(&sK[kc]);
                const float4 sc = *reinterpret_cast<const float4 *>(&sK[KMAX + kc]);
                const float4 be = *reinterpret_cast<const float4 *>(&sK[2 * KMAX + kc]);
                const float4 c1 = *reinterpret_cast<const float4 *>(&sK[3 * KMAX + kc]);
                const float4 c2 = *reinterpret_cast<const float4 *>(&sK[4 * KMAX + kc]);
                auto one = [&](float y, float dd, float m, float s_, float b, float k1, float k2) __attribute__((always_inline)) -> float {
                    const float yc = y - m;
                    const float z = yc * s_ + b;
                    const float dz = dd * (z > 0.0f ? 1.0f : pro.slope);
                    return s_ * ((dz - k1) - yc * k2);
                };
                return make_float4(one(raw.x, d.x, mu.x, sc.x, be.x, c1.x, c2.x), one(raw.y, d.y, mu.y, sc.y, be.y, c1.y, c2.y),
                                   one(raw.z, d.z, mu.z, sc.z, be.z, c1.z, c2.z), one(raw.w, d.w, mu.w, sc.w, be.w, c1.w, c2.w));
            };
            const float *dA = pro.dA;
            int64_t g_item = blockIdx.x, g_m0 = f_m0;  // the B cursor (one step behind the A cursor)
            int g_n0 = f_n0, g_ks = 0;
#define SP_FETCH_A(S)                                                                                                 \
    do {                                                                                                              \
        m0_##S = f_m0, n0_##S = f_n0, k0_##S = f_ks * SP_BK;                                                          \
        const int kk = min(k0_##S + fk4, K - 4);                                                                      \
        const int64_t o0 = min(f_m0 + frow + 0, M - 1) * K + kk, o1 = min(f_m0 + frow + 32, M - 1) * K + kk;          \
        const int64_t o2 = min(f_m0 + frow + 64, M - 1) * K + kk, o3 = min(f_m0 + frow + 96, M - 1) * K + kk;         \
        y0_##S = *reinterpret_cast<const float4 *>(A + o0), y1_##S = *reinterpret_cast<const float4 *>(A + o1);       \
        y2_##S = *reinterpret_cast<const float4 *>(A + o2), y3_##S = *reinterpret_cast<const float4 *>(A + o3);       \
        if constexpr (!POOL) {                                                                                        \
            d0_##S = *reinterpret_cast<const float4 *>(dA + o0), d1_##S = *reinterpret_cast<const float4 *>(dA + o1); \
            d2_##S = *reinterpret_cast<const float4 *>(dA + o2), d3_##S = *reinterpret_cast<const float4 *>(dA + o3); \
        }                                                                                                             \
        if (++f_ks == ksteps) {                                                                                       \
            f_ks = 0;                                                                                                 \
            f_item += gridDim.x;                                                                                      \
            decode(f_item, f_m0, f_n0);                                                                               \
        }                                                                                                             \
    } while (0)
#define SP_FETCH_B()                                                                                                  \
    do {                                                                                                              \
        bn0 = g_n0, bk0 = g_ks * SP_BK;                                                                               \
        const int kk = min(bk0 + fk4, K - 4);                                                                         \
            b0 = *reinterpret_cast<const float4 *>(Bt + (size_t)min(g_n0 + frow + 0, N - 1) * K + kk);                \
            b1 = *reinterpret_cast<const float4 *>(Bt + (size_t)min(g_n0 + frow + 32, N - 1) * K + kk);               \
            if constexpr (BN == 128) {                                                                                \
                b2 = *reinterpret_cast<const float4 *>(Bt + (size_t)min(g_n0 + frow + 64, N - 1) * K + kk);           \
                b3 = *reinterpret_cast<const float4 *>(Bt + (size_t)min(g_n0 + frow + 96, N - 1) * K + kk);           \
            }                                                                                                         \
        if constexpr (POOL) { /* groups of >= 64 rows: tile rows r and r + 32 share their group */                    \
            const int64_t g0 = (min(g_m0 + frow + 0, M - 1) >> pro.ns_shift) * K + kk;                                \
            const int64_t g2 = (min(g_m0 + frow + 64, M - 1) >> pro.ns_shift) * K + kk;                               \
            p0 = *reinterpret_cast<const float4 *>(dA + g0), q0 = *reinterpret_cast<const int4 *>(pro.arg + g0);      \
            p2 = *reinterpret_cast<const float4 *>(dA + g2), q2 = *reinterpret_cast<const int4 *>(pro.arg + g2);      \
        }                                                                                                             \
        if (++g_ks == ksteps) {                                                                                       \
            g_ks = 0;                                                                                                 \
            g_item += gridDim.x;                                                                                      \
            decode(g_item, g_m0, g_n0);                                                                               \
        }                                                                                                             \
    } while (0)
#define SP_STASH_D(S, BUF, I)                                                                                         \
    do {                                                                                                              \
        const bool in = kin && m0_##S + frow + 32 * I < M;                                                            \
        float4 dd;                                                                                                    \
        if constexpr (POOL) {                                                                                         \
            const int srow = (int)(m0_##S + frow + 32 * I) & ((1 << pro.ns_shift) - 1);                               \
            dd = make_float4(q##I.x == srow ? p##I.x : 0.0f, q##I.y == srow ? p##I.y : 0.0f,                          \
                             q##I.z == srow ? p##I.z : 0.0f, q##I.w == srow ? p##I.w : 0.0f);                         \
        } else {                                                                                                      \
            dd = d##I##_##S;                                                                                          \
        }                                                                                                             \
        const float4 v = sp_keep(in, bwd4(y##I##_##S, dd, k0_##S + fk4));                                             \
        *reinterpret_cast<float4 *>(da + I * 32 * SP_LD) = v;                                                         \
        if (side && in) *reinterpret_cast<float4 *>(act_out + (m0_##S + frow + 32 * I) * K + k0_##S + fk4) = v;       \
    } while (0)
#define SP_STASH2(S, BUF)                                                                                             \
    do {                                                                                                              \
        const bool kin = k0_##S + fk4 < K, kinb = bk0 + fk4 < K;                                                      \
        const bool side = act_out != nullptr && n0_##S == 0;                                                          \
        float *da = &sA[BUF][frow * SP_LD + fk4], *db = &sB[BUF][frow * SP_LD + fk4];                                 \
        SP_STASH_D(S, BUF, 0);                                                                                        \
        SP_STASH_D(S, BUF, 1);                                                                                        \
        SP_STASH_D(S, BUF, 2);                                                                                        \
        SP_STASH_D(S, BUF, 3);                                                                                        \
        *reinterpret_cast<float4 *>(db + 0 * 32 * SP_LD) = sp_keep(kinb && bn0 + frow + 0 < N, b0);                   \
        *reinterpret_cast<float4 *>(db + 1 * 32 * SP_LD) = sp_keep(kinb && bn0 + frow + 32 < N, b1);                  \
        if constexpr (BN == 128) {                                                                                    \
            *reinterpret_cast<float4 *>(db + 2 * 32 * SP_LD) = sp_keep(kinb && bn0 + frow + 64 < N, b2);              \
            *reinterpret_cast<float4 *>(db + 3 * 32 * SP_LD) = sp_keep(kinb && bn0 + frow + 96 < N, b3);              \
        }                                                                                                             \
    } while (0)
            float4 y0_0, y1_0, y2_0, y3_0, d0_0, d1_0, d2_0, d3_0, y0_1, y1_1, y2_1, y3_1, d0_1, d1_1, d2_1, d3_1;
            float4 b0, b1, b2, b3, p0, p2;
            int4 q0, q2;
            float4 &p1 = p0, &p3 = p2;  // (tile rows r + 32 and r + 96: the groups of r and r + 64)
            int4 &q1 = q0, &q3 = q2;
            int64_t m0_0, m0_1;
            int n0_0, n0_1, k0_0, k0_1, bn0, bk0;
            SP_FETCH_B();   // B of step 0
            SP_FETCH_A(0);  // A of step 0
            SP_STASH2(0, 0);
            SP_FETCH_A(1);  // A of step 1 (older than the B below: the order the loop keeps)
            SP_FETCH_B();   // B of step 1
            SP_FETCH_A(0);  // A of step 2
            __syncthreads();  // B0: buffer 0 holds step 0
#pragma unroll 1
            for (int64_t s = 0; s < total; s += 2) {
                // during compute step s: step s+1 (A stage 1, B registers) into buffer 1; then B of s+2, A of s+3
                SP_STASH2(1, 1);
                SP_FETCH_B();
                SP_FETCH_A(1);
                __syncthreads();
                if (s + 1 >= total) break;
                SP_STASH2(0, 0);
                SP_FETCH_B();
                SP_FETCH_A(0);
                __syncthreads();
            }
#undef SP_FETCH_A
#undef SP_FETCH_B
#undef SP_STASH2
#undef SP_STASH_D
            return;
        }
        // every fetch issues exactly eight unconditional loads (rows / columns past the matrix read a valid address and
        // are zeroed when they are written to LDS), so the compiler can count them and waits for the older stage only.
        // The two stages are plain named variables filled by macros: arrays or structs handed to lambdas ended up in
        // scratch memory.
#define SP_FETCH(S)                                                                                                   \
    do {                                                                                                              \
        m0_##S = f_m0, n0_##S = f_n0, k0_##S = f_ks * SP_BK;                                                          \
        const int kk = min(k0_##S + fk4, K - 4);                                                                      \
        a0_##S = *reinterpret_cast<const float4 *>(A + min(f_m0 + frow + 0, M - 1) * K + kk);                         \
        a1_##S = *reinterpret_cast<const float4 *>(A + min(f_m0 + frow + 32, M - 1) * K + kk);                        \
        a2_##S = *reinterpret_cast<const float4 *>(A + min(f_m0 + frow + 64, M - 1) * K + kk);                        \
        a3_##S = *reinterpret_cast<const float4 *>(A + min(f_m0 + frow + 96, M - 1) * K + kk);                        \
        b0_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 0, N - 1) * K + kk);                \
        b1_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 32, N - 1) * K + kk);               \
        if constexpr (BN == 128) {                                                                                    \
            b2_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 64, N - 1) * K + kk);           \
            b3_##S = *reinterpret_cast<const float4 *>(Bt + (size_t)min(f_n0 + frow + 96, N - 1) * K + kk);           \
        }                                                                                                             \
        if (++f_ks == ksteps) {                                                                                       \
            f_ks = 0;                                                                                                 \
            f_item += gridDim.x;                                                                                      \
            decode(f_item, f_m0, f_n0);                                                                               \
        }                                                                                                             \
    } while (0)
#define SP_STASH_A(S, BUF, I)                                                                                         \
    do {                                                                                                              \
        const bool in = kin && m0_##S + frow + 32 * I < M;                                                            \
        const float4 v = sp_keep(in, PRO ? act4(a##I##_##S, k0_##S + fk4) : a##I##_##S);                              \
        *reinterpret_cast<float4 *>(da + I * 32 * SP_LD) = v;                                                         \
        if (side && in) *reinterpret_cast<float4 *>(act_out + (m0_##S + frow + 32 * I) * K + k0_##S + fk4) = v;       \
    } while (0)
#define SP_STASH(S, BUF)                                                                                              \
    do {                                                                                                              \
        const bool kin = k0_##S + fk4 < K;                                                                            \
        const bool side = act_out != nullptr && n0_##S == 0; /* one column tile of a row block writes the side rows */ \
        float *da = &sA[BUF][frow * SP_LD + fk4], *db = &sB[BUF][frow * SP_LD + fk4];                                 \
        SP_STASH_A(S, BUF, 0);                                                                                        \
        SP_STASH_A(S, BUF, 1);                                                                                        \
        SP_STASH_A(S, BUF, 2);                                                                                        \
        SP_STASH_A(S, BUF, 3);                                                                                        \
        *reinterpret_cast<float4 *>(db + 0 * 32 * SP_LD) = sp_keep(kin && n0_##S + frow + 0 < N, b0_##S);             \
        *reinterpret_cast<float4 *>(db + 1 * 32 * SP_LD) = sp_keep(kin && n0_##S + frow + 32 < N, b1_##S);            \
        if constexpr (BN == 128) {                                                                                    \
            *reinterpret_cast<float4 *>(db + 2 * 32 * SP_LD) = sp_keep(kin && n0_##S + frow + 64 < N, b2_##S);        \
            *reinterpret_cast<float4 *>(db + 3 * 32 * SP_LD) = sp_keep(kin && n0_##S + frow + 96 < N, b3_##S);        \
        }                                                                                                             \
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
#undef SP_STASH_A
        return;
    }

    // ---------------------------------------------------------------------- compute waves
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    int64_t item = blockIdx.x, m0;
    int n0, ks = 0;
    decode(item, m0, n0);
    const int n0_first = n0;
    float run1[TJ] = {}, run2[TJ] = {}, kshift[TJ] = {};
    bool have_shift = false;
    int run_rows = 0;
    f32x16 acc[2][TJ];
    __syncthreads();  // B0
    for (int64_t s = 0; s < total; ++s) {
        const float *sa = sA[s & 1], *sb = sB[s & 1];
        if (ks == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        }
        const int ng = (ks + 1 < ksteps) ? SP_BK / 8 : tail_groups;
#pragma unroll
        for (int g = 0; g < SP_BK / 8; ++g) {
            if (g < ng) {
                float4 a[2], b[TJ];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    a[i] = *reinterpret_cast<const float4 *>(&sa[((wr * 2 + i) * 32 + l31) * SP_LD + g * 8 + lh * 4]);
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    b[j] = *reinterpret_cast<const float4 *>(&sb[((wc * TJ + j) * 32 + l31) * SP_LD + g * 8 + lh * 4]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j) {
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
                for (int j = 0; j < TJ; ++j) {
                    const int n = n0 + (wc * TJ + j) * 32 + l31;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int64_t m = m0 + (wr * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        if (m < M && n < N) C[m * ldc + n] = acc[i][j][e];
                        if (PRO >= 2 && j == 0 && wc == 0 && n0 == 0 && m < M) {  // the columns nobody computes stay defined
                            if (l31 < pro.pad_lo) C[m * ldc - pro.pad_lo + l31] = 0.0f;
                            if (l31 < pro.pad_hi) C[m * ldc + N + l31] = 0.0f;
                        }
                    }
                }
            if (STATS != 0) {
                const int valid = (int)min((int64_t)64, max((int64_t)0, M - (m0 + wr * 64)));  // wave-uniform
                const float pad = (float)(64 - valid);
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
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
        for (int j = 0; j < TJ; ++j) {
            const int n = n0_first + (wc * TJ + j) * 32 + l31;
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

// Workgroups of a launch: two resident per CU (512); twice as many for long launches WITH the side output, i.e. inside a
// training step, so that the dispatcher hands the second half out as slots come free -- the pipelined step runs this
// kernel beside the next step's furthest-point sampling, whose 32 CUs have room for one workgroup of this kernel
// instead of two, and a static split over exactly-resident workgroups would wait for those (8.95 -> 8.90 ms/step; alone
// on the chip the extra pipeline fills cost the forward pass 0.1 ms, hence not for the launches without side output).
static int sp_grid(int64_t items, bool side) { return (side && items >= 2048) ? 1024 : 512; }

// The statistics chunks (and the kernel) need every workgroup to stay on one column tile: grid % (8 * tiles_n) == 0.
// Not served (0): widths that end in a narrow remainder (192, 320 ... columns: gemm_rows.hip mixes tile widths there; up
// to 64 columns run on this kernel's 128 x 64 tiles), fewer than 512 items, contractions longer than the constants' LDS
// table.
static int sp_tiles_n(int64_t M, int N, int K, int kmax = tp3d::SP_PRO_KMAX, bool fixed_tile = true)
{
    if (M <= 0 || N <= 0 || K < 4 || (K & 3) || K > kmax) return 0;
    const int rem = N % tp3d::SP_BN;
    const int tiles_n = (N + tp3d::SP_BN - 1) / tp3d::SP_BN;
    if (fixed_tile) {  // (the forward form: statistics per workgroup; a narrow remainder tile is gemm_rows.hip's case)
        if (rem > 0 && rem <= 64 && N > 64) return 0;  // (N <= 64: one 128 x 64 tile per row block)
        if (512 % (8 * tiles_n)) return 0;
    }
    const int64_t row_blocks = (M + tp3d::SP_BM - 1) / tp3d::SP_BM;
    if ((row_blocks + 7) / 8 * 8 * tiles_n < 512) return 0;
    return tiles_n;
}
static int64_t sp_items(int64_t M, int tiles_n) { return ((M + tp3d::SP_BM - 1) / tp3d::SP_BM + 7) / 8 * 8 * tiles_n; }

// Statistics chunks tp3d_gemm_rows_bnact_sp_f32 writes for (M, N, K) with / without the side output -- stat_partial holds
// chunks * 4 * N floats in the layout tp3d_bn_finalize_f32 reads -- or 0 when the shape is not served by this kernel.
TP3D_EXPORT int tp3d_gemm_rows_sp_chunks(int64_t M, int N, int K, int with_act_out)
{
    const int tiles_n = sp_tiles_n(M, N, K);
    return tiles_n ? 2 * (sp_grid(sp_items(M, tiles_n), with_act_out != 0) / tiles_n) : 0;
}

TP3D_EXPORT int tp3d_gemm_rows_bnact_sp_f32(const float *Y, const float *mean, const float *scale, const float *beta,
                                            float slope, const float *Bt, int64_t M, int N, int K, float *C,
                                            float *stat_partial, float *act_out, int reverse, void *stream)
{
    using namespace tp3d;
    const int tiles_n = sp_tiles_n(M, N, K);
    if (!tiles_n || !Y || !mean || !scale || !beta || !Bt || !C) return TP3D_E_BADARG;
    const int64_t items = sp_items(M, tiles_n);
    const int grid = sp_grid(items, act_out != nullptr);
    SpPro pro{mean, scale, beta, slope, nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    hipStream_t s = (hipStream_t)stream;
#define TP3D_SP_LAUNCH(STATS, BN)                                                                                     \
    hipLaunchKernelGGL((gemm_rows_sp_kernel<STATS, 1, BN>), dim3(grid), dim3(SP_BLOCK), 0, s, Y, Bt, M, N, K, tiles_n, items, \
                       C, (int64_t)N, stat_partial, act_out, pro, reverse)
    if (N <= 64) {
        if (stat_partial)
            TP3D_SP_LAUNCH(2, 64);
        else
            TP3D_SP_LAUNCH(0, 64);
    } else {
        if (stat_partial)
            TP3D_SP_LAUNCH(2, 128);
        else
            TP3D_SP_LAUNCH(0, 128);
    }
#undef TP3D_SP_LAUNCH
    return check_launch();
}

// 1 when tp3d_gemm_rows_bnbwd_sp_f32 serves (M, N, K): the shape rule of the forward form with K <= 256.
TP3D_EXPORT int tp3d_gemm_rows_bnbwd_sp_serves(int64_t M, int N, int K)
{
    return sp_tiles_n(M, N, K, tp3d::SP_BWD_KMAX, true) ? 1 : 0;
}

TP3D_EXPORT int tp3d_gemm_rows_bnbwd_sp_f32(const float *Y, const float *dA, const float *mean, const float *scale,
                                            const float *beta, const float *c1, const float *c2, float slope,
                                            const float *Bt, int64_t M, int N, int K, float *C, int ldc, int pad_lo,
                                            int pad_hi, float *dY_out, const int *argmax, int ns, int reverse, void *stream)
{
    using namespace tp3d;
    // (no statistics here, so any number of column tiles would work -- but every column tile re-reads BOTH operand
    // streams: with three tiles (N = 320) the library GEMM behind the apply pass is faster, 0.76 vs 0.55 ms on config 3)
    const int tiles_n = sp_tiles_n(M, N, K, SP_BWD_KMAX, true);
    if (!tiles_n || ldc < N || !Y || !dA || !mean || !scale || !beta || !c1 || !c2 || !Bt || !C) return TP3D_E_BADARG;
    if (pad_lo < 0 || pad_hi < 0 || pad_lo > 32 || pad_hi > 32 || pad_lo + N + pad_hi > ldc)
        return TP3D_E_BADARG;
    if (argmax && (ns < 64 || (ns & (ns - 1)) || M % ns)) return TP3D_E_BADARG;  // groups of 64, 128, ... rows
    int ns_shift = 0;
    while (argmax && (1 << ns_shift) < ns) ++ns_shift;
    const int64_t items = sp_items(M, tiles_n);
    const int grid = sp_grid(items, true);
    SpPro pro{mean, scale, beta, slope, dA, c1, c2, argmax, ns_shift, pad_lo, pad_hi};
    hipStream_t s = (hipStream_t)stream;
#define TP3D_SP_BWD2(PRO, BN)                                                                                         \
    hipLaunchKernelGGL((gemm_rows_sp_kernel<0, PRO, BN>), dim3(grid), dim3(SP_BLOCK), 0, s, Y, Bt, M, N, K, tiles_n, items, C, \
                       (int64_t)ldc, (float *)nullptr, dY_out, pro, reverse)
    if (argmax) {
        if (N <= 64)
            TP3D_SP_BWD2(3, 64);
        else
            TP3D_SP_BWD2(3, 128);
    } else {
        if (N <= 64)
            TP3D_SP_BWD2(2, 64);
        else
            TP3D_SP_BWD2(2, 128);
    }
#undef TP3D_SP_BWD2
    return check_launch();
}
