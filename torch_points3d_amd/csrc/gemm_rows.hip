// Tall-skinny fp32 GEMM of the shared MLPs, with the BatchNorm statistics of its output folded into it:
//     C[M,N] = A[M,K] * Bt[N,K]^T       M = B*npoint*nsample rows (up to ~1e6), N, K <= ~1500, K contiguous in both
// (the forms that apply the neighbouring BatchNorm / activation passes to the A operand while it is staged live in
//  gemm_rows_sp.hip / gemm_rows_x3.hip, where dedicated loader waves do that work; the first design -- prologues in the
//  MFMA waves of this kernel, 230-256 VGPRs -- was measured slower, DESIGN.md section 5, and is gone)
// EPILOGUE (STATS): per column, shifted sums of C in `chunks` partial rows -> tp3d_bn_finalize_f32
//     (one chunk per (128-row block, wave row); launches that fill the persistent grid keep one running chunk per
//     (workgroup, wave row) instead: wave_rows*1024/tiles_n chunks whatever M is, so the finalize pass stays small)
// Reference semantics: Conv2d 1x1 (bias=False) -> BatchNorm2d (training) -> LeakyReLU, forward and autograd backward
// (torch_points3d/core/common_modules/dense_modules.py:5-12,25-29).
//
// 4 waves per workgroup; two tile shapes: 128 x 128 (waves 2 x 2, each 2 x 2 MFMA tiles of 32 x 32) and 128 x 64 (waves
// 4 x 1, each 1 x 2 tiles) for layers of 64 or fewer output columns (or whose width leaves such a remainder).
// v_mfma_f32_32x32x2_f32, exact fp32.  K is walked in steps of 32 through LDS, the global loads of step i+1 in flight
// during the MFMAs of step i; both operands are staged as [row][k] with a 36-float pitch: float4 stores stay aligned and
// the ds_read_b128 operand fetch (lane = row) is bank-conflict free (36*r mod 64 hits 16 disjoint 4-bank slots).
// Which physical k feeds which MFMA k-slot is free as long as A and B agree: in every group of 8 k's the lower
// half-wave takes k0..k0+3 and the upper half-wave k0+4..k0+7, so one 16-byte LDS read feeds four MFMAs.
// Long contractions with few output tiles (the 4096-row global / decoder layers, K up to 1536) are split over
// gridDim.y K-ranges into partial slabs that a second kernel sums in fixed order: more workgroups, and a two-level
// summation whose round-off is ~3x smaller than one sequential chain over K.
#include <algorithm>

#include "tp3d_common.h"

namespace tp3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GR_BLOCK_T = 256;
constexpr int GR_BM = 128, GR_BK = 32;
constexpr int GR_LD = GR_BK + 4;  // 36-float pitch

// Output EPILOGUE of the eval-mode layers (mean == nullptr: none): C = LeakyReLU((acc - mean[n]) * scale[n] + beta[n]) --
// BatchNorm on running statistics + activation applied to the accumulators, so that an inference pass has one launch per
// Linear -> BatchNorm -> LeakyReLU layer instead of two (core/common_modules/base_modules.py FastBatchNorm1d + activation)
struct RowsEpilogue {
    const float *mean, *scale, *beta;
    float slope;
};

// WIDE: 128 x 128 tile, else 128 x 64.  STATS: 0 none, 1 one statistics chunk per (128-row block, wave row), 2 one per
// (workgroup, wave row) (needs gridDim.x % (8*tiles_n) == 0: every item of a workgroup then lies in one column tile)
template <bool WIDE, int STATS>
__global__ __launch_bounds__(GR_BLOCK_T, 1) void gemm_rows_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                                                                int64_t M, int N, int K, int kchunk, int tiles_n,
                                                                int64_t items, float *__restrict__ C,
                                                                float *__restrict__ partial, RowsEpilogue epi)
{
    constexpr int BN = WIDE ? 128 : 64;
    constexpr int WR = WIDE ? 2 : 4;          // wave rows of the workgroup tile
    constexpr int WM = WIDE ? 2 : 1;          // 32-row MFMA tiles per wave
    constexpr int WN = 2;                     // 32-column MFMA tiles per wave
    constexpr int PB = BN * (GR_BK / 4) / GR_BLOCK_T;  // float4 slots of the B tile per thread (4 or 2)
    __shared__ __attribute__((aligned(16))) float sA[GR_BM * GR_LD];
    __shared__ __attribute__((aligned(16))) float sB[BN * GR_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = WIDE ? wave >> 1 : wave, wc = WIDE ? wave & 1 : 0;
    const int l31 = lane & 31, lh = lane >> 5;
    const int kbeg = blockIdx.y * kchunk, kend = min(K, kbeg + kchunk);
    const int ksteps = (kend - kbeg + GR_BK - 1) / GR_BK;
    if (blockIdx.y > 0) C += (size_t)blockIdx.y * (size_t)M * N;  // K-split: one partial slab per K-range

    // Work items = (row block, column tile) in an XCD-aware order: the column tiles of one row block are 8 ids
    // apart, i.e. on the same XCD (shared L2 for A).  Workgroups are PERSISTENT: each walks items id, id + G, ...
    // (G a multiple of 8) as one flat sequence of K-steps, so the loads of the next item's first K-step are already
    // in flight while this item's epilogue stores drain -- no exposed prologue/epilogue latency per tile.
    auto decode = [&](int64_t item, int64_t &m0, int &n0, int64_t &rb) __attribute__((always_inline)) {
        const int64_t grp = item / (8 * tiles_n);
        const int rem = (int)(item % (8 * tiles_n));
        rb = grp * 8 + (rem & 7);
        m0 = rb * GR_BM;
        n0 = (rem >> 3) * BN;
    };

    // staging registers: the A tile is 128 rows x 32 k = 1024 float4 (4 per thread), written to LDS one K-step later (the
    // loads have had a whole MFMA phase to land by then)
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ra[4] = {zero4, zero4, zero4, zero4};
    float4 rb0 = zero4, rb1 = zero4, rb2 = zero4, rb3 = zero4;  // B tile staging (named: an array of them ended up in scratch)
    const int k4 = (tid & 7) * 4;  // this thread's k offset inside a K-step (tid + i*256: the same for all four slots)
    auto fetch = [&](int64_t m0, int n0, int k0) __attribute__((always_inline)) {
        const bool kin = k0 + k4 < kend;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + i * 32;
            const int64_t m = m0 + row;
            const bool in = m < M && kin;
            ra[i] = in ? *reinterpret_cast<const float4 *>(A + m * K + k0 + k4) : zero4;
        }
        auto loadB = [&](int i) __attribute__((always_inline)) -> float4 {
            const int n = n0 + (tid >> 3) + i * 32;
            return (n < N && kin) ? *reinterpret_cast<const float4 *>(Bt + (size_t)n * K + k0 + k4) : zero4;
        };
        rb0 = loadB(0);
        rb1 = loadB(1);
        if (PB > 2) {
            rb2 = loadB(2);
            rb3 = loadB(3);
        }
    };
    int64_t item = blockIdx.x;
    if (item >= items) return;
    int64_t m0, rb;
    int n0;
    decode(item, m0, n0, rb);
    fetch(m0, n0, kbeg);

    const int n0_first = n0;
    float run1[2] = {0.0f, 0.0f}, run2[2] = {0.0f, 0.0f};  // STATS == 2: this thread's share over all its items
    float kshift[2] = {0.0f, 0.0f};
    bool have_shift = false;
    int run_rows = 0;
    f32x16 acc[WM][WN];
    while (item < items) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        const int64_t next_item = item + gridDim.x;
        int64_t nm0 = 0, nrb = 0;
        int nn0 = 0;
        if (next_item < items) decode(next_item, nm0, nn0, nrb);

        for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (tid >> 3) + i * 32;
                *reinterpret_cast<float4 *>(&sA[row * GR_LD + k4]) = ra[i];
            }
            *reinterpret_cast<float4 *>(&sB[((tid >> 3) + 0 * 32) * GR_LD + k4]) = rb0;
            *reinterpret_cast<float4 *>(&sB[((tid >> 3) + 1 * 32) * GR_LD + k4]) = rb1;
            if (PB > 2) {
                *reinterpret_cast<float4 *>(&sB[((tid >> 3) + 2 * 32) * GR_LD + k4]) = rb2;
                *reinterpret_cast<float4 *>(&sB[((tid >> 3) + 3 * 32) * GR_LD + k4]) = rb3;
            }
            __syncthreads();
            if (ks + 1 < ksteps) fetch(m0, n0, kbeg + (ks + 1) * GR_BK);
            else if (next_item < items) fetch(nm0, nn0, kbeg);  // next item's first K-step rides under this epilogue
#pragma unroll
            for (int g = 0; g < GR_BK / 8; ++g) {
                float4 a[WM], b[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    a[i] = *reinterpret_cast<const float4 *>(&sA[((wr * WM + i) * 32 + l31) * GR_LD + g * 8 + lh * 4]);
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    b[j] = *reinterpret_cast<const float4 *>(&sB[((wc * WN + j) * 32 + l31) * GR_LD + g * 8 + lh * 4]);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                    }
            }
            __syncthreads();
        }

        // ---- epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
        constexpr int WROWS = WM * 32;  // rows of the tile one wave owns
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = n0 + (wc * WN + j) * 32 + l31;
                float emu = 0.0f, esc = 1.0f, ebe = 0.0f;
                if (epi.mean && n < N) emu = epi.mean[n], esc = epi.scale[n], ebe = epi.beta[n];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t m = m0 + (wr * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    float v = acc[i][j][e];
                    if (epi.mean) {  // (wave-uniform)
                        v = (v - emu) * esc + ebe;
                        v = v > 0.0f ? v : v * epi.slope;
                    }
                    if (m < M && n < N) C[m * N + n] = v;
                }
            }
        if (STATS != 0) {
            // Column statistics of this wave's part of the tile as SHIFTED sums (d = v - K; K = the first value the
            // wave saw in that column): free of the cancellation E[y^2] - E[y]^2 suffers when |mean| >> std.
            // Rows past M were staged as zeros, i.e. each of the `pad` such rows of this wave added d = -K: taken out again
            // below (only the last row block of a matrix has any).  STATS == 1: one chunk per (row block, wave row);
            // STATS == 2: the sums run on over all items of the workgroup, one chunk per (workgroup, wave row).
            const int valid = (int)min((int64_t)WROWS, max((int64_t)0, M - (m0 + wr * WROWS)));  // wave-uniform
            const float pad = (float)(WROWS - valid);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (STATS == 1 || !have_shift) kshift[j] = __shfl(acc[0][j][0], l31);  // first row of this wave's part
                const float k = kshift[j];
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float d = acc[i][j][e] - k;
                        s1 += d;
                        s2 += d * d;
                    }
                s1 += __shfl_xor(s1, 32);  // the two half-waves hold different rows of the same column
                s2 += __shfl_xor(s2, 32);
                s1 += pad * k;
                s2 -= pad * (k * k);
                if (STATS == 2) {
                    run1[j] += s1;  // (both half-waves now hold the wave's sum; written once below)
                    run2[j] += s2;
                } else {
                    const int n = n0 + (wc * WN + j) * 32 + l31;
                    if (lh == 0 && n < N && m0 < M) {  // padding items (row block past M) own no statistics rows
                        float *pr = partial + ((size_t)(rb * WR + wr) * 4) * N + n;
                        pr[0] = s1;
                        pr[(size_t)N] = s2;
                        pr[(size_t)2 * N] = k;
                        pr[(size_t)3 * N] = (float)valid;
                    }
                }
            }
            if (STATS == 2) {
                if (valid > 0) have_shift = true;  // a wave whose first items were padding keeps looking for a shift
                run_rows += valid;
            }
        }
        item = next_item;
        m0 = nm0;
        n0 = nn0;
        rb = nrb;
    }
    if (STATS == 2) {
        const int per = 8 * tiles_n;
        const int64_t slot = (int64_t)(blockIdx.x / per) * 8 + (blockIdx.x & 7);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0_first + (wc * WN + j) * 32 + l31;
            if (lh == 0 && n < N) {
                float *pr = partial + ((size_t)(slot * WR + wr) * 4) * N + n;
                pr[0] = run1[j];
                pr[(size_t)N] = run2[j];
                pr[(size_t)2 * N] = kshift[j];
                pr[(size_t)3 * N] = (float)run_rows;
            }
        }
    }
}

// The plain 128 x 128 kernel (no prologue, no K-split): the forward GEMM of the wide layers, kept as its own
// instantiation because it is the one that has to fit two waves per SIMD (182 VGPRs + 64 accumulators, no scratch).
// STATS: 0 none, 1 one statistics chunk per (128-row block, wave row), 2 one per (workgroup, wave row) (needs
// gridDim.x % (8*tiles_n) == 0: every item of a workgroup then lies in the same column tile)
template <int STATS>
__global__ __launch_bounds__(GR_BLOCK_T, 2) void gemm_rows_wide_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                                                                int64_t M, int N, int K, int tiles_n, int64_t items,
                                                                float *__restrict__ C, float *__restrict__ partial,
                                                                RowsEpilogue epi)
{
    __shared__ __attribute__((aligned(16))) float sA[GR_BM * GR_LD];
    __shared__ __attribute__((aligned(16))) float sB[128 * GR_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ksteps = (K + GR_BK - 1) / GR_BK;
    const int tail_groups = (K - (ksteps - 1) * GR_BK + 7) / 8;  // 8-wide k groups of the last step that hold columns

    // Work items = (row block, column tile) in an XCD-aware order: the column tiles of one row block are 8 ids
    // apart, i.e. on the same XCD (shared L2 for A).  Workgroups are PERSISTENT: each walks items id, id + G, ...
    // (G a multiple of 8) as one flat sequence of K-steps, so the loads of the next item's first K-step are already
    // in flight while this item's epilogue stores drain -- no exposed prologue/epilogue latency per tile.
    auto decode = [&](int64_t item, int64_t &m0, int &n0, int64_t &rb) {
        const int64_t grp = item / (8 * tiles_n);
        const int rem = (int)(item % (8 * tiles_n));
        rb = grp * 8 + (rem & 7);
        m0 = rb * GR_BM;
        n0 = (rem >> 3) * 128;
    };

    // staging registers: each tile is 128 rows x 32 k = 1024 float4 (4 per thread and operand)
    float4 ra[4], rbv[4];
    auto fetch = [&](int64_t m0, int n0, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * GR_BLOCK_T;
            const int row = e >> 3, k4 = (e & 7) * 4;  // 8 float4 per row
            const int64_t m = m0 + row;
            ra[i] = (m < M && k0 + k4 < K) ? *reinterpret_cast<const float4 *>(A + m * K + k0 + k4)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
            const int n = n0 + row;
            rbv[i] = (n < N && k0 + k4 < K) ? *reinterpret_cast<const float4 *>(Bt + (size_t)n * K + k0 + k4)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    int64_t item = blockIdx.x;
    if (item >= items) return;
    int64_t m0, rb;
    int n0;
    decode(item, m0, n0, rb);
    fetch(m0, n0, 0);

    const int n0_first = n0;
    float run1[2] = {0.0f, 0.0f}, run2[2] = {0.0f, 0.0f};  // STATS == 2: this thread's share over all its items
    float kshift[2] = {0.0f, 0.0f};
    bool have_shift = false;
    int run_rows = 0;
    f32x16 acc[2][2];
    while (item < items) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
        const int64_t next_item = item + gridDim.x;
        int64_t nm0 = 0, nrb = 0;
        int nn0 = 0;
        if (next_item < items) decode(next_item, nm0, nn0, nrb);

        for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = tid + i * GR_BLOCK_T;
                const int o = (e >> 3) * GR_LD + (e & 7) * 4;
                *reinterpret_cast<float4 *>(&sA[o]) = ra[i];
                *reinterpret_cast<float4 *>(&sB[o]) = rbv[i];
            }
            __syncthreads();
            if (ks + 1 < ksteps) fetch(m0, n0, (ks + 1) * GR_BK);
            else if (next_item < items) fetch(nm0, nn0, 0);  // next item's first K-step rides under this epilogue
            // last step of a contraction that is not a multiple of 32 (131 + 1 padding columns, 3 + 5, ...): only the
            // 8-wide k groups that hold real columns -- a fifth step of 4 columns costs a quarter step, not a whole one
            const int ng = (ks + 1 < ksteps) ? GR_BK / 8 : tail_groups;  // wave-uniform
#pragma unroll
            for (int g = 0; g < GR_BK / 8; ++g) {
                if (g < ng) {
                    float4 a[2], b[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        a[i] = *reinterpret_cast<const float4 *>(&sA[((wr * 2 + i) * 32 + l31) * GR_LD + g * 8 + lh * 4]);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        b[j] = *reinterpret_cast<const float4 *>(&sB[((wc * 2 + j) * 32 + l31) * GR_LD + g * 8 + lh * 4]);
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
            __syncthreads();
        }

        // ---- epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + (wc * 2 + j) * 32 + l31;
                float emu = 0.0f, esc = 1.0f, ebe = 0.0f;
                if (epi.mean && n < N) emu = epi.mean[n], esc = epi.scale[n], ebe = epi.beta[n];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t m = m0 + (wr * 2 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    float v = acc[i][j][e];
                    if (epi.mean) {  // (wave-uniform)
                        v = (v - emu) * esc + ebe;
                        v = v > 0.0f ? v : v * epi.slope;
                    }
                    if (m < M && n < N) C[m * N + n] = v;
                }
            }
        if (STATS != 0) {
            // Column statistics of this wave's 64 x 64 part of the tile as SHIFTED sums (d = v - K; K = the first value
            // the wave saw in that column): free of the cancellation E[y^2] - E[y]^2 suffers when |mean| >> std.
            // Rows past M were staged as zeros, i.e. each of the `pad` such rows of this wave added d = -K: taken out again
            // below (only the last row block of a matrix has any).  STATS == 1: one chunk per (row block, wave row);
            // STATS == 2: the sums run on over all items of the workgroup, one chunk per (workgroup, wave row).
            const int valid = (int)min((int64_t)64, max((int64_t)0, M - (m0 + wr * 64)));  // wave-uniform
            const float pad = (float)(64 - valid);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (STATS == 1 || !have_shift) kshift[j] = __shfl(acc[0][j][0], l31);  // first row of this wave's part
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
                s1 += __shfl_xor(s1, 32);  // the two half-waves hold different rows of the same column
                s2 += __shfl_xor(s2, 32);
                s1 += pad * k;
                s2 -= pad * (k * k);
                if (STATS == 2) {
                    run1[j] += s1;  // (both half-waves now hold the wave's sum; written once below)
                    run2[j] += s2;
                } else {
                    const int n = n0 + (wc * 2 + j) * 32 + l31;
                    if (lh == 0 && n < N && m0 < M) {  // padding items (row block past M) own no statistics rows
                        float *pr = partial + ((size_t)(rb * 2 + wr) * 4) * N + n;
                        pr[0] = s1;
                        pr[(size_t)N] = s2;
                        pr[(size_t)2 * N] = k;
                        pr[(size_t)3 * N] = (float)valid;
                    }
                }
            }
            if (STATS == 2) {
                if (valid > 0) have_shift = true;  // a wave whose first items were padding keeps looking for a shift
                run_rows += valid;
            }
        }
        item = next_item;
        m0 = nm0;
        n0 = nn0;
        rb = nrb;
    }
    if (STATS == 2) {
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


// out[e] = sum_s slab[s][e], s ascending (fixed order => reproducible); one float4 per thread
__global__ __launch_bounds__(256) void gemm_rows_sum_slabs_kernel(const float *__restrict__ slabs, int S, int64_t MN4,
                                                                   float *__restrict__ out, int N, RowsEpilogue epi)
{
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < MN4; e += (int64_t)gridDim.x * 256) {
        float4 t = reinterpret_cast<const float4 *>(slabs)[e];
        for (int s = 1; s < S; ++s) {
            const float4 v = reinterpret_cast<const float4 *>(slabs + (size_t)s * MN4 * 4)[e];
            t.x += v.x;
            t.y += v.y;
            t.z += v.z;
            t.w += v.w;
        }
        if (epi.mean) {  // (N % 4 == 0 on this path: the four values share a row)
            const int n = (int)((e * 4) % N);
            const float4 mu = *reinterpret_cast<const float4 *>(epi.mean + n), sc = *reinterpret_cast<const float4 *>(epi.scale + n);
            const float4 be = *reinterpret_cast<const float4 *>(epi.beta + n);
            t.x = (t.x - mu.x) * sc.x + be.x, t.y = (t.y - mu.y) * sc.y + be.y;
            t.z = (t.z - mu.z) * sc.z + be.z, t.w = (t.w - mu.w) * sc.w + be.w;
            t.x = t.x > 0.0f ? t.x : t.x * epi.slope, t.y = t.y > 0.0f ? t.y : t.y * epi.slope;
            t.z = t.z > 0.0f ? t.z : t.z * epi.slope, t.w = t.w > 0.0f ? t.w : t.w * epi.slope;
        }
        reinterpret_cast<float4 *>(out)[e] = t;
    }
}

}  // namespace tp3d

using namespace tp3d;

namespace {
constexpr int GR_GRID = 1024;  // persistent: 4 workgroups per CU, a multiple of 8
struct RowsPlan {
    bool wide;  // 128 x 128 tiles (else 128 x 64)
    int tiles_n, wave_rows;
    int64_t row_blocks, items, blocks;
    bool per_workgroup;  // statistics chunks: per (workgroup, wave row) instead of per (128-row block, wave row)
    int64_t chunks;
    int ksplit, kchunk;  // K-ranges of a split launch (1 = no split)
};
RowsPlan rows_plan(int64_t M, int N, int K = 0, bool allow_split = false)
{
    RowsPlan p;
    // narrow tiles when they cover N with less padding (N <= 64, or a remainder of at most 64 columns: 192, 320 ...)
    const int rem = N % 128;
    p.wide = !(rem > 0 && rem <= 64);
    const int bn = p.wide ? 128 : 64;
    p.wave_rows = p.wide ? 2 : 4;
    p.tiles_n = (N + bn - 1) / bn;
    p.row_blocks = (M + GR_BM - 1) / GR_BM;
    const int64_t groups = (p.row_blocks + 7) / 8;
    p.items = groups * 8 * p.tiles_n;  // items past the last row block stage zeros and store nothing
    p.blocks = p.items < GR_GRID ? p.items : GR_GRID;
    p.per_workgroup = p.blocks == GR_GRID && GR_GRID % (8 * p.tiles_n) == 0;
    p.chunks = p.wave_rows * (p.per_workgroup ? GR_GRID / p.tiles_n : p.row_blocks);
    p.ksplit = 1;
    p.kchunk = K > 0 ? (K + GR_BK - 1) / GR_BK * GR_BK : 0;
    if (allow_split && K >= 1024 && p.items <= 128) {
        // few output tiles and a very long contraction (the 1280 / 1536-channel decoder layers): K-ranges of >= 128
        // (every range writes an M x N slab that the summing pass reads back, so no more ranges than needed)
        const int want = (int)((512 + p.items - 1) / p.items);
        const int by_k = K / 128;
        int s = want < by_k ? want : by_k;
        if (s > 16) s = 16;
        if (s > 1) {
            p.kchunk = ((K + s - 1) / s + GR_BK - 1) / GR_BK * GR_BK;
            p.ksplit = (K + p.kchunk - 1) / p.kchunk;
        }
    }
    return p;
}

template <bool WIDE>
int launch_rows(const RowsPlan &p, const float *A, const float *Bt, int64_t M, int N, int K, float *C, float *stat_partial,
                const RowsEpilogue &epi, hipStream_t s)
{
    const dim3 grid((unsigned)p.blocks, (unsigned)p.ksplit), block(GR_BLOCK_T);
    if (WIDE && p.ksplit == 1) {  // the dedicated plain 128 x 128 kernel
        if (stat_partial && p.per_workgroup)
            hipLaunchKernelGGL(gemm_rows_wide_kernel<2>, grid, block, 0, s, A, Bt, M, N, K, p.tiles_n, p.items, C, stat_partial, epi);
        else if (stat_partial)
            hipLaunchKernelGGL(gemm_rows_wide_kernel<1>, grid, block, 0, s, A, Bt, M, N, K, p.tiles_n, p.items, C, stat_partial, epi);
        else
            hipLaunchKernelGGL(gemm_rows_wide_kernel<0>, grid, block, 0, s, A, Bt, M, N, K, p.tiles_n, p.items, C, stat_partial, epi);
        return check_launch();
    }
    if (stat_partial && p.per_workgroup)
        hipLaunchKernelGGL((gemm_rows_kernel<WIDE, 2>), grid, block, 0, s, A, Bt, M, N, K, p.kchunk, p.tiles_n, p.items,
                           C, stat_partial, epi);
    else if (stat_partial)
        hipLaunchKernelGGL((gemm_rows_kernel<WIDE, 1>), grid, block, 0, s, A, Bt, M, N, K, p.kchunk, p.tiles_n, p.items,
                           C, stat_partial, epi);
    else
        hipLaunchKernelGGL((gemm_rows_kernel<WIDE, 0>), grid, block, 0, s, A, Bt, M, N, K, p.kchunk, p.tiles_n, p.items,
                           C, stat_partial, epi);
    return check_launch();
}

int launch_rows_any(const RowsPlan &p, const float *A, const float *Bt, int64_t M, int N, int K, float *C,
                    float *stat_partial, const RowsEpilogue &epi, hipStream_t s)
{
    return p.wide ? launch_rows<true>(p, A, Bt, M, N, K, C, stat_partial, epi, s)
                  : launch_rows<false>(p, A, Bt, M, N, K, C, stat_partial, epi, s);
}

// the contraction, optionally K-split into slabs + their sum, with an optional output epilogue
int rows_gemm(const float *A, const float *Bt, int64_t M, int N, int K, float *C, float *stat_partial, float *workspace,
              const RowsEpilogue &epi, hipStream_t s)
{
    const RowsEpilogue none = {nullptr, nullptr, nullptr, 1.0f};
    // K-split only without fused statistics (they need the finished column values) and with slabs to write to
    const RowsPlan p = rows_plan(M, N, K, !stat_partial && workspace && ((M * (int64_t)N) & 3) == 0 && (!epi.mean || (N & 3) == 0));
    if (p.ksplit == 1) return launch_rows_any(p, A, Bt, M, N, K, C, stat_partial, epi, s);
    if (int rc = launch_rows_any(p, A, Bt, M, N, K, workspace, nullptr, none, s)) return rc;
    const int64_t mn4 = M * (int64_t)N / 4;
    hipLaunchKernelGGL(gemm_rows_sum_slabs_kernel, dim3((unsigned)std::min<int64_t>((mn4 + 255) / 256, 4096)), dim3(256), 0, s,
                       workspace, p.ksplit, mn4, C, N, epi);
    return check_launch();
}
}  // namespace

// size of the statistics buffer: 4 rows of N floats per chunk (sum d, sum d^2, shift, rows); chunks = one per
// (128-row block, wave row) or per (persistent workgroup, wave row) -- whichever the launch uses, both covered
TP3D_EXPORT size_t tp3d_gemm_rows_stat_floats(int64_t M, int N)
{
    if (M <= 0 || N <= 0) return 0;
    const RowsPlan p = rows_plan(M, N);
    const int64_t by_block = p.wave_rows * p.row_blocks;
    const int64_t rows = p.chunks > by_block ? p.chunks : by_block;
    return (size_t)rows * 4 * (size_t)N;
}

// number of statistics chunks tp3d_gemm_rows_f32 writes for this shape = `chunks` of tp3d_bn_finalize_f32
TP3D_EXPORT int tp3d_gemm_rows_stat_chunks(int64_t M, int N)
{
    if (M <= 0 || N <= 0) return 0;
    return (int)rows_plan(M, N).chunks;
}

// floats of the K-split slabs tp3d_gemm_rows_f32 needs as `workspace` when it runs WITHOUT statistics (0: no split)
TP3D_EXPORT size_t tp3d_gemm_rows_workspace_floats(int64_t M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const RowsPlan p = rows_plan(M, N, K, true);
    return p.ksplit > 1 ? (size_t)p.ksplit * (size_t)M * (size_t)N : 0;
}

// plan[0..8] = column tiles, row blocks, work items, workgroups, statistics chunks written, 1 when they are one per
// workgroup (else per 128-row block), wave rows per tile, K-ranges of the split launch, columns of a tile
TP3D_EXPORT int tp3d_gemm_rows_plan(int64_t M, int N, int K, int64_t *plan)
{
    if (M <= 0 || N <= 0 || K < 0 || !plan) return TP3D_E_BADARG;
    const RowsPlan p = rows_plan(M, N, K, K > 0);
    plan[0] = p.tiles_n;
    plan[1] = p.row_blocks;
    plan[2] = p.items;
    plan[3] = p.blocks;
    plan[4] = p.chunks;
    plan[5] = p.per_workgroup ? 1 : 0;
    plan[6] = p.wave_rows;
    plan[7] = p.ksplit;
    plan[8] = p.wide ? 128 : 64;
    return TP3D_OK;
}

TP3D_EXPORT int tp3d_gemm_rows_f32(const float *A, const float *Bt, int64_t M, int N, int K, float *C, float *stat_partial,
                                   float *workspace, void *stream)
{
    if (M < 0 || N <= 0 || K <= 0 || (K & 3)) return TP3D_E_BADARG;  // operand rows must be 16-byte aligned
    if (M == 0) return TP3D_OK;
    if (!A || !Bt || !C) return TP3D_E_BADARG;
    const RowsEpilogue none = {nullptr, nullptr, nullptr, 1.0f};
    return rows_gemm(A, Bt, M, N, K, C, stat_partial, workspace, none, (hipStream_t)stream);
}

// The same contraction with the eval-mode BatchNorm + LeakyReLU of its OUTPUT applied to the accumulators:
//   C = LeakyReLU((A Bt^T - mean[n]) * scale[n] + beta[n])        (mean / scale / beta: N floats each)
// -- one launch per Linear -> BatchNorm (running statistics) -> activation layer of an inference pass.  workspace as for
// tp3d_gemm_rows_f32 (K-split slabs; the epilogue then runs in the slab sum).
TP3D_EXPORT int tp3d_gemm_rows_epi_f32(const float *A, const float *Bt, int64_t M, int N, int K, const float *mean,
                                       const float *scale, const float *beta, float slope, float *C, float *workspace,
                                       void *stream)
{
    if (M < 0 || N <= 0 || K <= 0 || (K & 3)) return TP3D_E_BADARG;
    if (M == 0) return TP3D_OK;
    if (!A || !Bt || !C || !mean || !scale || !beta) return TP3D_E_BADARG;
    const RowsEpilogue epi = {mean, scale, beta, slope};
    return rows_gemm(A, Bt, M, N, K, C, nullptr, workspace, epi, (hipStream_t)stream);
}
