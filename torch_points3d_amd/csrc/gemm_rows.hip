// Tall-skinny fp32 GEMM of the shared MLPs with the BatchNorm statistics fused into its epilogue:
//     C[M,N] = A[M,K] * Bt[N,K]^T       M = B*npoint*nsample rows (up to ~1e6), N, K <= ~1300, K contiguous in both
//     stats (optional): per column, sum and sum of squares of C in `chunks` partial rows -> tp3d_bn_finalize_f32
//     (one row per 128-row block; launches that fill the persistent grid keep one running row per workgroup instead:
//     1024 / tiles_n rows whatever M is, so the finalize pass stays small)
// Forward pass:  Y = rows @ W^T   (Bt = W as stored: Cout x Cin)  + column statistics of Y (saves a full read of Y)
// Reference semantics: Conv2d 1x1 (bias=False) followed by BatchNorm2d in training mode
// (torch_points3d/core/common_modules/dense_modules.py:5-12,25-29).
//
// 4 waves per workgroup as 2x2, each wave 2x2 MFMA tiles of 32x32 (v_mfma_f32_32x32x2_f32, exact fp32): a 128 x 128
// output tile per workgroup, K walked in steps of 32 through LDS, the global loads of step i+1 in flight during the
// MFMAs of step i.  Both operands are staged as [row][k] with a 36-float pitch: float4 stores stay aligned and the
// ds_read_b128 operand fetch (lane = row) is bank-conflict free (36*r mod 64 hits 16 disjoint 4-bank slots).
// Which physical k feeds which MFMA k-slot is free as long as A and B agree: in every group of 8 k's the lower
// half-wave takes k0..k0+3 and the upper half-wave k0+4..k0+7, so one 16-byte LDS read feeds four MFMAs.
#include "tp3d_common.h"

namespace tp3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GR_BLOCK_T = 256;
constexpr int GR_BM = 128, GR_BN = 128, GR_BK = 32;
constexpr int GR_LD = GR_BK + 4;  // 36-float pitch

// STATS: 0 none, 1 one statistics row per 128-row block, 2 one row per workgroup (needs gridDim.x % (8*tiles_n) == 0:
// every item of a workgroup then lies in the same column tile)
template <int STATS>
__global__ __launch_bounds__(GR_BLOCK_T) void gemm_rows_kernel(const float *__restrict__ A, const float *__restrict__ Bt,
                                                                int64_t M, int N, int K, int tiles_n, int64_t items,
                                                                float *__restrict__ C, float *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float sA[GR_BM * GR_LD];
    __shared__ __attribute__((aligned(16))) float sB[GR_BN * GR_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ksteps = (K + GR_BK - 1) / GR_BK;

    // Work items = (row block, column tile) in an XCD-aware order: the column tiles of one row block are 8 ids
    // apart, i.e. on the same XCD (shared L2 for A).  Workgroups are PERSISTENT: each walks items id, id + G, ...
    // (G a multiple of 8) as one flat sequence of K-steps, so the loads of the next item's first K-step are already
    // in flight while this item's epilogue stores drain -- no exposed prologue/epilogue latency per tile.
    auto decode = [&](int64_t item, int64_t &m0, int &n0, int64_t &rb) {
        const int64_t grp = item / (8 * tiles_n);
        const int rem = (int)(item % (8 * tiles_n));
        rb = grp * 8 + (rem & 7);
        m0 = rb * GR_BM;
        n0 = (rem >> 3) * GR_BN;
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
#pragma unroll
            for (int g = 0; g < GR_BK / 8; ++g) {
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
        if (STATS != 0) {
            // Column statistics of this wave's 64 x 64 part of the tile as SHIFTED sums (d = v - K; K = the first value
            // the wave saw in that column): free of the cancellation E[y^2] - E[y]^2 suffers when |mean| >> std.
            // Rows past M were staged as zeros and are masked out.  STATS == 1: one chunk per (row block, wave row);
            // STATS == 2: the sums run on over all items of the workgroup, one chunk per (workgroup, wave row).
            const int64_t mrow0 = m0 + wr * 64;
            const int valid = (int)min((int64_t)64, max((int64_t)0, M - mrow0));  // rows of this wave inside the matrix
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (STATS == 1 || !have_shift) kshift[j] = __shfl(acc[0][j][0], l31);  // row mrow0 of this column
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int r = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                        const float d = acc[i][j][e] - kshift[j];
                        if (r < valid) {
                            s1 += d;
                            s2 += d * d;
                        }
                    }
                if (STATS == 2) {
                    run1[j] += s1;
                    run2[j] += s2;
                } else {
                    s1 += __shfl_xor(s1, 32);  // the two half-waves hold different rows of the same column
                    s2 += __shfl_xor(s2, 32);
                    const int n = n0 + (wc * 2 + j) * 32 + l31;
                    if (lh == 0 && n < N && m0 < M) {  // padding items (row block past M) own no statistics rows
                        float *pr = partial + ((size_t)(rb * 2 + wr) * 4) * N + n;
                        pr[0] = s1;
                        pr[(size_t)N] = s2;
                        pr[(size_t)2 * N] = kshift[j];
                        pr[(size_t)3 * N] = (float)valid;
                    }
                }
            }
            if (STATS == 2) {
                if (valid > 0) have_shift = true;  // (wave-uniform) a wave whose first items were padding keeps looking
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
            const float s1 = run1[j] + __shfl_xor(run1[j], 32), s2 = run2[j] + __shfl_xor(run2[j], 32);
            const int n = n0_first + (wc * 2 + j) * 32 + l31;
            if (lh == 0 && n < N) {
                float *pr = partial + ((size_t)(slot * 2 + wr) * 4) * N + n;
                pr[0] = s1;
                pr[(size_t)N] = s2;
                pr[(size_t)2 * N] = kshift[j];
                pr[(size_t)3 * N] = (float)run_rows;
            }
        }
    }
}

}  // namespace tp3d

using namespace tp3d;

namespace {
constexpr int GR_GRID = 1024;  // persistent: 4 workgroups per CU, a multiple of 8
struct RowsPlan {
    int tiles_n;
    int64_t row_blocks, items, blocks;
    bool per_workgroup;  // statistics rows: one per workgroup instead of one per 128-row block
    int64_t chunks;
};
RowsPlan rows_plan(int64_t M, int N)
{
    RowsPlan p;
    p.tiles_n = (N + GR_BN - 1) / GR_BN;
    p.row_blocks = (M + GR_BM - 1) / GR_BM;
    const int64_t groups = (p.row_blocks + 7) / 8;
    p.items = groups * 8 * p.tiles_n;  // items past the last row block stage zeros and store nothing
    p.blocks = p.items < GR_GRID ? p.items : GR_GRID;
    p.per_workgroup = p.blocks == GR_GRID && GR_GRID % (8 * p.tiles_n) == 0;
    p.chunks = 2 * (p.per_workgroup ? GR_GRID / p.tiles_n : p.row_blocks);  // one per wave row of a workgroup tile
    return p;
}
}  // namespace

// size of the statistics buffer: one row per 128-row block, or per persistent workgroup -- whichever is more (the item
// count is rounded up to groups of 8 row blocks, so 1017..1023 row blocks already fill the 1024-workgroup grid)
TP3D_EXPORT size_t tp3d_gemm_rows_stat_floats(int64_t M, int N)
{
    if (M <= 0 || N <= 0) return 0;
    const RowsPlan p = rows_plan(M, N);
    const int64_t rows = p.chunks > 2 * p.row_blocks ? p.chunks : 2 * p.row_blocks;
    return (size_t)rows * 4 * (size_t)N;  // per chunk: sum d, sum d^2, shift, rows
}

// number of statistics rows tp3d_gemm_rows_f32 writes for this shape = `chunks` of tp3d_bn_finalize_f32
TP3D_EXPORT int tp3d_gemm_rows_stat_chunks(int64_t M, int N)
{
    if (M <= 0 || N <= 0) return 0;
    return (int)rows_plan(M, N).chunks;
}

// plan[0..5] = column tiles, row blocks, work items, workgroups launched, statistics rows written (= chunks),
// 1 when they are one per workgroup (else one per 128-row block)
TP3D_EXPORT int tp3d_gemm_rows_plan(int64_t M, int N, int64_t *plan)
{
    if (M <= 0 || N <= 0 || !plan) return TP3D_E_BADARG;
    const RowsPlan p = rows_plan(M, N);
    plan[0] = p.tiles_n;
    plan[1] = p.row_blocks;
    plan[2] = p.items;
    plan[3] = p.blocks;
    plan[4] = p.chunks;
    plan[5] = p.per_workgroup ? 1 : 0;
    return TP3D_OK;
}

TP3D_EXPORT int tp3d_gemm_rows_f32(const float *A, const float *Bt, int64_t M, int N, int K, float *C,
                                   float *stat_partial, void *stream)
{
    if (M < 0 || N <= 0 || K <= 0 || (K & 3)) return TP3D_E_BADARG;  // operand rows must be 16-byte aligned
    if (M == 0) return TP3D_OK;
    if (!A || !Bt || !C) return TP3D_E_BADARG;
    const RowsPlan p = rows_plan(M, N);
    const int tiles_n = p.tiles_n;
    const int64_t items = p.items, blocks = p.blocks;
    hipStream_t s = (hipStream_t)stream;
    if (stat_partial && p.per_workgroup)
        hipLaunchKernelGGL(gemm_rows_kernel<2>, dim3((unsigned)blocks), dim3(GR_BLOCK_T), 0, s, A, Bt, M, N, K, tiles_n,
                           items, C, stat_partial);
    else if (stat_partial)
        hipLaunchKernelGGL(gemm_rows_kernel<1>, dim3((unsigned)blocks), dim3(GR_BLOCK_T), 0, s, A, Bt, M, N, K, tiles_n,
                           items, C, stat_partial);
    else
        hipLaunchKernelGGL(gemm_rows_kernel<0>, dim3((unsigned)blocks), dim3(GR_BLOCK_T), 0, s, A, Bt, M, N, K, tiles_n,
                           items, C, stat_partial);
    return check_launch();
}
