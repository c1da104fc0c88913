// Weight-gradient contraction of the shared MLPs:  dW[n,k] = sum_r dY[r,n] * A[r,k]
// dY (M, N) and A (M, K) row-major with M = B*npoint*nsample up to ~1e6 rows and N, K <= ~1500: a GEMM whose
// OUTPUT is tiny and whose contraction dimension is huge.  Library GEMMs pick 32x32 macro-tiles without a
// split over M for it (measured 0.5-1.5 ms per layer on MI355X); here the rows are split over the grid, every
// workgroup streams its row range once through LDS and accumulates a (TN x TK) tile with fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD), and a second pass sums the splits in fixed order
// (no atomics: bitwise reproducible).
//
// (The large layers run on the bf16 matrix pipe -- gemm_tn_x3.hip, which also forms the activated operand in its loader
// waves; the first layer of grouped rows on gemm_tn_narrow.hip.  A form of THIS kernel that built both operands while
// staging them, in the waves that also issue the MFMAs, was the first design and measured slower: DESIGN.md section 5.)
//
// Reference semantics: the weight gradient of Conv2d 1x1 (bias=False) inside MLP2D
// (torch_points3d/core/common_modules/dense_modules.py:5-12,25-29) -- computed by autograd in the reference.
#include <type_traits>

#include "tp3d_common.h"

namespace tp3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TN_BLOCK = 256;  // 4 waves as 2 x 2
constexpr int TN_BR_MAX = 64;  // most rows staged per step (narrow tiles stage more rows per barrier)

// Each wave owns WM x WN MFMA tiles of 32x32; the four waves form a GN x (4/GN) grid, so the workgroup tile is
// (GN*WM*32) x ((4/GN)*WN*32): 2 x 2 waves normally, 1 x 4 for a handful of output rows (the class-score layer: N = 10
// would fill 10 of 64 tile rows otherwise, and the contraction is then bound by wasted MFMA work, not by HBM).
template <int WM, int WN, bool VECY, bool VECA, int GN = 2>
__global__ __launch_bounds__(TN_BLOCK) void gemm_tn_partial_kernel(const float *__restrict__ dY,
                                                                    const float *__restrict__ A, int64_t M, int N,
                                                                    int K, int64_t rows_per_split, int tiles_k,
                                                                    float *__restrict__ partial /*[S][N][K]*/)
{
    constexpr int GK = 4 / GN;
    constexpr int TN = GN * WM * 32, TK = GK * WN * 32;
    // ~16-32 KiB staged per step whatever the tile shape (three-tile-wide shapes: 16 rows, so that every thread
    // owns whole float4 slots of both operands)
    constexpr int TN_BR = (WM * WN) % 3 == 0 ? 16 : 64 / (WM * WN);
    constexpr int LDN = TN + 4, LDK = TK + 4;  // +4 floats: keeps float4 stores aligned, spreads rows over banks
    __shared__ __attribute__((aligned(16))) float sY[TN_BR * LDN];
    __shared__ __attribute__((aligned(16))) float sA[TN_BR * LDK];

    const int tile = blockIdx.x;
    const int n0 = (tile / tiles_k) * TN, k0 = (tile % tiles_k) * TK;
    const int split = blockIdx.y;
    const int64_t r_begin = (int64_t)split * rows_per_split;
    const int64_t r_end = min(r_begin + rows_per_split, M);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / GK, wc = wave % GK;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // VECY / VECA: that operand's rows are 16-byte aligned (ld % 4 == 0) -> float4 loads, else 4 scalar loads

    // four consecutive floats of row r starting at column c (zero past the row range / matrix edge)
    auto load4 = [&](const float *base, int64_t r, int c, int ld, auto aligned_tag) __attribute__((always_inline)) -> float4 {
        constexpr bool aligned = decltype(aligned_tag)::value;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < r_end) {
            const float *src = base + r * ld + c;
            if (aligned) {
                if (c < ld) v = *reinterpret_cast<const float4 *>(src);  // ld % 4 == 0: all four inside
            } else {
                if (c + 0 < ld) v.x = src[0];
                if (c + 1 < ld) v.y = src[1];
                if (c + 2 < ld) v.z = src[2];
                if (c + 3 < ld) v.w = src[3];
            }
        }
        return v;
    };

    // ---- software pipeline: the loads of step i+1 are in flight while step i runs on the MFMA pipe
    constexpr int PY = TN_BR * (TN / 4) / TN_BLOCK, PA = TN_BR * (TK / 4) / TN_BLOCK;  // float4 slots per thread
    static_assert(PY >= 1 && PA >= 1 && PY * TN_BLOCK == TN_BR * (TN / 4) && PA * TN_BLOCK == TN_BR * (TK / 4),
                  "staged tile must be a whole number of float4 slots per thread");
    float4 ry[PY], ra[PA];
    auto fetch = [&](int64_t r0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            const int e = tid + i * TN_BLOCK;
            const int64_t r = r0 + e / (TN / 4);
            const int c = n0 + (e % (TN / 4)) * 4;
            ry[i] = load4(dY, r, c, N, std::integral_constant<bool, VECY>());
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int e = tid + i * TN_BLOCK;
            ra[i] = load4(A, r0 + e / (TK / 4), k0 + (e % (TK / 4)) * 4, K, std::integral_constant<bool, VECA>());
        }
    };
    if (r_begin < r_end) fetch(r_begin);
    for (int64_t r0 = r_begin; r0 < r_end; r0 += TN_BR) {
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            const int e = tid + i * TN_BLOCK;
            *reinterpret_cast<float4 *>(&sY[(e / (TN / 4)) * LDN + (e % (TN / 4)) * 4]) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int e = tid + i * TN_BLOCK;
            *reinterpret_cast<float4 *>(&sA[(e / (TK / 4)) * LDK + (e % (TK / 4)) * 4]) = ra[i];
        }
        __syncthreads();
        if (r0 + TN_BR < r_end) fetch(r0 + TN_BR);
#pragma unroll
        for (int rr = 0; rr < TN_BR; rr += 2) {
            float a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = sY[(rr + lh) * LDN + (wr * WM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = sA[(rr + lh) * LDK + (wc * WN + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- write the split's partial tile: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float *out = partial + (size_t)split * N * K;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + (wr * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int k = k0 + (wc * WN + j) * 32 + l31;
                if (n < N && k < K) out[(size_t)n * K + k] = acc[i][j][e];
            }
}

// out[e] = sum_s partial[s][e]: 64 outputs x 16 split lanes per workgroup; every lane adds its splits in
// ascending order and the 16 lane sums are combined in lane order, so the result does not depend on timing.
constexpr int RD_E = 64, RD_S = 16;
__global__ __launch_bounds__(RD_E *RD_S) void gemm_tn_reduce_kernel(const float *__restrict__ partial, int S,
                                                                     int64_t NK, float *__restrict__ out)
{
    __shared__ float sm[RD_S][RD_E];
    const int tx = threadIdx.x % RD_E, ty = threadIdx.x / RD_E;
    const int64_t e = (int64_t)blockIdx.x * RD_E + tx;
    float acc = 0.0f;
    if (e < NK)
        for (int s = ty; s < S; s += RD_S) acc += partial[(size_t)s * NK + e];
    sm[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && e < NK) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < RD_S; ++k) t += sm[k][tx];
        out[e] = t;
    }
}

// the split sum as a launch other translation units can enqueue (gemm_tn_x3.hip)
int tn_reduce_splits(const float *partial, int splits, int64_t NK, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((NK + RD_E - 1) / RD_E)), dim3(RD_E * RD_S), 0, s, partial, splits,
                       NK, out);
    return check_launch();
}

struct TnPlan {
    int wm, wn, gn, tn, tk, tiles_n, tiles_k, splits;
    int64_t rows_per_split;
};

static TnPlan plan_tn(int64_t M, int N, int K)
{
    TnPlan p;
    p.wm = N > 64 ? 2 : 1;
    // K just above a multiple of 128 (131 = 128 features + xyz): 192-wide tiles halve the padded columns and read dY
    // once instead of twice; wider K: 192 only where it strictly removes padded columns
    if (K <= 64) p.wn = 1;
    else if (K <= 128) p.wn = 2;
    else if (K <= 192) p.wn = 3;
    else p.wn = ((K + 191) / 192) * 192 < ((K + 127) / 128) * 128 ? 3 : 2;
    p.gn = 2;
    if (N <= 32 && K >= 128) {  // a handful of output rows: one wave row, four wave columns (32 x 128 tiles)
        p.gn = 1;
        p.wm = 1;
        p.wn = 1;
    }
    p.tn = 32 * p.gn * p.wm;
    p.tk = 32 * (4 / p.gn) * p.wn;
    p.tiles_n = (N + p.tn - 1) / p.tn;
    p.tiles_k = (K + p.tk - 1) / p.tk;
    const int tiles = p.tiles_n * p.tiles_k;
    // ~4 workgroups per CU (256 CUs) keep the MFMA pipes fed; at least 256 rows per split, at most 512 splits
    // (every split writes an N x K partial tile that the reduction pass reads back)
    int64_t want = (1024 + tiles - 1) / tiles;
    const int min_rows = 256;
    int64_t max_by_rows = (M + min_rows - 1) / min_rows;
    int64_t s = want < max_by_rows ? want : max_by_rows;
    if (s < 1) s = 1;
    if (s > 512) s = 512;
    p.rows_per_split = ((M + s - 1) / s + TN_BR_MAX - 1) / TN_BR_MAX * TN_BR_MAX;
    p.splits = (int)((M + p.rows_per_split - 1) / p.rows_per_split);
    return p;
}

static int launch_tn(const float *dY, const float *A, int64_t M, int N, int K, float *out, float *workspace, hipStream_t s)
{
    const TnPlan p = plan_tn(M, N, K);
    if (p.splits > 65535) return TP3D_E_TOOBIG;
    dim3 grid(p.tiles_n * p.tiles_k, p.splits);
    const bool vy = (N & 3) == 0, va = (K & 3) == 0;
#define TP3D_TN_LAUNCH(WM_, WN_, VY_, VA_)                                                                          \
    hipLaunchKernelGGL((gemm_tn_partial_kernel<WM_, WN_, VY_, VA_>), grid, dim3(TN_BLOCK), 0, s, dY, A, M, N, K,        \
                       p.rows_per_split, p.tiles_k, workspace)
#define TP3D_TN_ALIGN(WM_, WN_)                                                                                     \
    do {                                                                                                            \
        if (vy && va) TP3D_TN_LAUNCH(WM_, WN_, true, true);                                                         \
        else if (vy) TP3D_TN_LAUNCH(WM_, WN_, true, false);                                                         \
        else if (va) TP3D_TN_LAUNCH(WM_, WN_, false, true);                                                         \
        else TP3D_TN_LAUNCH(WM_, WN_, false, false);                                                                \
    } while (0)
    if (p.gn == 1) {
        if (vy && va) hipLaunchKernelGGL((gemm_tn_partial_kernel<1, 1, true, true, 1>), grid, dim3(TN_BLOCK), 0, s, dY, A, M, N, K,
                                         p.rows_per_split, p.tiles_k, workspace);
        else if (va) hipLaunchKernelGGL((gemm_tn_partial_kernel<1, 1, false, true, 1>), grid, dim3(TN_BLOCK), 0, s, dY, A, M, N, K,
                                        p.rows_per_split, p.tiles_k, workspace);
        else if (vy) hipLaunchKernelGGL((gemm_tn_partial_kernel<1, 1, true, false, 1>), grid, dim3(TN_BLOCK), 0, s, dY, A, M, N, K,
                                        p.rows_per_split, p.tiles_k, workspace);
        else hipLaunchKernelGGL((gemm_tn_partial_kernel<1, 1, false, false, 1>), grid, dim3(TN_BLOCK), 0, s, dY, A, M, N, K,
                                p.rows_per_split, p.tiles_k, workspace);
    } else if (p.wm == 2 && p.wn == 3) TP3D_TN_ALIGN(2, 3);
    else if (p.wn == 3) TP3D_TN_ALIGN(1, 3);
    else if (p.wm == 2 && p.wn == 2) TP3D_TN_ALIGN(2, 2);
    else if (p.wm == 2) TP3D_TN_ALIGN(2, 1);
    else if (p.wn == 2) TP3D_TN_ALIGN(1, 2);
    else TP3D_TN_ALIGN(1, 1);
#undef TP3D_TN_ALIGN
#undef TP3D_TN_LAUNCH
    if (int rc = check_launch()) return rc;
    return tn_reduce_splits(workspace, p.splits, (int64_t)N * K, out, s);
}

}  // namespace tp3d

using namespace tp3d;

TP3D_EXPORT size_t tp3d_gemm_tn_workspace_floats(int64_t M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const TnPlan p = plan_tn(M, N, K);
    return (size_t)p.splits * (size_t)N * (size_t)K;
}

// plan[0..7] = splits, rows per split, tile rows (N side), tile columns (K side), tiles, rows staged per step,
// workspace floats the partial kernel writes, last row a split starts at
TP3D_EXPORT int tp3d_gemm_tn_plan(int64_t M, int N, int K, int64_t *plan)
{
    if (M <= 0 || N <= 0 || K <= 0 || !plan) return TP3D_E_BADARG;
    const TnPlan p = plan_tn(M, N, K);
    plan[0] = p.splits;
    plan[1] = p.rows_per_split;
    plan[2] = p.tn;
    plan[3] = p.tk;
    plan[4] = (int64_t)p.tiles_n * p.tiles_k;
    plan[5] = (p.wm * p.wn) % 3 == 0 ? 16 : 64 / (p.wm * p.wn);
    plan[6] = (int64_t)p.splits * N * K;
    plan[7] = (int64_t)(p.splits - 1) * p.rows_per_split;
    return TP3D_OK;
}

TP3D_EXPORT int tp3d_gemm_tn_f32(const float *dY, const float *A, int64_t M, int N, int K, float *out,
                                 float *workspace, void *stream)
{
    if (M < 0 || N <= 0 || K <= 0 || !out) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) return zero_async(out, (size_t)N * K * sizeof(float), s);
    if (!dY || !A || !workspace) return TP3D_E_BADARG;
    return launch_tn(dY, A, M, N, K, out, workspace, s);
}
