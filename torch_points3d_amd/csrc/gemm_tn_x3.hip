// Weight-gradient contraction  dW[n,k] = sum_r dY[r,n] * A[r,k]  with the fp32 products carried by the bf16 matrix pipe.
//
// Every fp32 operand value x is written as x = hi + mid + lo, three bf16 terms obtained by truncation
// (hi = x with the low 16 bits cleared, mid = (x - hi) likewise, lo = x - hi - mid): 3 x 8 significand bits = the 24 of an
// fp32 value, so the decomposition is EXACT, and so is each of the nine term products (8 x 8 bits) inside the MFMA.
// dY[r,n] * A[r,k] = sum of the nine term pairs, each a v_mfma_f32_32x32x16_bf16 with fp32 accumulation: the same
// arithmetic as the fp32 MFMA kernel (exact products, fp32 sums) at 9 x 32 cycles per 32x32x16 block instead of
// 8 x 64 -- the contraction leaves the matrix pipe's critical path and the kernel runs at the rate HBM delivers the rows
// (gemm_tn.hip: 100 TFLOP/s = 0.64 of the fp32 MFMA peak on the 128 x 128 layers, MFMA-bound).
//
// Split roles (as gemm_rows_sp.hip): 8-wave workgroups, one per CU.  Waves 4-7 stream BR rows of both operands per step
// from HBM (two steps ahead in registers), split every value and write three bf16 planes per operand into LDS in the
// order the rows arrive ([row][column]); waves 0-3 read their MFMA fragments from those planes with the transposing LDS
// read (ds_read_b64_tr_b16: the contraction index is the ROW of both operands, i.e. the slow index of both LDS images)
// and issue the MFMAs; one barrier per step, two LDS buffers.  Plane rows are padded to a pitch of 32 or 96 (mod 128)
// halfwords, which makes the four rows of a transposed read fall into disjoint bank groups.
// K just above a tile (131 = 128 features + xyz, padded to 132): a 32-column strip beside the 128 x 128 tile, one extra
// 32 x 32 block per wave, instead of a second tile column that would read dY twice.
//
// Partial tiles per row split, summed in fixed order by gemm_tn_reduce_kernel (gemm_tn.hip): bitwise reproducible.
// Reference semantics: weight gradient of Conv2d 1x1 (bias=False) inside MLP2D
// (torch_points3d/core/common_modules/dense_modules.py:5-12,25-29) -- autograd's `grad_output^T @ input` in the reference.
#include "tp3d_common.h"
#include "x3_split.h"

namespace tp3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int X3_BLOCK = 512;

__host__ __device__ constexpr int x3_pitch(int cols) { return cols % 64 == 0 ? cols + 32 : cols; }

// MFMA operand fragment of the 32x32x16 block whose 16 contraction rows start at `row0` and whose 32 output rows /
// columns are the plane columns col0 .. col0+31:  lane l (r = l & 31, h = l >> 5) gets plane[row0 + 8h + j][col0 + r],
// j = 0..7.  ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies the address of row q, columns 4p..4p+3 of a
// 4 x 16 block and receives column (lane & 15) of its four rows.
__device__ __forceinline__ bf16x8 x3_frag(const unsigned short *plane, int pitch, int row0, int col0, int lane)
{
    const int grp = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const unsigned short *a = plane + (row0 + 8 * (grp >> 1) + q) * pitch + col0 + 16 * (grp & 1) + 4 * p;
    typedef s16x4 __attribute__((address_space(3))) *lds_ptr;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * pitch));
    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
}

// WM, WN: 32x32 blocks per MFMA wave along n / k (waves 0-3 form a 2 x 2 grid: tile = 64 WM x 64 WN); STRIP: 32 more k
// columns beside the tile (WM == 2: one strip block per wave); BR: rows staged per step; TERMS: 9 = all nine term pairs
// (exact products), 6 = the six of weight >= 2^-16, 1 = the hi * hi pair alone (a diagnostic: loaders unchanged, a ninth of the MFMAs).
// A-operand prologue: A = LeakyReLU((Yp - mean_k) * scale_k + beta_k), the activated input of the layer formed from the
// previous layer's pre-BatchNorm output Yp (M, K) and its statistics rows (null mean = plain rows).  The loader waves
// apply it on their way to the split, in the forward kernel's operation order (the rows they form are bit for bit the
// activated rows the forward pass would have written), so the forward pass need not write that (M, K) side output at all.
struct X3Prologue {
    const float *mean, *scale, *beta;
    float slope;
    int reverse;  // walk the row blocks last to first (start on the rows the kernel before this one touched last)
    // RED: the BatchNorm-backward REDUCTIONS of the layer whose pre-BatchNorm output Yp this kernel's loader waves already
    // stream -- sum dZ and sum dZ * yhat per column k, dZ = dA * act'(z) -- from one more operand stream, dA (M, K), the
    // gradient of that layer's activated output: the separate pass read Yp again.  One chunk [2][K] per workgroup.
    const float *dA, *invstd;
    float *red_partial;
};

template <int WM, int WN, bool STRIP, int BR, int TERMS, bool RED = false>
__global__ __launch_bounds__(X3_BLOCK) void gemm_tn_x3_kernel(const float *__restrict__ dY, const float *__restrict__ A,
                                                              int64_t M, int N, int K, int64_t rows_per_split,
                                                              int tiles_k, float *__restrict__ partial /*[S][N][K]*/,
                                                              X3Prologue pro)
{
    static_assert(!STRIP || WM == 2, "the strip is one block per wave of a 128-row tile");
    static_assert(!(RED && STRIP), "the reductions ride on the plain tiles (one tile column: every workgroup sees all K columns)");
    constexpr int TN = 64 * WM, TK = 64 * WN, TKS = TK + (STRIP ? 32 : 0);
    constexpr int PN = x3_pitch(TN), PK = x3_pitch(TKS);
    constexpr int PLANE_N = BR * PN, PLANE_K = BR * PK;   // halfwords
    constexpr int BUF = 3 * PLANE_N + 3 * PLANE_K;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    __shared__ __attribute__((aligned(16))) float sK[RED ? 4 : 3][TKS];  // mean, scale, beta (, invstd) of the tile's k columns (zero past K)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int n0 = (tile / tiles_k) * TN, k0 = (tile % tiles_k) * TK;
    const bool prologue = pro.mean != nullptr;
    if (prologue) {
        for (int c = tid; c < TKS; c += X3_BLOCK) {
            const bool in = k0 + c < K;
            sK[0][c] = in ? pro.mean[k0 + c] : 0.0f;
            sK[1][c] = in ? pro.scale[k0 + c] : 0.0f;
            sK[2][c] = in ? pro.beta[k0 + c] : 0.0f;
            if (RED) sK[3][c] = in ? pro.invstd[k0 + c] : 0.0f;
        }
        __syncthreads();  // (before the roles part: every wave passes here once)
    }
    // Row blocks of BR rows are dealt round-robin to the splits: at any moment the workgroups of a launch read ONE contiguous
    // stretch of both operands (splits x BR rows), which spreads over every HBM channel; contiguous row ranges per split would
    // be 2^k bytes apart (e.g. 2048 rows x 512 B = 1 MiB) and walk the channels in lockstep.
    const int64_t r_end = M;
    const int64_t blocks = (M + BR - 1) / BR;
    const int S = gridDim.y;
    const int steps = (int)(((int64_t)blocks - blockIdx.y + S - 1) / S);
    (void)rows_per_split;

    if (__builtin_amdgcn_readfirstlane(tid) >= 256) {  // (provably wave-uniform: a scalar branch, s_setprio only on this side)
        // ------------------------------------------------------------------------------------------ loader waves
        // their instructions first: the MFMA waves have 24 of every 32 cycles to spare, the rows must keep coming
        __builtin_amdgcn_s_setprio(3);
        const int lt = tid - 256;
        constexpr int SY = BR * (TN / 4) / 256, SA = BR * (TK / 4) / 256, SS = STRIP ? BR * 8 / 256 : 0;
        static_assert(SY >= 1 && SA >= 1 && SY * 256 == BR * (TN / 4) && SA * 256 == BR * (TK / 4), "whole float4 slots per thread");
        static_assert(!STRIP || SS * 256 == BR * 8, "whole strip slots per thread");
        float4 ry0[SY], ra0[SA], rs0[STRIP ? SS : 1], ry1[SY], ra1[SA], rs1[STRIP ? SS : 1], ry2[SY], ra2[SA], rs2[STRIP ? SS : 1];
        float4 rd0[RED ? SA : 1], rd1[RED ? SA : 1], rd2[RED ? SA : 1];  // RED: the dA rows under the A rows
        float red1[4] = {0.f, 0.f, 0.f, 0.f}, red2[4] = {0.f, 0.f, 0.f, 0.f};  // this thread's four columns (the same in every slot)
        // the row blocks of a step are reduced by ONE of the tile rows' workgroups (they all stream the same A rows)
        const int red_tiles = (int)gridDim.x;  // (one tile column)
        auto mine = [&](int t) __attribute__((always_inline)) -> bool { return RED && (t % red_tiles) == tile; };
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        // every slot keeps its place inside a row block for the whole kernel: byte offset inside the block as a 32-bit lane
        // value, the block's address as a wave-uniform pointer (x3_serves bounds M * max(N, K) * 4 below 2^32)
        const bool cols_full = n0 + TN <= N && k0 + TK <= K;  // (the strip's columns are checked per lane)
        unsigned oy[SY], oa[SA], os[STRIP ? SS : 1];
#pragma unroll
        for (int i = 0; i < SY; ++i) oy[i] = (unsigned)((((lt + i * 256) / (TN / 4)) * N + n0 + ((lt + i * 256) % (TN / 4)) * 4) * 4);
#pragma unroll
        for (int i = 0; i < SA; ++i) oa[i] = (unsigned)((((lt + i * 256) / (TK / 4)) * K + k0 + ((lt + i * 256) % (TK / 4)) * 4) * 4);
        if (STRIP) {
#pragma unroll
            for (int i = 0; i < SS; ++i) os[i] = (unsigned)((((lt + i * 256) / 8) * K + k0 + TK + ((lt + i * 256) % 8) * 4) * 4);
        }
        auto fetch = [&](int t, float4 *ry, float4 *ra, float4 *rs, float4 *rd) __attribute__((always_inline)) {
            const int64_t r0 = ((int64_t)(pro.reverse ? steps - 1 - t : t) * S + blockIdx.y) * BR;  // (last row blocks first)
            const char *by_ = reinterpret_cast<const char *>(dY) + r0 * N * 4;
            const char *ba_ = reinterpret_cast<const char *>(A) + r0 * K * 4;
            if (r0 + BR <= M && cols_full) {  // (wave-uniform) the whole block lies inside both matrices
#pragma unroll
                for (int i = 0; i < SY; ++i) ry[i] = *reinterpret_cast<const float4 *>(by_ + oy[i]);
#pragma unroll
                for (int i = 0; i < SA; ++i) ra[i] = *reinterpret_cast<const float4 *>(ba_ + oa[i]);
                if (mine(t)) {
                    const char *bd_ = reinterpret_cast<const char *>(pro.dA) + r0 * K * 4;
#pragma unroll
                    for (int i = 0; i < SA; ++i) rd[i] = *reinterpret_cast<const float4 *>(bd_ + oa[i]);
                }
                if (STRIP) {
#pragma unroll
                    for (int i = 0; i < SS; ++i) {
                        rs[i] = zero;
                        if (k0 + TK + ((lt + i * 256) % 8) * 4 < K) rs[i] = *reinterpret_cast<const float4 *>(ba_ + os[i]);
                    }
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < SY; ++i) {
                const int e = lt + i * 256;
                const int64_t r = r0 + e / (TN / 4);
                const int c = n0 + (e % (TN / 4)) * 4;
                ry[i] = zero;
                if (r < r_end && c < N) ry[i] = *reinterpret_cast<const float4 *>(dY + r * N + c);
            }
#pragma unroll
            for (int i = 0; i < SA; ++i) {
                const int e = lt + i * 256;
                const int64_t r = r0 + e / (TK / 4);
                const int c = k0 + (e % (TK / 4)) * 4;
                ra[i] = zero;
                if (r < r_end && c < K) ra[i] = *reinterpret_cast<const float4 *>(A + r * K + c);
                if (mine(t)) {
                    rd[i] = zero;  // (a zero gradient adds nothing: rows past M, columns past K)
                    if (r < r_end && c < K) rd[i] = *reinterpret_cast<const float4 *>(pro.dA + r * K + c);
                }
            }
            if (STRIP) {
#pragma unroll
                for (int i = 0; i < SS; ++i) {
                    const int e = lt + i * 256;
                    const int64_t r = r0 + e / 8;
                    const int c = k0 + TK + (e % 8) * 4;
                    rs[i] = zero;
                    if (r < r_end && c < K) rs[i] = *reinterpret_cast<const float4 *>(A + r * K + c);
                }
            }
        };
        auto put = [&](unsigned short *plane0, int plane_len, int pitch, int row, int col, float4 v) __attribute__((always_inline)) {
            uint2 h, m, l;
            x3_split(v, h, m, l);
            unsigned short *dst = plane0 + row * pitch + col;
            *reinterpret_cast<uint2 *>(dst) = h;
            *reinterpret_cast<uint2 *>(dst + plane_len) = m;
            *reinterpret_cast<uint2 *>(dst + 2 * plane_len) = l;
        };
        // (y - mean) * scale + beta, LeakyReLU -- the expression and order of gemm_rows_sp.hip's forward prologue; a column
        // past K has mean = scale = beta = 0 and stays zero
        auto act4 = [&](float4 v, int col) __attribute__((always_inline)) -> float4 {
            const float4 mu = *reinterpret_cast<const float4 *>(&sK[0][col]);
            const float4 sc = *reinterpret_cast<const float4 *>(&sK[1][col]);
            const float4 be = *reinterpret_cast<const float4 *>(&sK[2][col]);
            const float z0 = (v.x - mu.x) * sc.x + be.x, z1 = (v.y - mu.y) * sc.y + be.y;
            const float z2 = (v.z - mu.z) * sc.z + be.z, z3 = (v.w - mu.w) * sc.w + be.w;
            return make_float4(z0 > 0.0f ? z0 : z0 * pro.slope, z1 > 0.0f ? z1 : z1 * pro.slope, z2 > 0.0f ? z2 : z2 * pro.slope,
                               z3 > 0.0f ? z3 : z3 * pro.slope);
        };
        auto store = [&](int t, const float4 *ry, const float4 *ra, const float4 *rs, const float4 *rd) __attribute__((always_inline)) {
            unsigned short *by = smem + (t & 1) * BUF, *ba = by + 3 * PLANE_N;
            if (mine(t)) {
                // dZ = dA * act'(z), sums of dZ and dZ * yhat: the expressions of colreduce_partial_kernel<., 1> (rows.hip)
                const int col = ((lt % (TK / 4))) * 4;  // (256 % (TK / 4) == 0: every slot of this thread has these columns)
                const float4 mu = *reinterpret_cast<const float4 *>(&sK[0][col]), sc = *reinterpret_cast<const float4 *>(&sK[1][col]);
                const float4 be = *reinterpret_cast<const float4 *>(&sK[2][col]), is = *reinterpret_cast<const float4 *>(&sK[3][col]);
#pragma unroll
                for (int i = 0; i < SA; ++i) {
                    const float y[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w}, d[4] = {rd[i].x, rd[i].y, rd[i].z, rd[i].w};
                    const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, s4[4] = {sc.x, sc.y, sc.z, sc.w};
                    const float b4[4] = {be.x, be.y, be.z, be.w}, i4[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float yc = y[c] - m4[c];
                        const float dz = d[c] * (yc * s4[c] + b4[c] > 0.0f ? 1.0f : pro.slope);
                        red1[c] += dz;
                        red2[c] += dz * (yc * i4[c]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < SY; ++i) {
                const int e = lt + i * 256;
                put(by, PLANE_N, PN, e / (TN / 4), (e % (TN / 4)) * 4, ry[i]);
            }
#pragma unroll
            for (int i = 0; i < SA; ++i) {
                const int e = lt + i * 256;
                const int col = (e % (TK / 4)) * 4;
                put(ba, PLANE_K, PK, e / (TK / 4), col, prologue ? act4(ra[i], col) : ra[i]);
            }
            if (STRIP) {
#pragma unroll
                for (int i = 0; i < SS; ++i) {
                    const int e = lt + i * 256;
                    const int col = TK + (e % 8) * 4;
                    put(ba, PLANE_K, PK, e / 8, col, prologue ? act4(rs[i], col) : rs[i]);
                }
            }
        };
        // three register stages: while stage t is split and written, stages t+1 and t+2 are in flight -- the barrier couples the
        // loaders to the MFMA waves step by step, and with two stages a late barrier delayed the next fetch (4.1 TB/s with
        // the MFMAs running against 5.3 TB/s without them)
        if (steps > 0) fetch(0, ry0, ra0, rs0, rd0);
        if (steps > 1) fetch(1, ry1, ra1, rs1, rd1);
        if (steps > 2) fetch(2, ry2, ra2, rs2, rd2);
        auto iter = [&](int t, float4 *ry, float4 *ra, float4 *rs, float4 *rd) __attribute__((always_inline)) {
            store(t, ry, ra, rs, rd);
            if (t + 3 < steps) fetch(t + 3, ry, ra, rs, rd);
            __syncthreads();
        };
        for (int t = 0; t < steps; t += 3) {
            iter(t, ry0, ra0, rs0, rd0);
            if (t + 1 < steps) iter(t + 1, ry1, ra1, rs1, rd1);
            if (t + 2 < steps) iter(t + 2, ry2, ra2, rs2, rd2);
        }
        if (RED) {
            // the threads that share a column group (lt % (TK/4)) fold their sums through the LDS buffer the MFMA waves are
            // NOT reading in their last step, in thread order (fixed: reproducible); one chunk [2][K] per workgroup
            float *scr = reinterpret_cast<float *>(smem + (steps & 1) * BUF);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                scr[lt * 8 + c] = red1[c];
                scr[lt * 8 + 4 + c] = red2[c];
            }
            __syncthreads();  // (met by the MFMA waves after their loop)
            constexpr int CG = TK / 4, PER = 256 / CG;
            if (lt < CG) {
                float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
                for (int x = 0; x < PER; ++x)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        a1[c] += scr[(lt + x * CG) * 8 + c];
                        a2[c] += scr[(lt + x * CG) * 8 + 4 + c];
                    }
                float *pr = pro.red_partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * K;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k = k0 + lt * 4 + c;
                    if (k < K) {
                        pr[k] = a1[c];
                        pr[K + k] = a2[c];
                    }
                }
            }
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------- MFMA waves
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    // Two accumulators per block: `acc` takes the hi * hi pair alone, `low` every other pair (each at most 2^-7 of a hi * hi
    // product).  The sum of the small pairs never meets the large running sum before the end, so the large accumulator is
    // rounded once per 16 rows (the fp32 MFMA kernel: eight times) and the small one's roundings weigh 2^-7 as much.
    f32x16 acc[WM][WN], low[WM][WN], accs, lows;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = low[i][j][e] = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) accs[e] = lows[e] = 0.0f;

    for (int t = 0; t < steps; ++t) {
        __syncthreads();
        const unsigned short *by = smem + (t & 1) * BUF, *ba = by + 3 * PLANE_N;
#pragma unroll
        for (int s = 0; s < BR; s += 16) {
            bf16x8 a[WM][3], b[WN][3], bs[3], as[3];  // as: the rows of the wave's strip block (block `wc` of its own rows,
                                                      // read again rather than selected from a[] by a run-time index)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < WM; ++i) a[i][p] = x3_frag(by + p * PLANE_N, PN, s, (wr * WM + i) * 32, lane);
#pragma unroll
                for (int j = 0; j < WN; ++j) b[j][p] = x3_frag(ba + p * PLANE_K, PK, s, (wc * WN + j) * 32, lane);
                if (STRIP) {
                    bs[p] = x3_frag(ba + p * PLANE_K, PK, s, TK, lane);
                    as[p] = x3_frag(by + p * PLANE_N, PN, s, (wr * WM + wc) * 32, lane);
                }
            }
            // term pairs, smallest first: (lo,lo) (lo,mid) (mid,lo) | (lo,hi) (hi,lo) (mid,mid) | (mid,hi) (hi,mid) | (hi,hi)
            constexpr int PA_[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
            constexpr int PB_[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int u = 9 - TERMS; u < 8; ++u) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        low[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA_[u]], b[j][PB_[u]], low[i][j], 0, 0, 0);
                if (STRIP) lows = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[PA_[u]], bs[PB_[u]], lows, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            if (STRIP) accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[0], bs[0], accs, 0, 0, 0);
        }
    }
    if (RED) __syncthreads();  // the loader waves fold their column sums through the idle LDS buffer

    // D[row][col]: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float *out = partial + (size_t)blockIdx.y * N * K;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + (wr * WM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int k = k0 + (wc * WN + j) * 32 + l31;
                if (n < N && k < K) out[(size_t)n * K + k] = acc[i][j][e] + low[i][j][e];
            }
    if (STRIP) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = n0 + (wr * WM + wc) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;  // the wave's block `wc` of its own rows
            const int k = k0 + TK + l31;
            if (n < N && k < K) out[(size_t)n * K + k] = accs[e] + lows[e];
        }
    }
}

struct X3Plan {
    int wm, wn, strip, br, tiles_n, tiles_k, splits, lds_bytes;
    int64_t rows_per_split;
};

static X3Plan x3_plan(int64_t M, int N, int K);

static bool x3_serves(int64_t M, int N, int K)
{
    // (below ~10^5 rows one workgroup per CU has too few steps to hide its prologue -- 40000 x 128 x 160: 65 vs 42 us.  64 x 64
    //  outputs: the plain form only matches the fp32 kernel, which runs at the rate HBM delivers those rows, but the forms
    //  that build the activated operand and the reductions of the layer below save a side output and a pass)
    if (!(M >= 131072 && N >= 64 && K >= 64 && (N & 3) == 0 && (K & 3) == 0 &&
          M * (int64_t)(N > K ? N : K) < ((int64_t)1 << 30)))
        return false;
    // every tile column re-reads dY and every tile row re-reads A, and a padded tile splits and multiplies zeros: the
    // kernel pays off while the output fills at least 80 % of at most two tiles (measured on BASELINE config 3:
    // 524288 x 256 x 196 -- four tiles -- 743 vs 606 us, 2097152 x 128 x 96 -- 75 % of a tile -- 793 vs 588 us on the fp32
    // kernel; 262144 x 256 x 128 -- two full tiles -- 156 vs 193 us)
    const X3Plan p = x3_plan(M, N, K);
    const int tiles = p.tiles_n * p.tiles_k;
    const int64_t padded = (int64_t)p.tiles_n * 64 * p.wm * ((int64_t)p.tiles_k * 64 * p.wn + (p.strip ? 32 : 0));
    return tiles <= 2 && (int64_t)N * K * 5 >= padded * 4;
}

static X3Plan x3_plan(int64_t M, int N, int K)
{
    X3Plan p;
    p.wm = N <= 64 ? 1 : 2;
    const int rem128 = K % 128, rem64 = K % 64;
    p.strip = 0;
    if (K <= 64) p.wn = 1;
    else if (K > 128 && K <= 160 && p.wm == 2) p.wn = 2, p.strip = 1;  // 128 + (1..32): strip beside the tile
    else if (K <= 128) p.wn = 2;
    else p.wn = 2;
    (void)rem128;
    (void)rem64;
    p.br = (p.wm == 1 && p.wn == 1) ? 64 : 32;
    const int tn = 64 * p.wm, tk = 64 * p.wn;
    p.tiles_n = (N + tn - 1) / tn;
    p.tiles_k = p.strip ? 1 : (K + tk - 1) / tk;
    const int tiles = p.tiles_n * p.tiles_k;
    // one 8-wave workgroup per CU (256 CUs): as many row splits as that gives, at least 1024 rows each
    int64_t s = (256 + tiles - 1) / tiles;
    const int64_t max_by_rows = (M + 1023) / 1024;
    if (s > max_by_rows) s = max_by_rows;
    if (s < 1) s = 1;
    p.rows_per_split = ((M + s - 1) / s + p.br - 1) / p.br * p.br;
    p.splits = (int)((M + p.rows_per_split - 1) / p.rows_per_split);
    const int pn = x3_pitch(tn), pk = x3_pitch(tk + (p.strip ? 32 : 0));
    p.lds_bytes = 2 * 3 * p.br * (pn + pk) * 2;
    return p;
}

template <int WM, int WN, bool STRIP, int BR, int TERMS, bool RED = false>
static void x3_launch(const X3Plan &p, const float *dY, const float *A, int64_t M, int N, int K, float *ws, X3Prologue pro,
                      hipStream_t s)
{
    static bool allowed[64] = {};
    auto kern = gemm_tn_x3_kernel<WM, WN, STRIP, BR, TERMS, RED>;
    allow_large_dynamic_lds(reinterpret_cast<const void *>(kern), p.lds_bytes, allowed);
    hipLaunchKernelGGL(kern, dim3(p.tiles_n * p.tiles_k, p.splits), dim3(X3_BLOCK), p.lds_bytes, s, dY, A, M, N, K,
                       p.rows_per_split, p.tiles_k, ws, pro);
}

}  // namespace tp3d

using namespace tp3d;

TP3D_EXPORT int tp3d_gemm_tn_x3_serves(int64_t M, int N, int K) { return x3_serves(M, N, K) ? 1 : 0; }

TP3D_EXPORT size_t tp3d_gemm_tn_x3_workspace_floats(int64_t M, int N, int K)
{
    if (!x3_serves(M, N, K)) return 0;
    return (size_t)x3_plan(M, N, K).splits * (size_t)N * (size_t)K;
}

// plan[0..7] = splits, rows per split, tile rows (n), tile columns (k, strip included), tiles, rows staged per step,
// workspace floats, dynamic LDS bytes
TP3D_EXPORT int tp3d_gemm_tn_x3_plan(int64_t M, int N, int K, int64_t *plan)
{
    if (!plan || !x3_serves(M, N, K)) return TP3D_E_BADARG;
    const X3Plan p = x3_plan(M, N, K);
    plan[0] = p.splits;
    plan[1] = p.rows_per_split;
    plan[2] = 64 * p.wm;
    plan[3] = 64 * p.wn + (p.strip ? 32 : 0);
    plan[4] = (int64_t)p.tiles_n * p.tiles_k;
    plan[5] = p.br;
    plan[6] = (int64_t)p.splits * N * K;
    plan[7] = p.lds_bytes;
    return TP3D_OK;
}

static int x3_run(const float *dY, const float *A, int64_t M, int N, int K, int terms, float *out, float *workspace,
                  X3Prologue pro, void *stream)
{
    if (!x3_serves(M, N, K) || !dY || !A || !out || !workspace || (terms != 9 && terms != 6 && terms != 1)) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const X3Plan p = x3_plan(M, N, K);
    if (p.splits > 65535) return TP3D_E_TOOBIG;
#define TP3D_X3(WM_, WN_, ST_, BR_)                                                                 \
    do {                                                                                            \
        if (terms == 9) x3_launch<WM_, WN_, ST_, BR_, 9>(p, dY, A, M, N, K, workspace, pro, s);     \
        else if (terms == 6) x3_launch<WM_, WN_, ST_, BR_, 6>(p, dY, A, M, N, K, workspace, pro, s); \
        else x3_launch<WM_, WN_, ST_, BR_, 1>(p, dY, A, M, N, K, workspace, pro, s);                \
    } while (0)
    if (p.wm == 2 && p.wn == 2 && p.strip) TP3D_X3(2, 2, true, 32);
    else if (p.wm == 2 && p.wn == 2) TP3D_X3(2, 2, false, 32);
    else if (p.wm == 2) TP3D_X3(2, 1, false, 32);
    else if (p.wn == 2) TP3D_X3(1, 2, false, 32);
    else TP3D_X3(1, 1, false, 64);
#undef TP3D_X3
    if (int rc = check_launch()) return rc;
    return tn_reduce_splits(workspace, p.splits, (int64_t)N * K, out, s);
}

TP3D_EXPORT int tp3d_gemm_tn_x3_f32(const float *dY, const float *A, int64_t M, int N, int K, int terms, float *out,
                                    float *workspace, int reverse, void *stream)
{
    const X3Prologue none = {nullptr, nullptr, nullptr, 1.0f, reverse, nullptr, nullptr, nullptr};
    return x3_run(dY, A, M, N, K, terms, out, workspace, none, stream);
}

// dW = dY^T * LeakyReLU((Yp - mean_k) * scale_k + beta_k): the A operand formed by the loader waves from the previous
// layer's pre-BatchNorm output Yp (M, K) and its statistics rows (K floats each) -- see X3Prologue
TP3D_EXPORT int tp3d_gemm_tn_x3_act_f32(const float *dY, const float *Yp, const float *mean_k, const float *scale_k,
                                        const float *beta_k, float slope_k, int64_t M, int N, int K, int terms, float *out,
                                        float *workspace, int reverse, void *stream)
{
    if (!mean_k || !scale_k || !beta_k) return TP3D_E_BADARG;
    const X3Prologue pro = {mean_k, scale_k, beta_k, slope_k, reverse, nullptr, nullptr, nullptr};
    return x3_run(dY, Yp, M, N, K, terms, out, workspace, pro, stream);
}

// ... and the BatchNorm-backward reductions of the layer Yp belongs to, from its dA (M, K) (see X3Prologue): red_out (4, K)
// = dbeta, dgamma, c1 = dbeta / M, c2 = invstd * dgamma / M (zero with training == 0) -- what tp3d_bn_bwd_reduce_f32 would
// return for (dA_k, Yp), without its pass over both.  Served (tp3d_gemm_tn_x3_red_chunks > 0): the shapes of
// tp3d_gemm_tn_x3_serves with one tile column and no strip (K <= 128), terms == 6.  red_workspace: chunks * 2 * K floats.
static bool x3_red_serves(int64_t M, int N, int K)
{
    if (!x3_serves(M, N, K)) return false;
    const X3Plan p = x3_plan(M, N, K);
    return p.tiles_k == 1 && !p.strip && (p.wm == 2 || p.wn == 1);  // <2,2>, <2,1>, <1,1> tiles
}

TP3D_EXPORT int tp3d_gemm_tn_x3_red_chunks(int64_t M, int N, int K)
{
    if (!x3_red_serves(M, N, K)) return 0;
    const X3Plan p = x3_plan(M, N, K);
    return p.tiles_n * p.splits;
}

TP3D_EXPORT int tp3d_gemm_tn_x3_act_red_f32(const float *dY, const float *Yp, const float *mean_k, const float *scale_k,
                                            const float *beta_k, const float *invstd_k, float slope_k, const float *dA_k,
                                            int training, int64_t M, int N, int K, int terms, float *out, float *workspace,
                                            float *red_out, float *red_workspace, int reverse, void *stream)
{
    if (!mean_k || !scale_k || !beta_k || !invstd_k || !dA_k || !red_out || !red_workspace || terms != 6) return TP3D_E_BADARG;
    if (!x3_red_serves(M, N, K) || !dY || !Yp || !out || !workspace) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const X3Plan p = x3_plan(M, N, K);
    if (p.splits > 65535) return TP3D_E_TOOBIG;
    const X3Prologue pro = {mean_k, scale_k, beta_k, slope_k, reverse, dA_k, invstd_k, red_workspace};
    if (p.wm == 2 && p.wn == 2) x3_launch<2, 2, false, 32, 6, true>(p, dY, Yp, M, N, K, workspace, pro, s);
    else if (p.wm == 2) x3_launch<2, 1, false, 32, 6, true>(p, dY, Yp, M, N, K, workspace, pro, s);
    else x3_launch<1, 1, false, 64, 6, true>(p, dY, Yp, M, N, K, workspace, pro, s);
    if (int rc = check_launch()) return rc;
    if (int rc = tn_reduce_splits(workspace, p.splits, (int64_t)N * K, out, s)) return rc;
    return bn_bwd_finalize_launch(red_workspace, p.tiles_n * p.splits, K, red_out, red_out + K, invstd_k, M, training,
                                  red_out + 2 * (size_t)K, red_out + 3 * (size_t)K, s);
}
