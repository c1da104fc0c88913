// KPConv rigid kernel-point convolution, stage 1: kernel-point weighted neighbourhood features.
//
// Reference: torch_points3d/modules/KPConv/convolution_ops.py:19-107 (KPConv_ops) with the gather of
// core/common_modules/gathering.py:1-33 (index -1 = shadow neighbour: point at 1e6, zero feature).
//   wf[q, k, :] = sum_n h(| (s[nbr[q,n]] - q) - K_k |) * x[nbr[q,n], :]          (convolution_ops.py:49-98)
//   out[q, :]   = sum_k wf[q, k, :] @ W[k]     == (Nq, KP*Cin) @ (KP*Cin, Cout)    (:101-105, one library GEMM)
// The reference materialises (Nq, Mn, KP, 3) differences and (Nq, KP, Mn) weights in HBM; here one wave owns a
// query: the KP x Mn influence weights live in LDS, neighbour feature rows are read as coalesced row segments
// and KP accumulators per lane stay in registers, so HBM sees the neighbour rows once and wf once.
#include <algorithm>

#include "tp3d_common.h"

namespace tp3d {

__device__ __forceinline__ float rl_bcast(float x, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}

// Every wave owns its slice of the LDS arrays, so the phases of a query only need the wave's own LDS writes to have
// landed: a wave-level barrier, not a workgroup one (which would make four unrelated queries wait for each other).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

constexpr int KP_BLOCK = 256;  // 4 waves, one query per wave
constexpr int KP_MAX = 16;     // kernel points (15 in every reference config)
constexpr int KP_NCH = 48;     // neighbours per LDS pass (covers every max_num_neighbors of the reference configs)
constexpr int KP_GROUP = 16;   // neighbour rows fetched together in the forward accumulation
static_assert(KP_NCH % KP_GROUP == 0 && KP_NCH <= 64, "phase A pads a chunk to whole groups, one lane per row");

// Phase A of both per-query kernels: influence weight of every (neighbour, kernel point) pair of a chunk of <= KP_NCH
// neighbours, into the wave's LDS slice.  Two steps so that a query costs two dependent global round trips instead of
// two per group of four neighbours: (a) lane n fetches neighbour n's id and its centred position into LDS; (b) every
// lane owns kernel point (lane & 15) and walks the neighbours four at a time, reading only LDS.  (The kernel was bound
// by exactly that chain: ~6 waves per SIMD each waiting ~14 serial L2 round trips.)
template <bool CLOSEST>
__device__ __forceinline__ void kp_influence_weights(const float *__restrict__ support, const int64_t *__restrict__ nbr_row,
                                                     int cnt, int64_t M, float qx, float qy, float qz,
                                                     const float *__restrict__ kpts, int KP, int influence,
                                                     float inv_extent, float gden, float (*w)[KP_MAX],
                                                     float (*dd)[KP_MAX], float4 *rel, int *ids, int lane)
{
    const int cntg = (cnt + KP_GROUP - 1) / KP_GROUP * KP_GROUP;  // rows [cnt, cntg) become shadows (zero weights)
    if (lane < cntg) {
        const int64_t id = lane < cnt ? nbr_row[lane] : -1;
        const bool shadow = id < 0 || id >= M;
        float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!shadow) {
            r.x = support[id * 3 + 0] - qx;
            r.y = support[id * 3 + 1] - qy;
            r.z = support[id * 3 + 2] - qz;
        }
        ids[lane] = shadow ? -1 : (int)id;
        rel[lane] = r;
    }
    wave_lds_sync();
    const int k = lane & (KP_MAX - 1);
    const bool kreal = k < KP;
    const float kx = kreal ? kpts[k * 3 + 0] : 0.0f, ky = kreal ? kpts[k * 3 + 1] : 0.0f, kz = kreal ? kpts[k * 3 + 2] : 0.0f;
    for (int n = lane / KP_MAX; n < cntg; n += 64 / KP_MAX) {
        const float4 r = rel[n];
        float wv = 0.0f, d2 = 3.0e38f;
        if (ids[n] >= 0 && kreal) {
            const float dx = r.x - kx, dy = r.y - ky, dz = r.z - kz;
            d2 = (dx * dx + dy * dy) + dz * dz;
            if (influence == 0) wv = 1.0f;
            // linear: 1-ulp v_sqrt_f32 and a reciprocal multiply (features carry a 1e-5 tolerance; the correctly
            // rounded sqrt + divide sequences cost ~25 instructions per pair)
            else if (influence == 1) wv = fmaxf(1.0f - __builtin_amdgcn_sqrtf(d2) * inv_extent, 0.0f);
            else wv = expf(-d2 / gden);
        }
        w[n][k] = wv;
        if (CLOSEST) dd[n][k] = d2;
    }
    wave_lds_sync();
    if (CLOSEST) {  // only the closest kernel point keeps its influence (first minimum)
        if (lane < cnt) {
            int kb = 0;
            float best = dd[lane][0];
            for (int kk = 1; kk < KP; ++kk)
                if (dd[lane][kk] < best) {
                    best = dd[lane][kk];
                    kb = kk;
                }
            for (int kk = 0; kk < KP; ++kk)
                if (kk != kb) w[lane][kk] = 0.0f;
        }
        wave_lds_sync();
    }
}

template <bool CLOSEST>
__global__ __launch_bounds__(KP_BLOCK) void kpconv_weighted_kernel(
    const float *__restrict__ query, const float *__restrict__ support, const int64_t *__restrict__ nbr,
    const float *__restrict__ feat, const float *__restrict__ kpts, int64_t Nq, int64_t M, int Mn, int Cin, int KP,
    float extent, int influence, float *__restrict__ wf)
{
    __shared__ __attribute__((aligned(16))) float s_w[KP_BLOCK / 64][KP_NCH][KP_MAX];
    __shared__ __attribute__((aligned(16))) float4 s_rel[KP_BLOCK / 64][KP_NCH];
    __shared__ float s_d[CLOSEST ? KP_BLOCK / 64 : 1][CLOSEST ? KP_NCH : 1][KP_MAX];  // distances: closest mode only
    __shared__ int s_id[KP_BLOCK / 64][KP_NCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t qraw = (int64_t)blockIdx.x * (KP_BLOCK / 64) + wave;
    if (qraw >= Nq) return;  // wave-uniform; the kernel has no workgroup barrier
    const bool live = true;
    const int64_t q = qraw;
    const float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
    float(*w)[KP_MAX] = s_w[wave];
    float(*dd)[KP_MAX] = s_d[CLOSEST ? wave : 0];
    float4 *rel = s_rel[wave];
    int *ids = s_id[wave];
    const float sigma = extent * 0.3f;
    const float gden = 2.0f * sigma * sigma + 1e-9f;
    const float inv_extent = 1.0f / extent;

    const bool single_pass = Mn <= KP_NCH;  // then the weights of phase A serve every channel chunk
    for (int c0 = 0; c0 < Cin; c0 += 64) {
        float acc[KP_MAX];
#pragma unroll
        for (int k = 0; k < KP_MAX; ++k) acc[k] = 0.0f;
        const int c = c0 + lane;
        for (int n0 = 0; n0 < Mn; n0 += KP_NCH) {
            const int cnt = min(KP_NCH, Mn - n0);
            // ---- phase A: influence weight of every (neighbour, kernel point) pair of this chunk
            if (!(single_pass && c0 > 0))
                kp_influence_weights<CLOSEST>(support, nbr + q * Mn + n0, cnt, M, qx, qy, qz, kpts, KP, influence, inv_extent,
                                              gden, w, dd, rel, ids, lane);
            // ---- phase B: accumulate the neighbour rows into the KP accumulators (lanes over channels)
            if (c < Cin) {
                // KP_GROUP neighbour rows are requested before any is used: the accumulation was a chain of dependent
                // (LDS id -> global row) round trips, ~0.3 us per neighbour (measured with in-kernel clocks)
                const int cntg = (cnt + KP_GROUP - 1) / KP_GROUP * KP_GROUP;
                for (int g0 = 0; g0 < cntg; g0 += KP_GROUP) {
                    float v[KP_GROUP];
#pragma unroll
                    for (int u = 0; u < KP_GROUP; ++u)  // shadow rows carry zero weights: row 0 stands in for them
                        v[u] = feat[(size_t)max(ids[g0 + u], 0) * Cin + c];
#pragma unroll
                    for (int u = 0; u < KP_GROUP; ++u) {
                        const int n = g0 + u;
                        const float4 w0 = *reinterpret_cast<const float4 *>(&w[n][0]);
                        const float4 w1 = *reinterpret_cast<const float4 *>(&w[n][4]);
                        const float4 w2 = *reinterpret_cast<const float4 *>(&w[n][8]);
                        const float4 w3 = *reinterpret_cast<const float4 *>(&w[n][12]);
                        // explicit fused multiply-adds: the translation unit is built with contraction off for the
                        // distance expressions, the feature accumulation has no bit-exactness contract (1e-5 relative)
                        acc[0] = __builtin_fmaf(w0.x, v[u], acc[0]);   acc[1] = __builtin_fmaf(w0.y, v[u], acc[1]);
                        acc[2] = __builtin_fmaf(w0.z, v[u], acc[2]);   acc[3] = __builtin_fmaf(w0.w, v[u], acc[3]);
                        acc[4] = __builtin_fmaf(w1.x, v[u], acc[4]);   acc[5] = __builtin_fmaf(w1.y, v[u], acc[5]);
                        acc[6] = __builtin_fmaf(w1.z, v[u], acc[6]);   acc[7] = __builtin_fmaf(w1.w, v[u], acc[7]);
                        acc[8] = __builtin_fmaf(w2.x, v[u], acc[8]);   acc[9] = __builtin_fmaf(w2.y, v[u], acc[9]);
                        acc[10] = __builtin_fmaf(w2.z, v[u], acc[10]); acc[11] = __builtin_fmaf(w2.w, v[u], acc[11]);
                        acc[12] = __builtin_fmaf(w3.x, v[u], acc[12]); acc[13] = __builtin_fmaf(w3.y, v[u], acc[13]);
                        acc[14] = __builtin_fmaf(w3.z, v[u], acc[14]); acc[15] = __builtin_fmaf(w3.w, v[u], acc[15]);
                    }
                }
            }
            wave_lds_sync();
        }
        if (live && c < Cin) {
#pragma unroll
            for (int k = 0; k < KP_MAX; ++k)
                if (k < KP) wf[((size_t)q * KP + k) * Cin + c] = acc[k];
        }
    }
}


// The same product on the matrix pipe (sum aggregation, <= 64 neighbours): per query the weighted features are a small GEMM,
//   wf[k][c] = sum_n w[n][k] * f[n][c]          (16 kernel points x Mn neighbours x Cin channels),
// and the kernel above spends its time broadcasting w out of LDS (one ds_read_b128 per four weights and per 64 channel
// lanes: 128 reads = 1024 LDS cycles per query wave, 125 us over 65536 queries whatever Cin is) and on 512 wave-wide FMAs.
// v_mfma_f32_16x16x4_f32 takes A = w[kernel point = lane & 15][neighbour = 4 s + (lane >> 4)] -- exactly the layout phase A
// computes the influence weights in, so they stay in registers -- and B = f[neighbour][channel = lane & 15], a gather of
// four 64-byte row segments per load instruction; exact fp32 multiply-adds, the 16 x 16 result per channel block in four
// registers per lane.  LDS holds only the neighbours' ids and centred positions (1.3 KB per wave).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int KPM_NMAX = 64;   // neighbours per query (one lane each in step (a))
constexpr int KPM_CB = 4;      // 16-channel blocks accumulated together (64 channels per pass)

template <int SMAX>  // most MFMA steps of four neighbours: 8 (Mn <= 32) or 16 (Mn <= 64)
__global__ __launch_bounds__(KP_BLOCK) void kpconv_weighted_mfma_kernel(
    const float *__restrict__ query, const float *__restrict__ support, const int64_t *__restrict__ nbr,
    const float *__restrict__ feat, const float *__restrict__ kpts, int64_t Nq, int64_t M, int Mn, int Cin, int KP,
    float extent, int influence, int cpass, float *__restrict__ wf)
{
    __shared__ __attribute__((aligned(16))) float4 s_rel[KP_BLOCK / 64][KPM_NMAX];
    __shared__ int s_id[KP_BLOCK / 64][KPM_NMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * (KP_BLOCK / 64) + wave;
    if (q >= Nq) return;  // wave-uniform; no workgroup barrier below
    float4 *rel = s_rel[wave];
    int *ids = s_id[wave];
    const float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
    const float sigma = extent * 0.3f;
    const float gden = 2.0f * sigma * sigma + 1e-9f;
    const float inv_extent = 1.0f / extent;
    const int steps = (Mn + 3) / 4;  // MFMA steps of four neighbours (<= SMAX)
    {   // (a) lane n: neighbour n's id and centred position (rows past Mn and shadow neighbours: id -1, zero weights)
        const int64_t id = lane < Mn ? nbr[q * Mn + lane] : -1;
        const bool shadow = id < 0 || id >= M;
        float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!shadow) {
            r.x = support[id * 3 + 0] - qx;
            r.y = support[id * 3 + 1] - qy;
            r.z = support[id * 3 + 2] - qz;
        }
        ids[lane] = shadow ? -1 : (int)id;
        rel[lane] = r;
    }
    wave_lds_sync();
    // (b) influence weights in the A-operand layout: lane = (kernel point k = lane & 15, neighbour 4 s + (lane >> 4))
    const int k = lane & 15, nsub = lane >> 4, c16 = lane & 15;
    const bool kreal = k < KP;
    const float kx = kreal ? kpts[k * 3 + 0] : 0.0f, ky = kreal ? kpts[k * 3 + 1] : 0.0f, kz = kreal ? kpts[k * 3 + 2] : 0.0f;
    float a[SMAX];
    unsigned roff[SMAX];  // FLOAT offset of the neighbour row this lane gathers in step s (host: M * Cin < 2^30); a shadow
                          // neighbour has zero weights and row 0 stands in for it (as in the kernel above)
#pragma unroll
    for (int s = 0; s < SMAX; ++s) {
        a[s] = 0.0f;
        roff[s] = 0;
        if (s < steps) {
            const int n = 4 * s + nsub;
            const float4 r = rel[n];
            const int id = ids[n];
            roff[s] = (unsigned)max(id, 0) * (unsigned)Cin;
            if (id >= 0 && kreal) {
                const float dx = r.x - kx, dy = r.y - ky, dz = r.z - kz;
                const float d2 = (dx * dx + dy * dy) + dz * dz;
                if (influence == 0) a[s] = 1.0f;
                else if (influence == 1) a[s] = fmaxf(1.0f - __builtin_amdgcn_sqrtf(d2) * inv_extent, 0.0f);
                else a[s] = expf(-d2 / gden);
            }
        }
    }
    // (c) per pass of up to 64 channels: gather the rows (a 16-channel block per load, blocks past Cin skipped by wave-
    // uniform branches), accumulate on the matrix pipe.  Addresses: a wave-uniform base + a 32-bit lane offset.
    float *__restrict__ wq = wf + (size_t)q * KP * Cin;  // this query's (KP, Cin) output
    // (few queries with many channels -- the deep levels of a U-Net -- spread their channel passes over gridDim.y: a wave per
    //  (query, 64 channels) instead of eight serial passes in 27 waves)
    const int c_lo = (int)blockIdx.y * cpass, c_hi = min(Cin, c_lo + cpass);
    for (int c0 = c_lo; c0 < c_hi; c0 += 16 * KPM_CB) {
        const int nblk = min(KPM_CB, (c_hi - c0 + 15) / 16);  // (wave-uniform)
        f32x4 acc[KPM_CB];
#pragma unroll
        for (int b = 0; b < KPM_CB; ++b) acc[b] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        const float *__restrict__ fb = feat + c0;
        unsigned coff[KPM_CB];  // this lane's channel inside the pass, clamped to the row (lanes past Cin are not stored)
#pragma unroll
        for (int b = 0; b < KPM_CB; ++b) coff[b] = (unsigned)min(16 * b + c16, Cin - 1 - c0);
#pragma unroll
        for (int s0 = 0; s0 < SMAX; s0 += 4) {  // four steps' loads in flight (16 rows x up to 4 blocks)
            if (s0 < steps) {
                float v[4][KPM_CB];
#pragma unroll
                for (int b = 0; b < KPM_CB; ++b)
                    if (b < nblk) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u][b] = fb[roff[s0 + u] + coff[b]];
                    }
#pragma unroll
                for (int b = 0; b < KPM_CB; ++b)
                    if (b < nblk) {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s0 + u], v[u][b], acc[b], 0, 0, 0);
                    }
            }
        }
        // D[kernel point = 4 (lane >> 4) + j][channel = lane & 15]
#pragma unroll
        for (int b = 0; b < KPM_CB; ++b)
            if (b < nblk) {
                const int c = c0 + 16 * b + c16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kp = 4 * nsub + j;
                    if (kp < KP && c < Cin) wq[(unsigned)(kp * Cin + c)] = acc[b][j];
                }
            }
    }
}

}  // namespace tp3d

using namespace tp3d;

TP3D_EXPORT int tp3d_kpconv_weighted_f32(const float *query, const float *support, const int64_t *neighbors,
                                         const float *features, const float *k_points, int64_t Nq, int64_t M,
                                         int Mn, int Cin, int KP, float extent, int influence, int closest,
                                         float *weighted, void *stream)
{
    if (Nq < 0 || M < 0 || Mn < 0 || Cin <= 0 || KP <= 0 || influence < 0 || influence > 2) return TP3D_E_BADARG;
    if (KP > KP_MAX) return TP3D_E_TOOBIG;
    if (Nq == 0) return TP3D_OK;
    if (!query || !weighted || !k_points || (Mn > 0 && (!neighbors || !support || !features))) return TP3D_E_BADARG;
    if (M == 0 || Mn == 0)  // no support row to read: every neighbour is a shadow
        return zero_async(weighted, (size_t)Nq * KP * Cin * sizeof(float), (hipStream_t)stream);
    const int64_t blocks = (Nq + KP_BLOCK / 64 - 1) / (KP_BLOCK / 64);
    if (blocks > 0x7fffffff) return TP3D_E_TOOBIG;
    if (closest)
        hipLaunchKernelGGL(kpconv_weighted_kernel<true>, dim3((unsigned)blocks), dim3(KP_BLOCK), 0, (hipStream_t)stream,
                           query, support, neighbors, features, k_points, Nq, M, Mn, Cin, KP, extent, influence, weighted);
    else if (Mn <= KPM_NMAX && M * (int64_t)Cin < ((int64_t)1 << 30)) {  // (the reference configs ask for 25 ... 38 neighbours)
        const int passes = (Cin + 16 * KPM_CB - 1) / (16 * KPM_CB);
        const bool spread = passes > 1 && Nq <= 4096;  // latency-bound launches: one wave per (query, channel pass)
        const dim3 grid((unsigned)blocks, spread ? passes : 1);
        const int cpass = spread ? 16 * KPM_CB : Cin;
        if (Mn <= 32)
            hipLaunchKernelGGL(kpconv_weighted_mfma_kernel<8>, grid, dim3(KP_BLOCK), 0, (hipStream_t)stream, query, support,
                               neighbors, features, k_points, Nq, M, Mn, Cin, KP, extent, influence, cpass, weighted);
        else
            hipLaunchKernelGGL(kpconv_weighted_mfma_kernel<16>, grid, dim3(KP_BLOCK), 0, (hipStream_t)stream, query, support,
                               neighbors, features, k_points, Nq, M, Mn, Cin, KP, extent, influence, cpass, weighted);
    }
    else
        hipLaunchKernelGGL(kpconv_weighted_kernel<false>, dim3((unsigned)blocks), dim3(KP_BLOCK), 0, (hipStream_t)stream,
                           query, support, neighbors, features, k_points, Nq, M, Mn, Cin, KP, extent, influence, weighted);
    return check_launch();
}

// =====================================================================================================
// Backward of stage 1 with respect to the input features:
//   d_x[m, :] = sum over slots (q, n) with nbr[q,n] == m of  sum_k h(|(s_m - q) - K_k|) * d_wf[q, k, :]
// (reference: autograd through convolution_ops.py:92-98).  No float atomics: the neighbour table is inverted first
// (global counting sort over the Nq*Mn slots: integer histogram -> scan -> fill -> per-point sort of its short run,
// which makes the summation order ascending in (q, n) whatever the timing), then one wave per support point
// recomputes the KP influence weights of each of its slots and accumulates coalesced d_wf row segments.
namespace tp3d {

// bin of slot s: the table entry itself, or -- for a batch of per-cloud tables flattened into one -- the entry clamped
// to its cloud's bins plus the cloud's offset
__device__ __forceinline__ int64_t slot_bin(const int64_t *__restrict__ nbr, int64_t s, int64_t L, int64_t nbins)
{
    const int64_t m = nbr[s];
    return L > 0 ? min(max(m, (int64_t)0), nbins - 1) + (s / L) * nbins : m;
}

__global__ void nbr_hist_kernel(const int64_t *__restrict__ nbr, int64_t slots, int64_t M, int *__restrict__ cnt,
                                int64_t L, int64_t nbins)
{
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = slot_bin(nbr, s, L, nbins);
        if (m >= 0 && m < M) atomicAdd(&cnt[m], 1);
    }
}

// exclusive scan of cnt[0..M) into start[0..M] (one workgroup), cursor := start
__global__ __launch_bounds__(1024) void nbr_scan_kernel(const int *__restrict__ cnt, int64_t M, int *__restrict__ start,
                                                         int *__restrict__ cursor)
{
    __shared__ int s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t per = (M + 1023) / 1024;
    const int64_t k0 = min((int64_t)tid * per, M), k1 = min(k0 + per, M);
    int sum = 0;
    for (int64_t k = k0; k < k1; ++k) sum += cnt[k];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < 16; ++w) {
            const int v = s_w[w];
            s_w[w] = run;
            run += v;
        }
    }
    __syncthreads();
    int run = s_w[wave] + incl - sum;
    for (int64_t k = k0; k < k1; ++k) {
        const int v = cnt[k];
        start[k] = run;
        cursor[k] = run;
        run += v;
    }
    if (k1 == M) start[M] = run;  // every thread whose range ends at M holds the grand total
}

__global__ void nbr_fill_kernel(const int64_t *__restrict__ nbr, int64_t slots, int64_t M, int *__restrict__ cursor,
                                int *__restrict__ order, int64_t L, int64_t nbins)
{
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < slots; s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = slot_bin(nbr, s, L, nbins);
        if (m >= 0 && m < M) order[atomicAdd(&cursor[m], 1)] = (int)s;
    }
}

// canonical order inside every bin (ascending slot id): one lane per bin for bins of up to NBR_SMALL_BIN slots
// (insertion sort; the fill leaves them nearly sorted), a whole wave's bitonic network for larger ones -- a point that
// hundreds of slots reference (padded tails of dense ball queries) would otherwise be hundreds of dependent global
// round trips in one thread
constexpr int NBR_SMALL_BIN = 24;
// A bin of more than 1024 slots, sorted by one wave: 1024-slot runs through the bitonic network, then log2(runs) merge
// passes between `order` and `tmp` -- every lane merges an equal share of a run pair, its split found by bisection
// (merge path).  Slot ids are unique, so the result is the ascending order whatever the arrival order was.
// (one thread's insertion sort needed tens of seconds for a 52 800-slot bin)
__device__ void wave_merge_sort_bin(int *order, int *tmp, int n, int lane)
{
    for (int a = 0; a < n; a += 1024) wave_sort_bin<16>(order + a, min(1024, n - a), lane);
    int *src = order, *dst = tmp;
    for (int width = 1024; width < n; width <<= 1) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int lo = 0; lo < n; lo += 2 * width) {
            const int mid = min(lo + width, n), hi = min(lo + 2 * width, n);
            const int *A = src + lo, *B = src + mid;
            const int na = mid - lo, nb = hi - mid, total = hi - lo;
            const int per = (total + 63) / 64;
            const int o0 = min(lane * per, total), o1 = min(o0 + per, total);
            // i = how many of the first o0 outputs come from A: smallest i with A[i] > B[o0 - i - 1]
            int x = max(0, o0 - nb), y = min(o0, na);
            while (x < y) {
                const int i = (x + y) >> 1, j = o0 - i;
                if (j > 0 && A[i] < B[j - 1]) x = i + 1;
                else y = i;
            }
            int i = x, j = o0 - x;
            for (int o = o0; o < o1; ++o) {
                const bool from_a = j >= nb || (i < na && A[i] < B[j]);
                dst[lo + o] = from_a ? A[i++] : B[j++];
            }
        }
        int *t = src;
        src = dst;
        dst = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (src != order)
        for (int e = lane; e < n; e += 64) order[e] = src[e];
}

__global__ __launch_bounds__(256) void nbr_sort_kernel(const int *__restrict__ start, int64_t M, int *order, int *tmp)
{
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int s0 = 0, s1 = 0;
    if (m < M) {
        s0 = start[m];
        s1 = start[m + 1];
    }
    const int n = s1 - s0;
    if (n <= NBR_SMALL_BIN || (n > 1024 && !tmp)) {
        for (int a = s0 + 1; a < s1; ++a) {
            const int v = order[a];
            int p = a;
            while (p > s0 && order[p - 1] > v) {
                order[p] = order[p - 1];
                --p;
            }
            order[p] = v;
        }
    }
    unsigned long long big = __ballot(n > NBR_SMALL_BIN && n <= 1024);
    while (big) {  // wave-uniform
        const int l = __builtin_ctzll(big);
        big &= big - 1;
        const int blo = __builtin_amdgcn_readlane(s0, l), bn = __builtin_amdgcn_readlane(n, l);
        if (bn <= 64) wave_sort_bin<1>(order + blo, bn, lane);
        else if (bn <= 128) wave_sort_bin<2>(order + blo, bn, lane);
        else if (bn <= 256) wave_sort_bin<4>(order + blo, bn, lane);
        else if (bn <= 512) wave_sort_bin<8>(order + blo, bn, lane);
        else wave_sort_bin<16>(order + blo, bn, lane);
    }
    if (tmp) {
        unsigned long long giant = __ballot(n > 1024);
        while (giant) {  // wave-uniform
            const int l = __builtin_ctzll(giant);
            giant &= giant - 1;
            const int blo = __builtin_amdgcn_readlane(s0, l), bn = __builtin_amdgcn_readlane(n, l);
            wave_merge_sort_bin(order + blo, tmp + blo, bn, lane);
        }
    }
}

// Inverse of an index table over any number of workgroups: for every bin m the slots that reference it, ascending
// (integer histogram -> scan -> fill -> per-bin insertion sort of its short run).  Entries outside [0, M) are skipped.
// cnt, cursor: M ints; start: M + 1 ints; order: `slots` ints.
int invert_table(const int64_t *idx, int64_t slots, int64_t M, int *cnt, int *start, int *cursor, int *order,
                 hipStream_t s, int64_t per_cloud_slots, int64_t per_cloud_bins, int *merge_tmp)
{
    if (int rc = zero_async(cnt, (size_t)M * 4, s)) return rc;
    const unsigned gs = (unsigned)std::min<int64_t>((slots + 255) / 256, 4096);
    hipLaunchKernelGGL(nbr_hist_kernel, dim3(gs), dim3(256), 0, s, idx, slots, M, cnt, per_cloud_slots, per_cloud_bins);
    hipLaunchKernelGGL(nbr_scan_kernel, dim3(1), dim3(1024), 0, s, cnt, M, start, cursor);
    hipLaunchKernelGGL(nbr_fill_kernel, dim3(gs), dim3(256), 0, s, idx, slots, M, cursor, order, per_cloud_slots,
                       per_cloud_bins);
    hipLaunchKernelGGL(nbr_sort_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, start, M, order, merge_tmp);
    return check_launch();
}

// workspace: tp3d_kpconv_bwd_workspace_bytes(M, slots)
static int invert_neighbors(const int64_t *neighbors, int64_t slots, int64_t M, void *workspace, int **start_out,
                            int **order_out, hipStream_t s, bool ready = false)
{
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    char *p = static_cast<char *>(workspace);
    int *cnt = reinterpret_cast<int *>(p);
    int *start = reinterpret_cast<int *>(p + up((size_t)M * 4));
    int *cursor = reinterpret_cast<int *>(p + up((size_t)M * 4) + up((size_t)(M + 1) * 4));
    int *order = reinterpret_cast<int *>(p + up((size_t)M * 4) + up((size_t)(M + 1) * 4) + up((size_t)M * 4));
    // second buffer of the run merge that sorts a bin of more than 1024 slots (a hub support point: many queries padding
    // onto one index, duplicated points); without it such a bin fell back to one lane's insertion sort -- tens of seconds
    int *merge_tmp = reinterpret_cast<int *>(p + up((size_t)M * 4) + up((size_t)(M + 1) * 4) + up((size_t)M * 4) +
                                             up((size_t)slots * 4));
    *start_out = start;
    *order_out = order;
    if (ready) return TP3D_OK;  // the caller kept the table of an earlier call on the same neighbours
    return invert_table(neighbors, slots, M, cnt, start, cursor, order, s, 0, 0, merge_tmp);
}

// Strided shortcut of ResnetBBlock (reference modules/KPConv/blocks.py:206-210): max over each query's neighbours of
// the support features, a shadow neighbour (-1 or >= M) contributing the zero row.  arg = winning slot (first max).
__global__ __launch_bounds__(256) void nbr_maxpool_kernel(const float *__restrict__ x, const int64_t *__restrict__ nbr,
                                                           int64_t Nq, int64_t M, int Mn, int C, float *__restrict__ out,
                                                           int *__restrict__ arg)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= Nq * C) return;
    const int64_t q = t / C;
    const int c = (int)(t - q * C);
    float best = -3.4028235e38f;
    int barg = 0;
    for (int n = 0; n < Mn; ++n) {
        const int64_t m = nbr[q * Mn + n];
        const float v = (m >= 0 && m < M) ? x[m * C + c] : 0.0f;
        if (v > best) {
            best = v;
            barg = n;
        }
    }
    out[t] = best;
    if (arg) arg[t] = barg;
}

// d_x[m, c] = sum over the slots (q, n) referencing m (ascending) with arg[q, c] == n of g[q, c]; one wave per point
__global__ __launch_bounds__(KP_BLOCK) void nbr_maxpool_bwd_kernel(const float *__restrict__ g, const int *__restrict__ arg,
                                                                    const int *__restrict__ start,
                                                                    const int *__restrict__ order, int64_t M, int Mn,
                                                                    int C, float *__restrict__ d_x)
{
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * (KP_BLOCK / 64) + (threadIdx.x >> 6);
    if (m >= M) return;
    const int s0 = start[m], s1 = start[m + 1];
    for (int c = lane; c < C; c += 64) {
        float acc = 0.0f;
        for (int j = s0; j < s1; ++j) {
            const int slot = order[j];
            const int64_t q = slot / Mn;
            const int n = slot - (int)q * Mn;
            if (arg[q * C + c] == n) acc += g[q * C + c];
        }
        d_x[m * C + c] = acc;
    }
}

// Backward, step 1 (one wave per query): per-slot gradient rows
//   g[q, n, :] = sum_k h(|(s[nbr[q,n]] - q) - K_k|) * d_wf[q, k, :]
// d_wf[q] (KP x Cin) is read ONCE into registers (lanes over channels) and combined with the KP x Mn influence weights
// recomputed in LDS exactly like the forward pass.  (The first version walked, per support point, every slot that
// references it and re-read the whole d_wf[q] block each time: Mn times the traffic.)
template <bool CLOSEST>
__global__ __launch_bounds__(KP_BLOCK) void kpconv_bwd_slots_kernel(
    const float *__restrict__ query, const float *__restrict__ support, const int64_t *__restrict__ nbr,
    const float *__restrict__ kpts, const float *__restrict__ d_wf, int64_t Nq, int64_t M, int Mn, int Cin, int KP,
    float extent, int influence, float *__restrict__ g)
{
    __shared__ __attribute__((aligned(16))) float s_w[KP_BLOCK / 64][KP_NCH][KP_MAX];
    __shared__ __attribute__((aligned(16))) float4 s_rel[KP_BLOCK / 64][KP_NCH];
    __shared__ float s_d[CLOSEST ? KP_BLOCK / 64 : 1][CLOSEST ? KP_NCH : 1][KP_MAX];
    __shared__ int s_id[KP_BLOCK / 64][KP_NCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t qraw = (int64_t)blockIdx.x * (KP_BLOCK / 64) + wave;
    if (qraw >= Nq) return;  // wave-uniform; the kernel has no workgroup barrier
    const bool live = true;
    const int64_t q = qraw;
    const float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
    float(*w)[KP_MAX] = s_w[wave];
    float(*dd)[KP_MAX] = s_d[CLOSEST ? wave : 0];
    float4 *rel = s_rel[wave];
    int *ids = s_id[wave];
    const float sigma = extent * 0.3f;
    const float gden = 2.0f * sigma * sigma + 1e-9f;
    const float inv_extent = 1.0f / extent;
    const bool single_pass = Mn <= KP_NCH;
    for (int c0 = 0; c0 < Cin; c0 += 64) {
        const int c = c0 + lane;
        float dk[KP_MAX];
#pragma unroll
        for (int k = 0; k < KP_MAX; ++k) dk[k] = (k < KP && c < Cin) ? d_wf[((size_t)q * KP + k) * Cin + c] : 0.0f;
        for (int n0 = 0; n0 < Mn; n0 += KP_NCH) {
            const int cnt = min(KP_NCH, Mn - n0);
            if (!(single_pass && c0 > 0))
                kp_influence_weights<CLOSEST>(support, nbr + q * Mn + n0, cnt, M, qx, qy, qz, kpts, KP, influence, inv_extent,
                                              gden, w, dd, rel, ids, lane);
            if (live && c < Cin) {
#pragma unroll 5
                for (int n = 0; n < cnt; ++n) {
                    const float4 w0 = *reinterpret_cast<const float4 *>(&w[n][0]);
                    const float4 w1 = *reinterpret_cast<const float4 *>(&w[n][4]);
                    const float4 w2 = *reinterpret_cast<const float4 *>(&w[n][8]);
                    const float4 w3 = *reinterpret_cast<const float4 *>(&w[n][12]);
                    // same k order as the first version's accumulation (ascending k), fused multiply-adds
                    float acc = w0.x * dk[0];
                    acc = __builtin_fmaf(w0.y, dk[1], acc);   acc = __builtin_fmaf(w0.z, dk[2], acc);
                    acc = __builtin_fmaf(w0.w, dk[3], acc);   acc = __builtin_fmaf(w1.x, dk[4], acc);
                    acc = __builtin_fmaf(w1.y, dk[5], acc);   acc = __builtin_fmaf(w1.z, dk[6], acc);
                    acc = __builtin_fmaf(w1.w, dk[7], acc);   acc = __builtin_fmaf(w2.x, dk[8], acc);
                    acc = __builtin_fmaf(w2.y, dk[9], acc);   acc = __builtin_fmaf(w2.z, dk[10], acc);
                    acc = __builtin_fmaf(w2.w, dk[11], acc);  acc = __builtin_fmaf(w3.x, dk[12], acc);
                    acc = __builtin_fmaf(w3.y, dk[13], acc);  acc = __builtin_fmaf(w3.z, dk[14], acc);
                    acc = __builtin_fmaf(w3.w, dk[15], acc);
                    g[((size_t)q * Mn + n0 + n) * Cin + c] = acc;
                }
            }
            wave_lds_sync();
        }
    }
}

// Backward, step 2 (one wave per support point): d_x[m, :] = sum of the slot rows that reference m, ascending slot
__global__ __launch_bounds__(KP_BLOCK) void kpconv_bwd_gather_kernel(const float *__restrict__ g,
                                                                      const int *__restrict__ start,
                                                                      const int *__restrict__ order, int64_t M, int Cin,
                                                                      float *__restrict__ d_x)
{
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * (KP_BLOCK / 64) + (threadIdx.x >> 6);
    if (m >= M) return;  // wave-uniform; no workgroup barrier in this kernel
    const int s0 = start[m], s1 = start[m + 1];
    for (int c0 = 0; c0 < Cin; c0 += 64) {
        const int c = min(c0 + lane, Cin - 1);
        float acc = 0.0f;
        // the run's slot ids are fetched 64 at a time with one coalesced load and handed out by v_readlane, so the row
        // reads (four in flight) no longer wait for a dependent index load each; summed in slot order
        for (int j0 = s0; j0 < s1; j0 += 64) {
            const int cnt = min(64, s1 - j0);
            const int mine = lane < cnt ? order[j0 + lane] : 0;
            int t = 0;
            for (; t + 4 <= cnt; t += 4) {
                const int r0 = __builtin_amdgcn_readlane(mine, t), r1 = __builtin_amdgcn_readlane(mine, t + 1);
                const int r2 = __builtin_amdgcn_readlane(mine, t + 2), r3 = __builtin_amdgcn_readlane(mine, t + 3);
                const float v0 = g[(size_t)r0 * Cin + c], v1 = g[(size_t)r1 * Cin + c];
                const float v2 = g[(size_t)r2 * Cin + c], v3 = g[(size_t)r3 * Cin + c];
                acc = (((acc + v0) + v1) + v2) + v3;
            }
            for (; t < cnt; ++t) acc += g[(size_t)__builtin_amdgcn_readlane(mine, t) * Cin + c];
        }
        if (c0 + lane < Cin) d_x[(size_t)m * Cin + c0 + lane] = acc;
    }
}

}  // namespace tp3d

TP3D_EXPORT size_t tp3d_kpconv_bwd_workspace_bytes(int64_t M, int64_t slots)
{
    if (M < 0 || slots < 0) return 0;
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    return up((size_t)M * 4) + up((size_t)(M + 1) * 4) + up((size_t)M * 4) + 2 * up((size_t)slots * 4);  // + merge buffer
}

TP3D_EXPORT size_t tp3d_kpconv_grad_workspace_bytes(int64_t M, int64_t slots, int Cin)
{
    if (M < 0 || slots < 0 || Cin <= 0) return 0;
    return ((size_t)slots * Cin * 4 + 15) & ~(size_t)15;  // the per-slot gradient rows
}

TP3D_EXPORT int tp3d_kpconv_bwd_features_f32(const float *query, const float *support, const int64_t *neighbors,
                                             const float *k_points, const float *d_weighted, int64_t Nq, int64_t M,
                                             int Mn, int Cin, int KP, float extent, int influence, int closest,
                                             float *d_features, void *inverse, size_t inverse_bytes, int inverse_ready,
                                             void *workspace, size_t workspace_bytes, void *stream)
{
    if (Nq < 0 || M < 0 || Mn < 0 || Cin <= 0 || KP <= 0 || influence < 0 || influence > 2) return TP3D_E_BADARG;
    if (KP > KP_MAX) return TP3D_E_TOOBIG;
    if (M == 0) return TP3D_OK;
    if (!d_features) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t slots = Nq * Mn;
    if (slots == 0) return zero_async(d_features, (size_t)M * Cin * sizeof(float), s);
    if (!query || !support || !neighbors || !k_points || !d_weighted || !workspace || !inverse) return TP3D_E_BADARG;
    if (slots > INT32_MAX || M > INT32_MAX / 2) return TP3D_E_TOOBIG;
    if (inverse_bytes < tp3d_kpconv_bwd_workspace_bytes(M, slots)) return TP3D_E_BADARG;
    if (workspace_bytes < tp3d_kpconv_grad_workspace_bytes(M, slots, Cin)) return TP3D_E_BADARG;
    int *start = nullptr, *order = nullptr;
    if (int rc = invert_neighbors(neighbors, slots, M, inverse, &start, &order, s, inverse_ready != 0)) return rc;
    float *g = static_cast<float *>(workspace);
    const dim3 qgrid((unsigned)((Nq + KP_BLOCK / 64 - 1) / (KP_BLOCK / 64)));
    if (closest)
        hipLaunchKernelGGL(kpconv_bwd_slots_kernel<true>, qgrid, dim3(KP_BLOCK), 0, s, query, support, neighbors, k_points,
                           d_weighted, Nq, M, Mn, Cin, KP, extent, influence, g);
    else
        hipLaunchKernelGGL(kpconv_bwd_slots_kernel<false>, qgrid, dim3(KP_BLOCK), 0, s, query, support, neighbors, k_points,
                           d_weighted, Nq, M, Mn, Cin, KP, extent, influence, g);
    hipLaunchKernelGGL(kpconv_bwd_gather_kernel, dim3((unsigned)((M + KP_BLOCK / 64 - 1) / (KP_BLOCK / 64))),
                       dim3(KP_BLOCK), 0, s, g, start, order, M, Cin, d_features);
    return check_launch();
}

TP3D_EXPORT int tp3d_nbr_maxpool_fwd_f32(const float *x, const int64_t *neighbors, int64_t Nq, int64_t M, int Mn, int C,
                                         float *out, int32_t *argmax, void *stream)
{
    if (Nq < 0 || M < 0 || Mn <= 0 || C <= 0) return TP3D_E_BADARG;
    if (Nq == 0) return TP3D_OK;
    if (!neighbors || !out || (M > 0 && !x)) return TP3D_E_BADARG;
    const int64_t blocks = (Nq * C + 255) / 256;
    if (blocks > 0x7fffffff) return TP3D_E_TOOBIG;
    hipLaunchKernelGGL(tp3d::nbr_maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, neighbors, Nq,
                       M, Mn, C, out, argmax);
    return tp3d::check_launch();
}

TP3D_EXPORT int tp3d_nbr_maxpool_bwd_f32(const float *grad_out, const int32_t *argmax, const int64_t *neighbors, int64_t Nq,
                                         int64_t M, int Mn, int C, float *d_x, void *inverse, size_t inverse_bytes,
                                         int inverse_ready, void *stream)
{
    if (Nq < 0 || M < 0 || Mn <= 0 || C <= 0) return TP3D_E_BADARG;
    if (M == 0) return TP3D_OK;
    if (!d_x) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t slots = Nq * Mn;
    if (slots == 0) return tp3d::zero_async(d_x, (size_t)M * C * sizeof(float), s);
    if (!grad_out || !argmax || !neighbors || !inverse) return TP3D_E_BADARG;
    if (slots > INT32_MAX || M > INT32_MAX / 2) return TP3D_E_TOOBIG;
    if (inverse_bytes < tp3d_kpconv_bwd_workspace_bytes(M, slots)) return TP3D_E_BADARG;
    int *start = nullptr, *order = nullptr;
    if (int rc = tp3d::invert_neighbors(neighbors, slots, M, inverse, &start, &order, s, inverse_ready != 0)) return rc;
    hipLaunchKernelGGL(tp3d::nbr_maxpool_bwd_kernel, dim3((unsigned)((M + tp3d::KP_BLOCK / 64 - 1) / (tp3d::KP_BLOCK / 64))),
                       dim3(tp3d::KP_BLOCK), 0, s, grad_out, argmax, start, order, M, Mn, C, d_x);
    return tp3d::check_launch();
}
