// Channel-last ("rows") kernels of the grouped-MLP aggregation: everything around the 1x1-conv GEMMs of
// PointNetMSGDown / DenseFPModule / GlobalDenseBaseModule, with activations stored as (rows, C) row-major
// so that every access is a coalesced row segment and the GEMMs need no NCHW<->NHWC transposes.
//
// Reference semantics restated here (fp32):
//   torch_points3d/modules/pointnet2/dense.py:36-54   group -> centre (-> /radius) -> concat [xyz, feats]
//   torch_points3d/core/common_modules/dense_modules.py:5-29  Conv2d 1x1 (bias=False) -> BatchNorm2d -> LeakyReLU
//   torch_points3d/modules/pointnet2/dense.py:72-73   max over nsample
//   torch_points3d/core/base_conv/dense.py:132-144    inverse-distance 3-NN interpolation (+ skip concat :117)
// HBM-bound: each kernel reads and writes every activation byte at most once per pass.
#include <type_traits>

#include "tp3d_common.h"

namespace tp3d {

constexpr int RW_BLOCK = 256;

// -------------------------------------------------------------------------------------------------
// group + centre + concat:  out[(b,j,s), :] = [ (pos[b,idx] - new_pos[b,j]) (/ radius) , x_cl[b,idx,:] ]
__global__ __launch_bounds__(RW_BLOCK) void group_concat_fwd_kernel(const float *__restrict__ pos,
                                                                     const float *__restrict__ new_pos,
                                                                     const float *__restrict__ x_cl,
                                                                     const int64_t *__restrict__ idx, int N, int np,
                                                                     int ns, int C, int ld, float radius,
                                                                     int normalize, int64_t total,
                                                                     float *__restrict__ out)
{
    const int Cw = C + 3;  // columns [Cw, ld) are zero padding (keeps rows 16-byte aligned for the GEMMs)
    for (int64_t e = (int64_t)blockIdx.x * RW_BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * RW_BLOCK) {
        const int64_t row = e / ld;
        const int c = (int)(e - row * ld);
        if (c >= Cw) {
            out[e] = 0.0f;
            continue;
        }
        const int64_t bj = row / ns;          // b*np + j
        const int b = (int)(bj / np);
        const int k = min(max((int)idx[row], 0), N - 1);
        float v;
        if (c < 3) {
            v = pos[((size_t)b * N + k) * 3 + c] - new_pos[bj * 3 + c];
            if (normalize) v = v / radius;  // the reference divides (modules/pointnet2/dense.py:41-42)
        } else {
            v = x_cl[((size_t)b * N + k) * C + (c - 3)];
        }
        out[e] = v;
    }
}

// same, four output columns per thread (ld % 4 == 0): one row decode and one index fetch per float4 stored; the
// feature columns of a quad are contiguous in x_cl (offset by the three xyz columns, hence 4-byte aligned only)
struct __attribute__((packed, aligned(4))) f4u {
    float v[4];
};
__global__ __launch_bounds__(RW_BLOCK) void group_concat_fwd4_kernel(const float *__restrict__ pos,
                                                                      const float *__restrict__ new_pos,
                                                                      const float *__restrict__ x_cl,
                                                                      const int64_t *__restrict__ idx, int N, int np,
                                                                      int ns, int C, int ld, float radius,
                                                                      int normalize, int64_t total4,
                                                                      float *__restrict__ out)
{
    const int Cw = C + 3, q = ld >> 2;
    for (int64_t e4 = (int64_t)blockIdx.x * RW_BLOCK + threadIdx.x; e4 < total4; e4 += (int64_t)gridDim.x * RW_BLOCK) {
        const int64_t row = e4 / q;
        const int c = (int)(e4 - row * q) * 4;
        const int64_t bj = row / ns;  // b*np + j
        const int b = (int)(bj / np);
        const int k = min(max((int)idx[row], 0), N - 1);
        float t[4];
        if (c == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float v = pos[((size_t)b * N + k) * 3 + j] - new_pos[bj * 3 + j];
                if (normalize) v = v / radius;  // the reference divides (modules/pointnet2/dense.py:41-42)
                t[j] = v;
            }
            t[3] = C > 0 ? x_cl[((size_t)b * N + k) * C] : 0.0f;
        } else if (c + 3 < Cw) {  // four feature columns c-3 .. c
            const f4u f = *reinterpret_cast<const f4u *>(x_cl + ((size_t)b * N + k) * C + (c - 3));
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = f.v[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = (c + j < Cw) ? x_cl[((size_t)b * N + k) * C + (c + j - 3)] : 0.0f;
        }
        *reinterpret_cast<float4 *>(out + e4 * 4) = make_float4(t[0], t[1], t[2], t[3]);
    }
}

__device__ __forceinline__ float rl_f(float x, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}

// grad_x_cl[b,k,:] = sum over slots l (ascending) with idx[b,l]==k of grad_rows[(b,l), col0 + :]   (CSR gather)
// one wave per destination point, lanes over channels: every read is a contiguous row segment.  NP = channel slots per
// lane (c = lane + 64 p): a row of up to 64 NP channels is fetched in one traversal of the run, so 4 NP independent
// loads are in flight per lane (two traversals of 64 channels each ran the 128-channel decoder tables at 1.6 TB/s).
//
// acc[p] += sum over the slots [lo, hi) of the run, in slot order, of (weight *) row[c[p]]
template <int NP>
__device__ __forceinline__ void gather_run(const float *__restrict__ base, const int *__restrict__ od,
                                           const float *__restrict__ ws, int lo, int hi, int ld, const int (&c)[NP],
                                           float (&acc)[NP], int lane)
{
    // the run's (row, weight) pairs are fetched 64 at a time with one coalesced load, then broadcast lane by
    // lane (v_readlane), so independent row reads are in flight instead of a dependent index->row chain
    for (int j0 = lo; j0 < hi; j0 += 64) {
        const int cnt = min(64, hi - j0);
        const int my_r = (lane < cnt) ? od[j0 + lane] : 0;
        const float my_w = (ws && lane < cnt) ? ws[j0 + lane] : 1.0f;
        int t = 0;
        // U rows per step, all their loads issued before the first add (latency per step counts on a long run)
        auto take = [&](auto utag) {
            constexpr int U = decltype(utag)::value;
            int r[U];
            float wv[U], v[U][NP];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                r[u] = __builtin_amdgcn_readlane(my_r, t + u);
                wv[u] = ws ? rl_f(my_w, t + u) : 1.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < NP; ++p) v[u][p] = base[(size_t)r[u] * ld + c[p]];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < NP; ++p) acc[p] = acc[p] + (ws ? wv[u] * v[u][p] : v[u][p]);  // slot order
            t += U;
        };
        while (t + 8 <= cnt) take(std::integral_constant<int, 8>());
        if (t + 4 <= cnt) take(std::integral_constant<int, 4>());
        // tail of 1..3 rows (most runs of a grouping table are that short): requested together, summed in order
        const int rem = cnt - t;
        if (rem > 0) {
            const int r0 = __builtin_amdgcn_readlane(my_r, t);
            const int r1 = __builtin_amdgcn_readlane(my_r, min(t + 1, cnt - 1));
            const int r2 = __builtin_amdgcn_readlane(my_r, min(t + 2, cnt - 1));
            const float w0 = ws ? rl_f(my_w, t) : 1.0f, w1 = ws ? rl_f(my_w, min(t + 1, cnt - 1)) : 1.0f,
                        w2 = ws ? rl_f(my_w, min(t + 2, cnt - 1)) : 1.0f;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const float v0 = base[(size_t)r0 * ld + c[p]], v1 = base[(size_t)r1 * ld + c[p]],
                            v2 = base[(size_t)r2 * ld + c[p]];
                acc[p] = acc[p] + (ws ? w0 * v0 : v0);
                if (rem > 1) acc[p] = acc[p] + (ws ? w1 * v1 : v1);
                if (rem > 2) acc[p] = acc[p] + (ws ? w2 * v2 : v2);
            }
        }
    }
}

// Runs longer than HUB_MIN slots are left to rows_gather_hub_kernel (hub_min > 0): the first hit of a padded ball query
// collects every padding slot of every ball it opens -- 1295 slots on a 128-centre, 128-slot table over 512 points --
// and one wave walking that run alone held the whole launch (0.65 ms where a table of the same size without hubs takes
// 0.14 ms).
constexpr int HUB_MIN = 128;
constexpr int HUB_BLOCK = 1024;

template <int NP>
__global__ __launch_bounds__(RW_BLOCK) void rows_gather_sum_kernel(const float *__restrict__ grad_rows,
                                                                    const int *__restrict__ start,
                                                                    const int *__restrict__ order,
                                                                    const float *__restrict__ wsorted, int nbins,
                                                                    int L, int rows_per_cloud, int ld, int col0,
                                                                    int C, float *__restrict__ out, int flat, int hub_min)
{
    const int lane = threadIdx.x & 63;
    const int64_t dest = (int64_t)blockIdx.x * (RW_BLOCK / 64) + (threadIdx.x >> 6);  // destination point k
    const int b = blockIdx.y;
    if (dest >= nbins) return;
    // flat: one table over the whole batch (bins b*nbins + k, entries are batch-wide row ids); else one per cloud
    const int *st = flat ? start + (size_t)b * nbins : start + (size_t)b * (nbins + 1);
    const int *od = flat ? order : order + (size_t)b * L;
    const float *ws = wsorted ? (flat ? wsorted : wsorted + (size_t)b * L) : nullptr;
    const int lo = st[dest], hi = st[dest + 1];
    if (hub_min > 0 && hi - lo > hub_min) return;  // a hub: rows_gather_hub_kernel
    const float *base = grad_rows + (flat ? (size_t)0 : (size_t)b * rows_per_cloud * ld) + col0;
    for (int c0 = 0; c0 < C; c0 += 64 * NP) {
        int c[NP];
        float acc[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            c[p] = min(c0 + p * 64 + lane, C - 1);
            acc[p] = 0.0f;
        }
        gather_run<NP>(base, od, ws, lo, hi, ld, c, acc, lane);
#pragma unroll
        for (int p = 0; p < NP; ++p)
            if (c0 + p * 64 + lane < C) out[((size_t)b * nbins + dest) * C + c0 + p * 64 + lane] = acc[p];
    }
}

// hubs[0] = number of destinations with more than HUB_MIN slots, hubs[1..] = their ids b * nbins + k (any order)
__global__ __launch_bounds__(256) void find_hubs_kernel(const int *__restrict__ start, int B, int nbins, int flat,
                                                        int *__restrict__ hubs)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)B * nbins) return;
    const int b = (int)(e / nbins), k = (int)(e - (int64_t)b * nbins);
    const int *st = flat ? start + (size_t)b * nbins : start + (size_t)b * (nbins + 1);
    if (st[k + 1] - st[k] > HUB_MIN) hubs[1 + atomicAdd(&hubs[0], 1)] = (int)e;
}

// One workgroup of 16 waves per hub: the run is cut into 16 contiguous pieces, one per wave (each summed in slot order),
// and the pieces are added in piece order -- a fixed association, so the result is reproducible run to run.
template <int NP>
__global__ __launch_bounds__(HUB_BLOCK) void rows_gather_hub_kernel(const float *__restrict__ grad_rows,
                                                                     const int *__restrict__ start,
                                                                     const int *__restrict__ order,
                                                                     const float *__restrict__ wsorted,
                                                                     const int *__restrict__ hubs, int nbins, int L,
                                                                     int rows_per_cloud, int ld, int col0, int C,
                                                                     float *__restrict__ out, int flat)
{
    constexpr int NW = HUB_BLOCK / 64;
    __shared__ float part[NW][64 * NP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int count = hubs[0];
    for (int h = blockIdx.x; h < count; h += gridDim.x) {
        const int e = hubs[1 + h];
        const int b = e / nbins, dest = e - b * nbins;
        const int *st = flat ? start + (size_t)b * nbins : start + (size_t)b * (nbins + 1);
        const int *od = flat ? order : order + (size_t)b * L;
        const float *ws = wsorted ? (flat ? wsorted : wsorted + (size_t)b * L) : nullptr;
        const int lo = st[dest], hi = st[dest + 1];
        const int per = ((hi - lo + NW - 1) / NW + 7) & ~7;  // slots per wave, a multiple of the 8-row step
        const int wlo = min(lo + wave * per, hi), whi = min(wlo + per, hi);
        const float *base = grad_rows + (flat ? (size_t)0 : (size_t)b * rows_per_cloud * ld) + col0;
        for (int c0 = 0; c0 < C; c0 += 64 * NP) {
            int c[NP];
            float acc[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                c[p] = min(c0 + p * 64 + lane, C - 1);
                acc[p] = 0.0f;
            }
            gather_run<NP>(base, od, ws, wlo, whi, ld, c, acc, lane);
#pragma unroll
            for (int p = 0; p < NP; ++p) part[wave][p * 64 + lane] = acc[p];
            __syncthreads();
            for (int x = threadIdx.x; x < 64 * NP; x += HUB_BLOCK) {
                float sum = part[0][x];
                for (int w = 1; w < NW; ++w) sum = sum + part[w][x];
                const int cc = c0 + x;  // x = p * 64 + lane
                if (cc < C) out[((size_t)b * nbins + dest) * C + cc] = sum;
            }
            __syncthreads();
        }
    }
}

// -------------------------------------------------------------------------------------------------
// per-channel statistics of Y (M, C): partial (sum, sum of squares) per row chunk, then a small finalize.
// A workgroup is a (column-thread x row-lane) tile: V floats per column thread (float4 when C % 4 == 0), the
// row lanes stride over the chunk with 4 independent rows in flight each, so every wave keeps several KiB
// of coalesced loads outstanding (the first version had one 256-B row per wave in flight: 0.46 TB/s).
constexpr int ST_ROWS = 256;        // rows per workgroup, large matrices
constexpr int ST_ROWS_SMALL = 32;   // small matrices: the pass is a chain of dependent loads per thread, so more and
                                    // shorter workgroups (a 4096-row layer took 27 us with 16 workgroups of 256 rows)
constexpr int64_t ST_SMALL_BELOW = 131072;
static inline int stat_rows(int64_t R) { return R >= ST_SMALL_BELOW ? ST_ROWS : ST_ROWS_SMALL; }

struct StatTile {
    int colthreads, rowlanes, gridx;
};
static inline StatTile stat_tile(int C, int V)
{
    int cols = (C + V - 1) / V;  // column threads needed
    int ct = 1;
    while (ct < cols && ct < 64) ct <<= 1;
    StatTile t;
    t.colthreads = ct;
    t.rowlanes = RW_BLOCK / ct;
    t.gridx = (cols + ct - 1) / ct;
    return t;
}

// MODE 0: a += d, q += d*d with d = y - K, K = the chunk's first row of that column   (forward statistics; the
//         shift keeps the sums free of the cancellation E[y^2] - E[y]^2 suffers when |mean| >> std);
//         partial[chunk] = 4 rows of C: sum d, sum d^2, K, rows in the chunk  -> bn_finalize_kernel (Chan's merge)
// MODE 1: a += dz, q += dz*yhat, dz = dA*act'   (backward reductions; partial[chunk] = 2 rows of C)
template <int V, int MODE>
__global__ __launch_bounds__(RW_BLOCK) void colreduce_partial_kernel(
    const float *__restrict__ Y, const float *__restrict__ dA, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd, float slope,
    int64_t M, int C, int colthreads, int chunk_rows, float *__restrict__ partial /*[chunks][2][C]*/, int reverse = 0)
{
    __shared__ float s1[RW_BLOCK * V], s2[RW_BLOCK * V];
    const int ct = threadIdx.x % colthreads, rl = threadIdx.x / colthreads;
    const int rowlanes = RW_BLOCK / colthreads;
    const int c = (blockIdx.x * colthreads + ct) * V;
    // MODE 1 can walk the chunks LAST ROWS FIRST: when dA was written front to back by the kernel just before this one, its
    // tail is what the memory-side cache (256 MB) still holds
    const int chunk = (MODE == 1 && reverse) ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;
    const int64_t r0 = (int64_t)chunk * chunk_rows;
    const int64_t r1 = min(r0 + chunk_rows, M);
    float a[V], q[V], sc[V], sh[V], mu[V], is[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        a[v] = 0.0f;
        q[v] = 0.0f;
        if (MODE == 0) mu[v] = (c + v < C && r0 < M) ? Y[r0 * C + c + v] : 0.0f;  // the shift K
        if (MODE == 1 && c + v < C) {
            sc[v] = scale[c + v];
            sh[v] = shift[c + v];
            mu[v] = mean[c + v];
            is[v] = invstd[c + v];
        }
    }
    if (c < C) {
        constexpr int U = 4;  // rows in flight per thread
        for (int64_t r = r0 + rl; r < r1; r += (int64_t)rowlanes * U) {
            float y[U][V], d[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t rr = r + (int64_t)u * rowlanes;
                if (rr < r1) {
                    if (V == 4) {
                        *reinterpret_cast<float4 *>(y[u]) = *reinterpret_cast<const float4 *>(Y + rr * C + c);
                        if (MODE == 1) *reinterpret_cast<float4 *>(d[u]) = *reinterpret_cast<const float4 *>(dA + rr * C + c);
                    } else {
                        y[u][0] = Y[rr * C + c];
                        if (MODE == 1) d[u][0] = dA[rr * C + c];
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        y[u][v] = mu[v];  // contributes exactly zero (MODE 0: y - K = 0; MODE 1: dA = 0)
                        d[u][v] = 0.0f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    if (MODE == 0) {
                        const float dd = y[u][v] - mu[v];
                        a[v] += dd;
                        q[v] += dd * dd;
                    } else {
                        const float z = (y[u][v] - mu[v]) * sc[v] + sh[v];
                        const float dz = d[u][v] * (z > 0.0f ? 1.0f : slope);
                        a[v] += dz;
                        q[v] += dz * ((y[u][v] - mu[v]) * is[v]);
                    }
                }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        s1[(rl * colthreads + ct) * V + v] = a[v];
        s2[(rl * colthreads + ct) * V + v] = q[v];
    }
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (c + v < C) {
                float sa = 0.0f, sq = 0.0f;
                for (int k = 0; k < rowlanes; ++k) {  // fixed order: reproducible
                    sa += s1[(k * colthreads + ct) * V + v];
                    sq += s2[(k * colthreads + ct) * V + v];
                }
                if (MODE == 0) {
                    partial[((size_t)chunk * 4 + 0) * C + c + v] = sa;
                    partial[((size_t)chunk * 4 + 1) * C + c + v] = sq;
                    partial[((size_t)chunk * 4 + 2) * C + c + v] = mu[v];
                    partial[((size_t)chunk * 4 + 3) * C + c + v] = (float)(r1 - r0);
                } else {
                    partial[((size_t)chunk * 2 + 0) * C + c + v] = sa;
                    partial[((size_t)chunk * 2 + 1) * C + c + v] = sq;
                }
            }
        }
    }
}

// sums the chunk partials of one channel: one workgroup per channel, fixed tree => reproducible; double accum.
__device__ __forceinline__ void sum_partials(const float *__restrict__ partial, int chunks, int C, int c, double &a,
                                             double &q)
{
    __shared__ double r1[RW_BLOCK], r2[RW_BLOCK];
    double la = 0.0, lq = 0.0;
    for (int k = threadIdx.x; k < chunks; k += RW_BLOCK) {
        la += (double)partial[((size_t)k * 2 + 0) * C + c];
        lq += (double)partial[((size_t)k * 2 + 1) * C + c];
    }
    r1[threadIdx.x] = la;
    r2[threadIdx.x] = lq;
    __syncthreads();
    for (int off = RW_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            r1[threadIdx.x] += r1[threadIdx.x + off];
            r2[threadIdx.x] += r2[threadIdx.x + off];
        }
        __syncthreads();
    }
    a = r1[0];
    q = r2[0];
}

// finalize (training): the chunks' (sum d, sum d^2, shift K, rows n) are merged like Chan et al.'s parallel variance.
// Many chunks (a rows GEMM that fills the persistent grid leaves 2048 of them, 4 MB at 128 channels): one workgroup
// per channel would gather its column 4 bytes at a time from rows 4*C*4 bytes apart.  First pass instead: workgroup
// (column tile of 32, slice s) folds the chunks [s*per, (s+1)*per) of its 32 columns -- 128-byte rows, coalesced -- in
// double, ascending order per thread then a fixed tree, and writes the result back IN PLACE as the slice's first chunk
// (K = float(mean), S = n (mean - K), Q = M2 + S^2/n, n): the same shifted-sum format, so bn_finalize_kernel folds the slices with a stride.
constexpr int BM_COLS = 32, BM_LANES = RW_BLOCK / BM_COLS;
__global__ __launch_bounds__(RW_BLOCK) void bn_merge_slices_kernel(float *__restrict__ partial, int chunks, int per, int C)
{
    __shared__ double rn[BM_LANES][BM_COLS], rm[BM_LANES][BM_COLS], r2[BM_LANES][BM_COLS];
    const int cx = threadIdx.x % BM_COLS, ry = threadIdx.x / BM_COLS;
    const int c = blockIdx.x * BM_COLS + cx;
    const int k0 = blockIdx.y * per, k1 = min(k0 + per, chunks);
    double n = 0.0, mu = 0.0, m2 = 0.0;
    if (c < C)
        for (int k = k0 + ry; k < k1; k += BM_LANES) {
            const float *pk = partial + (size_t)k * 4 * C + c;
            const double nb = (double)pk[(size_t)3 * C];
            if (nb > 0.0) {
                const double S = (double)pk[0], Q = (double)pk[(size_t)C];
                const double mb = (double)pk[(size_t)2 * C] + S / nb, m2b = Q - S * S / nb;
                const double tot = n + nb, dm = mb - mu;
                mu += dm * (nb / tot);
                m2 += m2b + dm * dm * (n * nb / tot);
                n = tot;
            }
        }
    rn[ry][cx] = n;
    rm[ry][cx] = mu;
    r2[ry][cx] = m2;
    __syncthreads();
    for (int off = BM_LANES / 2; off > 0; off >>= 1) {
        if (ry < off) {
            const double na = rn[ry][cx], nb = rn[ry + off][cx];
            const double tot = na + nb;
            if (nb > 0.0) {
                const double dm = rm[ry + off][cx] - rm[ry][cx];
                rm[ry][cx] += dm * (nb / tot);
                r2[ry][cx] += r2[ry + off][cx] + dm * dm * (na * nb / tot);
                rn[ry][cx] = tot;
            }
        }
        __syncthreads();
    }
    if (ry == 0 && c < C && k0 < k1) {
        float *pk = partial + (size_t)k0 * 4 * C + c;
        const double nn = rn[0][cx];
        const float kf = (float)rm[0][cx];
        const double sres = nn * (rm[0][cx] - (double)kf);  // what the float shift misses of the slice mean
        pk[0] = (float)sres;
        pk[(size_t)C] = (float)(r2[0][cx] + (nn > 0.0 ? sres * sres / nn : 0.0));
        pk[(size_t)2 * C] = kf;
        pk[(size_t)3 * C] = (float)nn;
    }
}

// A chunk is (n, mean = K + S/n, M2 = Q - S^2/n); two sets merge as
//   n = na + nb,  mean = ma + (mb - ma) nb / n,  M2 = M2a + M2b + (mb - ma)^2 na nb / n
// in double: every thread folds its chunks in ascending order, then a fixed tree over the workgroup (reproducible,
// ONE pass over the partials).  mean, biased var -> scale = gamma*invstd, shift_out = beta; running stats update.
// eval: scale from the running statistics.  One workgroup per channel.
__global__ __launch_bounds__(RW_BLOCK) void bn_finalize_kernel(
    const float *__restrict__ partial, int chunks, int chunk_stride, int64_t M, int C, float eps, float momentum,
    const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ running_mean,
    float *__restrict__ running_var, int64_t *__restrict__ batches, int training, float *__restrict__ mean_out,
    float *__restrict__ invstd_out, float *__restrict__ scale_out, float *__restrict__ shift_out)
{
    __shared__ double rn[RW_BLOCK], rm[RW_BLOCK], r2[RW_BLOCK];
    const int c = blockIdx.x;
    float mean, var;
    if (training) {
        if (batches && c == 0 && threadIdx.x == 0) *batches += 1;  // BatchNorm's num_batches_tracked
        double n = 0.0, mu = 0.0, m2 = 0.0;
        for (int k = threadIdx.x; k < chunks; k += RW_BLOCK) {
            const float *pk = partial + (size_t)k * chunk_stride * 4 * C + c;
            const double nb = (double)pk[(size_t)3 * C];
            if (nb > 0.0) {
                const double S = (double)pk[0], Q = (double)pk[(size_t)C];
                const double mb = (double)pk[(size_t)2 * C] + S / nb, m2b = Q - S * S / nb;
                const double tot = n + nb, dm = mb - mu;
                mu += dm * (nb / tot);
                m2 += m2b + dm * dm * (n * nb / tot);
                n = tot;
            }
        }
        rn[threadIdx.x] = n;
        rm[threadIdx.x] = mu;
        r2[threadIdx.x] = m2;
        __syncthreads();
        for (int off = RW_BLOCK / 2; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) {
                const double na = rn[threadIdx.x], nb = rn[threadIdx.x + off];
                const double tot = na + nb;
                if (nb > 0.0) {
                    const double dm = rm[threadIdx.x + off] - rm[threadIdx.x];
                    rm[threadIdx.x] += dm * (nb / tot);
                    r2[threadIdx.x] += r2[threadIdx.x + off] + dm * dm * (na * nb / tot);
                    rn[threadIdx.x] = tot;
                }
            }
            __syncthreads();
        }
        double v = r2[0] / (double)M;
        if (v < 0.0) v = 0.0;
        mean = (float)rm[0];
        var = (float)v;
        if (running_mean && threadIdx.x == 0) {
            const double unb = M > 1 ? v * ((double)M / (double)(M - 1)) : v;
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
        }
    } else {
        mean = running_mean[c];
        var = running_var[c];
    }
    if (threadIdx.x != 0) return;
    const float invstd = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.0f;
    mean_out[c] = mean;
    invstd_out[c] = invstd;
    scale_out[c] = g * invstd;
    shift_out[c] = beta ? beta[c] : 0.0f;  // the kernels apply (y - mean) * scale + beta: no folded shift, whose
                                            // product mean * scale cancels against y * scale when |mean| >> std
}

// Grid-stride walk over a (rows, C) matrix in units of V floats that tracks the channel without a division per
// element: the stride is a fixed number of elements, so the channel advances by (stride % C) modulo C.
struct ChanWalk {
    int c, step, C;
    __device__ ChanWalk(int64_t e0, int64_t stride, int C_) : C(C_)
    {
        c = (int)(e0 % C_);
        step = (int)(stride % C_);
    }
    __device__ __forceinline__ void next()
    {
        c += step;
        if (c >= C) c -= C;
    }
};

__device__ __forceinline__ float leaky(float z, float slope) { return z > 0.0f ? z : z * slope; }

// A = act((Y - mean)*scale + beta), act = LeakyReLU(slope) (slope = 1 -> identity)
template <int V>
__global__ __launch_bounds__(RW_BLOCK) void bn_act_kernel(const float *__restrict__ Y, const float *__restrict__ mean,
                                                           const float *__restrict__ scale,
                                                           const float *__restrict__ shift, float slope,
                                                           int64_t total, int C, float *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * RW_BLOCK * V;
    int64_t e = ((int64_t)blockIdx.x * RW_BLOCK + threadIdx.x) * V;
    ChanWalk cw(e, stride, C);
    for (; e < total; e += stride, cw.next()) {
        if (V == 4) {
            const float4 y = *reinterpret_cast<const float4 *>(Y + e);
            const float4 mu = *reinterpret_cast<const float4 *>(mean + cw.c);
            const float4 sc = *reinterpret_cast<const float4 *>(scale + cw.c);
            const float4 sh = *reinterpret_cast<const float4 *>(shift + cw.c);
            float4 o;
            o.x = leaky((y.x - mu.x) * sc.x + sh.x, slope);
            o.y = leaky((y.y - mu.y) * sc.y + sh.y, slope);
            o.z = leaky((y.z - mu.z) * sc.z + sh.z, slope);
            o.w = leaky((y.w - mu.w) * sc.w + sh.w, slope);
            *reinterpret_cast<float4 *>(out + e) = o;
        } else {
            out[e] = leaky((Y[e] - mean[cw.c]) * scale[cw.c] + shift[cw.c], slope);
        }
    }
}

// fused BN-affine + LeakyReLU + max over the ns consecutive rows of a group (first max wins, like max_pool2d)
__global__ __launch_bounds__(RW_BLOCK) void bn_act_maxpool_kernel(const float *__restrict__ Y,
                                                                   const float *__restrict__ mean,
                                                                   const float *__restrict__ scale,
                                                                   const float *__restrict__ shift, float slope,
                                                                   int64_t G, int ns, int C,
                                                                   float *__restrict__ out, int *__restrict__ arg)
{
    const int64_t total = G * C;
    for (int64_t e = (int64_t)blockIdx.x * RW_BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * RW_BLOCK) {
        const int64_t g = e / C;
        const int c = (int)(e - g * C);
        const float mu = mean[c], sc = scale[c], sh = shift[c];
        const float *y = Y + (size_t)g * ns * C + c;
        float best = -INFINITY;
        int bs = 0;
        int s = 0;
        // a NaN wins and stays (max_pool2d's rule: the first NaN of the window is the result); tested on the bit pattern
        // because the file is compiled with -fno-honor-nans, which lets the compiler drop floating-point NaN tests
        auto take = [&](float a, int row) {
            const bool is_nan = (__float_as_uint(a) & 0x7fffffffu) > 0x7f800000u;
            const bool have_nan = (__float_as_uint(best) & 0x7fffffffu) > 0x7f800000u;
            if (!have_nan && (is_nan || a > best)) {
                best = a;
                bs = row;
            }
        };
        for (; s + 8 <= ns; s += 8) {  // eight independent row loads in flight, compared in row order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = y[(size_t)(s + u) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float z = (v[u] - mu) * sc + sh;
                take(z > 0.0f ? z : z * slope, s + u);
            }
        }
        for (; s < ns; ++s) {
            const float z = (y[(size_t)s * C] - mu) * sc + sh;
            take(z > 0.0f ? z : z * slope, s);
        }
        out[e] = best;
        arg[e] = bs;
    }
}

// -------------------------------------------------------------------------------------------------
// backward of  A = act(BN_train(Y)):  dZ = dA * act'(z);  dbeta = sum dZ;  dgamma = sum dZ * yhat
//   pass 1: per-chunk partial sums of dZ and dZ*yhat (colreduce_partial_kernel<V, 1> above; pooled form below)
// same reduction when dA is the gradient of the POOLED output: only the arg-max row of each group is non-zero
__global__ __launch_bounds__(RW_BLOCK) void bn_pool_bwd_partial_kernel(const float *__restrict__ dP,
                                                                        const int *__restrict__ arg,
                                                                        const float *__restrict__ Y,
                                                                        const float *__restrict__ scale,
                                                                        const float *__restrict__ shift,
                                                                        const float *__restrict__ mean,
                                                                        const float *__restrict__ invstd, float slope,
                                                                        int64_t G, int ns, int C, int chunk_rows,
                                                                        float *__restrict__ partial)
{
    __shared__ float s1[4][64], s2[4][64];
    const int tc = threadIdx.x & 63, tr = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tc;
    const int64_t g0 = (int64_t)blockIdx.y * chunk_rows;
    const int64_t g1 = min(g0 + chunk_rows, G);
    float a = 0.0f, q = 0.0f;
    if (c < C) {
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        for (int64_t g = g0 + tr; g < g1; g += 4) {
            const int s = arg[g * C + c];
            const float y = Y[((size_t)g * ns + s) * C + c];
            const float z = (y - mu) * sc + sh;
            const float dz = dP[g * C + c] * (z > 0.0f ? 1.0f : slope);
            a += dz;
            q += dz * ((y - mu) * is);
        }
    }
    s1[tr][tc] = a;
    s2[tr][tc] = q;
    __syncthreads();
    if (tr == 0 && c < C) {
        a = (s1[0][tc] + s1[1][tc]) + (s1[2][tc] + s1[3][tc]);
        q = (s2[0][tc] + s2[1][tc]) + (s2[2][tc] + s2[3][tc]);
        partial[((size_t)blockIdx.y * 2 + 0) * C + c] = a;
        partial[((size_t)blockIdx.y * 2 + 1) * C + c] = q;
    }
}

// sums the chunk partials: dbeta[c], dgamma[c]  (one workgroup per channel); optionally also the two per-channel terms
// the fused GEMM prologues subtract: c1 = dbeta / M, c2 = invstd * dgamma / M (zero when BatchNorm ran on running stats)
__global__ __launch_bounds__(RW_BLOCK) void bn_bwd_finalize_kernel(const float *__restrict__ partial, int chunks,
                                                                    int C, float *__restrict__ dbeta,
                                                                    float *__restrict__ dgamma,
                                                                    const float *__restrict__ invstd, int64_t M,
                                                                    int training, float *__restrict__ c1,
                                                                    float *__restrict__ c2)
{
    double a, q;
    sum_partials(partial, chunks, C, blockIdx.x, a, q);
    if (threadIdx.x == 0) {
        dbeta[blockIdx.x] = (float)a;
        dgamma[blockIdx.x] = (float)q;
        if (c1) {
            const float invM = 1.0f / (float)M;
            c1[blockIdx.x] = training ? (float)a * invM : 0.0f;
            c2[blockIdx.x] = training ? invstd[blockIdx.x] * ((float)q * invM) : 0.0f;
        }
    }
}

//   pass 2: dY = scale * (dZ - dbeta/M - yhat * dgamma/M)     (training);   dY = scale * dZ   (eval)
__device__ __forceinline__ float bn_bwd_elem(float da, float y, float sc, float sh, float mu, float is, float db,
                                             float dg, float slope, float invM, int training)
{
    const float z = (y - mu) * sc + sh;
    const float dz = da * (z > 0.0f ? 1.0f : slope);
    float v = dz;
    if (training) v = dz - db * invM - ((y - mu) * is) * (dg * invM);
    return sc * v;
}

template <int V>
__global__ __launch_bounds__(RW_BLOCK) void bn_act_bwd_apply_kernel(
    const float *__restrict__ dA, const float *__restrict__ Y, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ dbeta, const float *__restrict__ dgamma, float slope, int64_t M, int C, int training,
    float *__restrict__ dY)
{
    const int64_t total = M * C;
    const float invM = 1.0f / (float)M;
    const int64_t stride = (int64_t)gridDim.x * RW_BLOCK * V;
    int64_t e = ((int64_t)blockIdx.x * RW_BLOCK + threadIdx.x) * V;
    ChanWalk cw(e, stride, C);
    if (cw.step == 0 && e < total) {
        // the sweep stride is a multiple of C (every power-of-two width): this thread stays on its V channels, so the six
        // per-channel constants live in registers instead of being fetched again for every float4 (same arithmetic)
        float sc[V], sh[V], mu[V], is[V], k1[V], k2[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = cw.c + v;
            sc[v] = scale[c], sh[v] = shift[c], mu[v] = mean[c], is[v] = invstd[c];
            k1[v] = dbeta[c] * invM, k2[v] = dgamma[c] * invM;
        }
        for (; e < total; e += stride) {
            float y[V], d[V], o[V];
            if (V == 4) {
                *reinterpret_cast<float4 *>(y) = *reinterpret_cast<const float4 *>(Y + e);
                *reinterpret_cast<float4 *>(d) = *reinterpret_cast<const float4 *>(dA + e);
            } else {
                y[0] = Y[e];
                d[0] = dA[e];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float z = (y[v] - mu[v]) * sc[v] + sh[v];
                const float dz = d[v] * (z > 0.0f ? 1.0f : slope);
                float t = dz;
                if (training) t = dz - k1[v] - ((y[v] - mu[v]) * is[v]) * k2[v];
                o[v] = sc[v] * t;
            }
            if (V == 4) *reinterpret_cast<float4 *>(dY + e) = *reinterpret_cast<float4 *>(o);
            else dY[e] = o[0];
        }
        return;
    }
    for (; e < total; e += stride, cw.next()) {
        float y[V], d[V], o[V];
        if (V == 4) {
            *reinterpret_cast<float4 *>(y) = *reinterpret_cast<const float4 *>(Y + e);
            *reinterpret_cast<float4 *>(d) = *reinterpret_cast<const float4 *>(dA + e);
        } else {
            y[0] = Y[e];
            d[0] = dA[e];
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int c = cw.c + v;
            o[v] = bn_bwd_elem(d[v], y[v], scale[c], shift[c], mean[c], invstd[c], dbeta[c], dgamma[c], slope, invM,
                               training);
        }
        if (V == 4) *reinterpret_cast<float4 *>(dY + e) = *reinterpret_cast<float4 *>(o);
        else dY[e] = o[0];
    }
}

// pooled variant: dP (G, C) reaches only the arg-max row of each group; one lane per (group, channel) walks ns rows
// (few groups: the ns rows are cut into `split` segments with a lane each, so that the launch still fills the chip)
__global__ __launch_bounds__(RW_BLOCK) void bn_pool_bwd_apply_kernel(
    const float *__restrict__ dP, const int *__restrict__ arg, const float *__restrict__ Y,
    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ dbeta, const float *__restrict__ dgamma, float slope,
    int64_t G, int ns, int C, int split, int training, float *__restrict__ dY)
{
    const int64_t GC = G * C, total = GC * split;
    const int seg_len = (ns + split - 1) / split;
    const float invM = 1.0f / (float)(G * ns);
    for (int64_t w = (int64_t)blockIdx.x * RW_BLOCK + threadIdx.x; w < total; w += (int64_t)gridDim.x * RW_BLOCK) {
        const int seg = (int)(w / GC);
        const int64_t e = w - (int64_t)seg * GC;
        const int64_t g = e / C;
        const int c = (int)(e - g * C);
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c], db = dbeta[c], dg = dgamma[c];
        const float dp = dP[e];
        const int sa = arg[e];
        const size_t base = (size_t)g * ns * C + c;
        const int s_end = min(ns, (seg + 1) * seg_len);
        int s = seg * seg_len;
        for (; s + 8 <= s_end; s += 8) {  // eight independent row loads in flight
            float y[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) y[u] = Y[base + (size_t)(s + u) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                dY[base + (size_t)(s + u) * C] =
                    bn_bwd_elem(s + u == sa ? dp : 0.0f, y[u], sc, sh, mu, is, db, dg, slope, invM, training);
        }
        for (; s < s_end; ++s) {
            const float y = Y[base + (size_t)s * C];
            dY[base + (size_t)s * C] = bn_bwd_elem(s == sa ? dp : 0.0f, y, sc, sh, mu, is, db, dg, slope, invM, training);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// three_nn weights + interpolation + skip concat (DenseFPModule.conv + BaseDenseConvolutionUp.forward):
//   out[(b,i), :] = [ (w0*f0 + w1*f1) + w2*f2 , skip_cl[b,i,:] ],  f_t = feat_cl[b, idx[b,i,t], :]
__global__ __launch_bounds__(RW_BLOCK) void interp_concat_fwd_kernel(const float *__restrict__ feat_cl,
                                                                      const int64_t *__restrict__ idx,
                                                                      const float *__restrict__ w,
                                                                      const float *__restrict__ skip_cl, int m, int n,
                                                                      int C1, int C2, int ld, int64_t total,
                                                                      float *__restrict__ out)
{
    const int Cw = C1 + C2;  // columns [Cw, ld) are zero padding
    for (int64_t e = (int64_t)blockIdx.x * RW_BLOCK + threadIdx.x; e < total; e += (int64_t)gridDim.x * RW_BLOCK) {
        const int64_t row = e / ld;  // b*n + i
        const int c = (int)(e - row * ld);
        if (c >= Cw) {
            out[e] = 0.0f;
            continue;
        }
        float v;
        if (c < C1) {
            const int b = (int)(row / n);
            const int64_t *ip = idx + row * 3;
            const float *wp = w + row * 3;
            const int k0 = min(max((int)ip[0], 0), m - 1), k1 = min(max((int)ip[1], 0), m - 1),
                      k2 = min(max((int)ip[2], 0), m - 1);
            const float *fb = feat_cl + (size_t)b * m * C1 + c;
            const float a0 = wp[0] * fb[(size_t)k0 * C1];
            const float a1 = wp[1] * fb[(size_t)k1 * C1];
            const float a2 = wp[2] * fb[(size_t)k2 * C1];
            v = (a0 + a1) + a2;
        } else {
            v = skip_cl[row * C2 + (c - C1)];
        }
        out[e] = v;
    }
}

// same, four output columns per thread (ld % 4 == 0, C1 % 4 == 0): one row decode, one index/weight fetch and three
// 16-byte feature loads per float4 stored -- the one-column kernel above spent its time on per-element address
// arithmetic (216 us for the 277 MB decoder tensor); per-element arithmetic and its order are unchanged
__global__ __launch_bounds__(RW_BLOCK) void interp_concat_fwd4_kernel(const float *__restrict__ feat_cl,
                                                                       const int64_t *__restrict__ idx,
                                                                       const float *__restrict__ w,
                                                                       const float *__restrict__ skip_cl, int m, int n,
                                                                       int C1, int C2, int ld, int64_t total4,
                                                                       float *__restrict__ out)
{
    const int Cw = C1 + C2, q = ld >> 2;
    for (int64_t e4 = (int64_t)blockIdx.x * RW_BLOCK + threadIdx.x; e4 < total4; e4 += (int64_t)gridDim.x * RW_BLOCK) {
        const int64_t row = e4 / q;  // b*n + i
        const int c = (int)(e4 - row * q) * 4;
        float4 v;
        if (c < C1) {  // C1 % 4 == 0: the four columns are all interpolated ones
            const int b = (int)(row / n);
            const int64_t *ip = idx + row * 3;
            const float *wp = w + row * 3;
            const int k0 = min(max((int)ip[0], 0), m - 1), k1 = min(max((int)ip[1], 0), m - 1),
                      k2 = min(max((int)ip[2], 0), m - 1);
            const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
            const float *fb = feat_cl + (size_t)b * m * C1 + c;
            const float4 f0 = *reinterpret_cast<const float4 *>(fb + (size_t)k0 * C1);
            const float4 f1 = *reinterpret_cast<const float4 *>(fb + (size_t)k1 * C1);
            const float4 f2 = *reinterpret_cast<const float4 *>(fb + (size_t)k2 * C1);
            v.x = (w0 * f0.x + w1 * f1.x) + w2 * f2.x;
            v.y = (w0 * f0.y + w1 * f1.y) + w2 * f2.y;
            v.z = (w0 * f0.z + w1 * f1.z) + w2 * f2.z;
            v.w = (w0 * f0.w + w1 * f1.w) + w2 * f2.w;
        } else {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = (c + j < Cw) ? skip_cl[row * C2 + (c + j - C1)] : 0.0f;
            v = make_float4(t[0], t[1], t[2], t[3]);
        }
        *reinterpret_cast<float4 *>(out + e4 * 4) = v;
    }
}

// inverse-distance weights exactly as the reference builds them (core/base_conv/dense.py:137-139):
//   r_t = 1/(dist_t + 1e-8);  w_t = r_t / ((r0 + r1) + r2)
__global__ void idw_weights_kernel(const float *__restrict__ dist, int64_t rows, float *__restrict__ w)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float r0 = 1.0f / (dist[r * 3 + 0] + 1e-8f);
    const float r1 = 1.0f / (dist[r * 3 + 1] + 1e-8f);
    const float r2 = 1.0f / (dist[r * 3 + 2] + 1e-8f);
    const float norm = (r0 + r1) + r2;
    w[r * 3 + 0] = r0 / norm;
    w[r * 3 + 1] = r1 / norm;
    w[r * 3 + 2] = r2 / norm;
}

static inline unsigned grid_for(int64_t total)
{
    int64_t g = (total + RW_BLOCK - 1) / RW_BLOCK;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace tp3d

using namespace tp3d;

TP3D_EXPORT int tp3d_group_concat_fwd_f32(const float *pos, const float *new_pos, const float *x_cl,
                                          const int64_t *idx, int B, int N, int np, int ns, int C, int ld,
                                          float radius, int normalize, float *out, void *stream)
{
    if (B < 0 || N <= 0 || np < 0 || ns < 0 || C < 0 || ld < C + 3) return TP3D_E_BADARG;
    const int64_t total = (int64_t)B * np * ns * ld;
    if (total == 0) return TP3D_OK;
    if (!pos || !new_pos || !idx || !out || (C > 0 && !x_cl)) return TP3D_E_BADARG;
    if ((ld & 3) == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(group_concat_fwd4_kernel, dim3(grid_for(total / 4)), dim3(RW_BLOCK), 0, (hipStream_t)stream,
                           pos, new_pos, x_cl, idx, N, np, ns, C, ld, radius, normalize, total / 4, out);
    else
        hipLaunchKernelGGL(group_concat_fwd_kernel, dim3(grid_for(total)), dim3(RW_BLOCK), 0, (hipStream_t)stream, pos,
                           new_pos, x_cl, idx, N, np, ns, C, ld, radius, normalize, total, out);
    return check_launch();
}

namespace tp3d {
// order[j] (slot id) -> row id (slot / div); wsorted[j] = weight[slot]; only the first start[nbins] entries are real
__global__ void slots_to_rows_kernel(int *__restrict__ order, const float *__restrict__ weight, int div, int L,
                                     const int *__restrict__ start, int nbins, float *__restrict__ wsorted)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= L || j >= start[nbins]) return;
    const int slot = order[j];
    if (wsorted) wsorted[j] = weight[slot];
    order[j] = slot / div;
}
}  // namespace tp3d

namespace tp3d {
static bool scatter_goes_flat(int B, int L, int nbins)
{
    return L >= 2 * nbins && (int64_t)B * L < 0x7fffffff && (int64_t)B * nbins < 0x3fffffff &&
           ((B == 1 && L >= 16384) || !csr_fits_lds(L, nbins));
}
}  // namespace tp3d

// plan[0..8] = byte offsets of start, order, scratch, wsorted (-1 without weights), merge_tmp in the workspace, its
// size in bytes, 1 when the table is inverted flat over the whole batch, ints of `scratch` that path uses, byte offset
// of the hub list (1 + B*nbins ints)
TP3D_EXPORT int tp3d_scatter_plan(int B, int L, int nbins, int with_weights, int64_t *plan)
{
    if (B <= 0 || L <= 0 || nbins <= 0 || !plan) return TP3D_E_BADARG;
    const ScatterWorkspace w = carve_scatter_workspace(nullptr, B, L, nbins, with_weights != 0);
    plan[0] = (char *)w.start - (char *)nullptr;
    plan[1] = (char *)w.order - (char *)nullptr;
    plan[2] = (char *)w.scratch - (char *)nullptr;
    plan[3] = w.wsorted ? (char *)w.wsorted - (char *)nullptr : -1;
    plan[4] = (char *)w.merge_tmp - (char *)nullptr;
    plan[5] = (int64_t)w.bytes;
    plan[6] = scatter_goes_flat(B, L, nbins) ? 1 : 0;
    plan[7] = plan[6] ? 2 * (int64_t)B * nbins : 0;  // invert_table: histogram + cursors
    plan[8] = (char *)w.hubs - (char *)nullptr;
    return TP3D_OK;
}

// The inverted neighbour table ("which slots point at support point k") depends on idx / weight only -- geometry, not
// features: tp3d_rows_scatter_invert builds it into `workspace`, tp3d_rows_scatter_apply_f32 consumes a table built
// earlier for the same (idx, weight, B, L, div, nbins) -- e.g. one step ahead on another stream, beside the sampling
// and the searches -- and tp3d_rows_scatter_bwd_f32 does both.
TP3D_EXPORT int tp3d_rows_scatter_invert(const int64_t *idx, const float *weight, int B, int L, int div, int nbins,
                                         void *workspace, size_t workspace_bytes, void *stream)
{
    if (B < 0 || L < 0 || div <= 0 || nbins <= 0) return TP3D_E_BADARG;
    if (B == 0 || L == 0) return TP3D_OK;
    if (!idx || !workspace || B > 65535) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ScatterWorkspace w = carve_scatter_workspace(workspace, B, L, nbins, weight != nullptr);
    if (workspace_bytes < w.bytes) return TP3D_E_BADARG;
    // One large cloud (partial-dense decoders), or clouds whose tables do not fit one workgroup's LDS (multi-scale
    // grouping: 512 x 128 slots per cloud): invert ONE flat table over the whole device instead of one table per
    // workgroup (scratch holds the histogram and the cursors), then turn slot ids into row ids and line the weights up.
    // (measured on the 49 152-slot decoder tables, which fit LDS: flat 523 us vs per-cloud 354 us, so off by default)
    const bool flat = scatter_goes_flat(B, L, nbins);
    if (flat) {
        const int64_t slots = (int64_t)B * L, bins = (int64_t)B * nbins;
        if (int rc = invert_table(idx, slots, bins, w.scratch, w.start, w.scratch + bins, w.order, s, L, nbins, w.merge_tmp))
            return rc;
        hipLaunchKernelGGL(slots_to_rows_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, w.order, weight, div,
                           (int)slots, w.start, (int)bins, w.wsorted);
        if (int rc = check_launch()) return rc;
    } else if (int rc = csr_transpose(idx, B, L, nbins, div, weight, w.start, w.order, w.wsorted, w.scratch, s)) {
        return rc;
    }
    // the destinations whose runs are long enough to be summed by a whole workgroup (rows_gather_hub_kernel)
    if (int rc = zero_async(w.hubs, sizeof(int), s)) return rc;
    hipLaunchKernelGGL(find_hubs_kernel, dim3((unsigned)(((int64_t)B * nbins + 255) / 256)), dim3(256), 0, s, w.start, B, nbins,
                       flat ? 1 : 0, w.hubs);
    return check_launch();
}

TP3D_EXPORT int tp3d_rows_scatter_apply_f32(const float *grad_rows, int B, int L, int div, int nbins, int ld, int col0,
                                            int C, int with_weights, float *grad_x_cl, void *table, size_t table_bytes,
                                            void *stream)
{
    // grad_rows: (B, L/div rows, ld); slot l of cloud b refers to row l/div; destinations: (B, nbins, C)
    if (B < 0 || L < 0 || div <= 0 || nbins <= 0 || ld <= 0 || col0 < 0 || C < 0 || col0 + C > ld) return TP3D_E_BADARG;
    if (B == 0 || C == 0) return TP3D_OK;
    if (!grad_x_cl) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (L == 0) return zero_async(grad_x_cl, (size_t)B * nbins * C * sizeof(float), s);
    if (!grad_rows || !table || B > 65535) return TP3D_E_BADARG;
    ScatterWorkspace w = carve_scatter_workspace(table, B, L, nbins, with_weights != 0);
    if (table_bytes < w.bytes) return TP3D_E_BADARG;
    const bool flat = scatter_goes_flat(B, L, nbins);
    dim3 grid((nbins + RW_BLOCK / 64 - 1) / (RW_BLOCK / 64), B);
    const dim3 hub_grid(256);  // the hub count lives on the device: a fixed grid strides over the list (usually a few dozen)
    const int fl = flat ? 1 : 0;
    if (C > 128) {
        hipLaunchKernelGGL(rows_gather_sum_kernel<4>, grid, dim3(RW_BLOCK), 0, s, grad_rows, w.start, w.order, w.wsorted,
                           nbins, L, L / div, ld, col0, C, grad_x_cl, fl, HUB_MIN);
        hipLaunchKernelGGL(rows_gather_hub_kernel<4>, hub_grid, dim3(HUB_BLOCK), 0, s, grad_rows, w.start, w.order, w.wsorted,
                           w.hubs, nbins, L, L / div, ld, col0, C, grad_x_cl, fl);
    } else if (C > 64) {
        hipLaunchKernelGGL(rows_gather_sum_kernel<2>, grid, dim3(RW_BLOCK), 0, s, grad_rows, w.start, w.order, w.wsorted,
                           nbins, L, L / div, ld, col0, C, grad_x_cl, fl, HUB_MIN);
        hipLaunchKernelGGL(rows_gather_hub_kernel<2>, hub_grid, dim3(HUB_BLOCK), 0, s, grad_rows, w.start, w.order, w.wsorted,
                           w.hubs, nbins, L, L / div, ld, col0, C, grad_x_cl, fl);
    } else {
        hipLaunchKernelGGL(rows_gather_sum_kernel<1>, grid, dim3(RW_BLOCK), 0, s, grad_rows, w.start, w.order, w.wsorted,
                           nbins, L, L / div, ld, col0, C, grad_x_cl, fl, HUB_MIN);
        hipLaunchKernelGGL(rows_gather_hub_kernel<1>, hub_grid, dim3(HUB_BLOCK), 0, s, grad_rows, w.start, w.order, w.wsorted,
                           w.hubs, nbins, L, L / div, ld, col0, C, grad_x_cl, fl);
    }
    return check_launch();
}

TP3D_EXPORT int tp3d_rows_scatter_bwd_f32(const float *grad_rows, const int64_t *idx, const float *weight, int B,
                                          int L, int div, int nbins, int ld, int col0, int C, float *grad_x_cl,
                                          void *workspace, size_t workspace_bytes, void *stream)
{
    if (B < 0 || L < 0 || div <= 0 || nbins <= 0 || ld <= 0 || col0 < 0 || C < 0 || col0 + C > ld) return TP3D_E_BADARG;
    if (B == 0 || C == 0) return TP3D_OK;
    if (!grad_x_cl) return TP3D_E_BADARG;
    if (L == 0) return zero_async(grad_x_cl, (size_t)B * nbins * C * sizeof(float), (hipStream_t)stream);
    if (!grad_rows || !idx || !workspace || B > 65535) return TP3D_E_BADARG;
    if (int rc = tp3d_rows_scatter_invert(idx, weight, B, L, div, nbins, workspace, workspace_bytes, stream)) return rc;
    return tp3d_rows_scatter_apply_f32(grad_rows, B, L, div, nbins, ld, col0, C, weight != nullptr, grad_x_cl, workspace,
                                       workspace_bytes, stream);
}

TP3D_EXPORT size_t tp3d_bn_workspace_floats(int64_t M, int C)
{
    if (M < 0 || C < 0) return 0;
    // covers the reduction over M rows and over any M / ns pooled groups (which may fall in the small-matrix regime)
    int64_t chunks = M >= ST_SMALL_BELOW ? (M + ST_ROWS - 1) / ST_ROWS : (M + ST_ROWS_SMALL - 1) / ST_ROWS_SMALL;
    if (M >= ST_SMALL_BELOW && chunks < ST_SMALL_BELOW / ST_ROWS_SMALL) chunks = ST_SMALL_BELOW / ST_ROWS_SMALL;
    return (size_t)chunks * 4 * (size_t)C;  // forward statistics keep 4 rows per chunk, the backward reductions 2
}

// plan[0..2] = rows per chunk, chunks, workspace floats written, for the reduction tp3d_bn_stats_f32 (pooled_ns = 0)
// or tp3d_bn_act_bwd_f32 (pooled_ns = ns of its argmax form, else 1) runs over an (M, C) matrix
TP3D_EXPORT int tp3d_bn_plan(int64_t M, int C, int pooled_ns, int64_t *plan)
{
    if (M <= 0 || C <= 0 || pooled_ns < 0 || !plan) return TP3D_E_BADARG;
    const int64_t R = pooled_ns > 1 ? M / pooled_ns : M;
    const int crow = stat_rows(R);
    plan[0] = crow;
    plan[1] = (R + crow - 1) / crow;
    plan[2] = plan[1] * (pooled_ns == 0 ? 4 : 2) * C;
    return TP3D_OK;
}

// Statistics from chunk partials: more than BN_DIRECT_CHUNKS of them are first folded into BN_SLICES slices (coalesced,
// in place), then one workgroup per channel folds those.
constexpr int BN_DIRECT_CHUNKS = 64, BN_SLICES = 32;
static int launch_bn_finalize(float *partial, int chunks, int64_t M, int C, float eps, float momentum, const float *gamma,
                              const float *beta, float *running_mean, float *running_var, int64_t *batches, int training,
                              float *mean, float *invstd, float *scale, float *shift, hipStream_t s)
{
    int n = chunks, stride = 1;
    if (training && chunks > BN_DIRECT_CHUNKS) {
        stride = (chunks + BN_SLICES - 1) / BN_SLICES;
        n = (chunks + stride - 1) / stride;
        hipLaunchKernelGGL(bn_merge_slices_kernel, dim3((C + BM_COLS - 1) / BM_COLS, n), dim3(RW_BLOCK), 0, s, partial, chunks,
                           stride, C);
        if (int rc = check_launch()) return rc;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(RW_BLOCK), 0, s, partial, n, stride, M, C, eps, momentum, gamma, beta,
                       running_mean, running_var, batches, training, mean, invstd, scale, shift);
    return check_launch();
}

TP3D_EXPORT int tp3d_bn_stats_f32(const float *Y, int64_t M, int C, float eps, float momentum, const float *gamma,
                                  const float *beta, float *running_mean, float *running_var,
                                  int64_t *num_batches_tracked, int training, float *mean, float *invstd, float *scale,
                                  float *shift, float *workspace, void *stream)
{
    if (M <= 0 || C <= 0 || !mean || !invstd || !scale || !shift) return TP3D_E_BADARG;
    if (!training && (!running_mean || !running_var)) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int crow = stat_rows(M);
    const int chunks = (int)((M + crow - 1) / crow);
    if (training) {
        if (!Y || !workspace) return TP3D_E_BADARG;
        if (chunks > 65535) return TP3D_E_TOOBIG;
        if ((C & 3) == 0) {
            const StatTile t = stat_tile(C, 4);
            hipLaunchKernelGGL((colreduce_partial_kernel<4, 0>), dim3(t.gridx, chunks), dim3(RW_BLOCK), 0, s, Y,
                               (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                               (const float *)nullptr, (const float *)nullptr, 0.0f, M, C, t.colthreads, crow, workspace);
        } else {
            const StatTile t = stat_tile(C, 1);
            hipLaunchKernelGGL((colreduce_partial_kernel<1, 0>), dim3(t.gridx, chunks), dim3(RW_BLOCK), 0, s, Y,
                               (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                               (const float *)nullptr, (const float *)nullptr, 0.0f, M, C, t.colthreads, crow, workspace);
        }
        if (int rc = check_launch()) return rc;
    }
    return launch_bn_finalize(workspace, chunks, M, C, eps, momentum, gamma, beta, running_mean, running_var,
                              num_batches_tracked, training, mean, invstd, scale, shift, s);
}

// training-mode statistics from per-chunk (sum, sum of squares) partials written by another kernel's epilogue
// (tp3d_gemm_rows_f32): partial[chunk][2][C]
TP3D_EXPORT int tp3d_bn_finalize_f32(float *partial, int chunks, int64_t M, int C, float eps, float momentum,
                                     const float *gamma, const float *beta, float *running_mean, float *running_var,
                                     int64_t *num_batches_tracked, float *mean, float *invstd, float *scale, float *shift,
                                     void *stream)
{
    if (M <= 0 || C <= 0 || chunks <= 0 || !partial || !mean || !invstd || !scale || !shift) return TP3D_E_BADARG;
    return launch_bn_finalize(partial, chunks, M, C, eps, momentum, gamma, beta, running_mean, running_var,
                              num_batches_tracked, 1, mean, invstd, scale, shift, (hipStream_t)stream);
}

TP3D_EXPORT int tp3d_bn_act_f32(const float *Y, const float *mean, const float *scale, const float *shift, float slope,
                                int64_t M, int C, float *out, void *stream)
{
    if (M < 0 || C <= 0) return TP3D_E_BADARG;
    if (M == 0) return TP3D_OK;
    if (!Y || !mean || !scale || !shift || !out) return TP3D_E_BADARG;
    const int64_t total = M * C;
    if ((C & 3) == 0)
        hipLaunchKernelGGL(bn_act_kernel<4>, dim3(grid_for(total / 4)), dim3(RW_BLOCK), 0, (hipStream_t)stream, Y, mean,
                           scale, shift, slope, total, C, out);
    else
        hipLaunchKernelGGL(bn_act_kernel<1>, dim3(grid_for(total)), dim3(RW_BLOCK), 0, (hipStream_t)stream, Y, mean,
                           scale, shift, slope, total, C, out);
    return check_launch();
}

TP3D_EXPORT int tp3d_bn_act_maxpool_f32(const float *Y, const float *mean, const float *scale, const float *shift,
                                        float slope, int64_t G, int ns, int C, float *out, int *argmax, void *stream)
{
    if (G < 0 || ns <= 0 || C <= 0) return TP3D_E_BADARG;
    if (G == 0) return TP3D_OK;
    if (!Y || !mean || !scale || !shift || !out || !argmax) return TP3D_E_BADARG;
    hipLaunchKernelGGL(bn_act_maxpool_kernel, dim3(grid_for(G * C)), dim3(RW_BLOCK), 0, (hipStream_t)stream, Y, mean, scale,
                       shift, slope, G, ns, C, out, argmax);
    return check_launch();
}

namespace tp3d {
int bn_bwd_finalize_launch(const float *partial, int chunks, int C, float *dbeta, float *dgamma, const float *invstd, int64_t M,
                           int training, float *c1, float *c2, hipStream_t s)
{
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(RW_BLOCK), 0, s, partial, chunks, C, dbeta, dgamma, invstd, M,
                       training, c1, c2);
    return check_launch();
}

// pass 1 of the BatchNorm + activation backward: partial sums, then dbeta / dgamma (and c1 / c2 when asked for)
static int bn_bwd_reduce(const float *dA, const int *argmax, const float *Y, const float *scale, const float *shift,
                         const float *mean, const float *invstd, float slope, int64_t M, int ns, int C, int training,
                         float *dbeta, float *dgamma, float *c1, float *c2, float *workspace, int reverse, hipStream_t s)
{
    const int64_t R = argmax ? M / ns : M;  // rows of the reduction domain
    const int crow = stat_rows(R);
    const int chunks = (int)((R + crow - 1) / crow);
    if (chunks > 65535) return TP3D_E_TOOBIG;
    if (argmax) {
        hipLaunchKernelGGL(bn_pool_bwd_partial_kernel, dim3((C + 63) / 64, chunks), dim3(RW_BLOCK), 0, s, dA, argmax, Y,
                           scale, shift, mean, invstd, slope, R, ns, C, crow, workspace);
    } else if ((C & 3) == 0) {
        const StatTile t = stat_tile(C, 4);
        hipLaunchKernelGGL((colreduce_partial_kernel<4, 1>), dim3(t.gridx, chunks), dim3(RW_BLOCK), 0, s, Y, dA, scale,
                           shift, mean, invstd, slope, M, C, t.colthreads, crow, workspace, reverse);
    } else {
        const StatTile t = stat_tile(C, 1);
        hipLaunchKernelGGL((colreduce_partial_kernel<1, 1>), dim3(t.gridx, chunks), dim3(RW_BLOCK), 0, s, Y, dA, scale,
                           shift, mean, invstd, slope, M, C, t.colthreads, crow, workspace);
    }
    if (int rc = check_launch()) return rc;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(RW_BLOCK), 0, s, workspace, chunks, C, dbeta, dgamma, invstd, M,
                       training, c1, c2);
    return check_launch();
}
}  // namespace tp3d

TP3D_EXPORT int tp3d_bn_bwd_reduce_f32(const float *dA, const int *argmax, const float *Y, const float *scale,
                                       const float *shift, const float *mean, const float *invstd, float slope,
                                       int64_t M, int ns, int C, int training, float *dbeta, float *dgamma, float *c1,
                                       float *c2, float *workspace, int reverse, void *stream)
{
    if (M <= 0 || C <= 0 || ns <= 0 || (argmax && M % ns)) return TP3D_E_BADARG;
    if (!dA || !Y || !scale || !shift || !mean || !invstd || !dbeta || !dgamma || !c1 || !c2 || !workspace)
        return TP3D_E_BADARG;
    return bn_bwd_reduce(dA, argmax, Y, scale, shift, mean, invstd, slope, M, ns, C, training, dbeta, dgamma, c1, c2,
                         workspace, reverse, (hipStream_t)stream);
}

TP3D_EXPORT int tp3d_bn_act_bwd_f32(const float *dA, const int *argmax, const float *Y, const float *scale,
                                    const float *shift, const float *mean, const float *invstd, float slope,
                                    int64_t M, int ns, int C, int training, float *dbeta, float *dgamma, float *dY,
                                    float *workspace, void *stream)
{
    // dA: (M, C) dense when argmax == NULL, else the pooled gradient (M/ns, C) with its arg-max rows
    if (M <= 0 || C <= 0 || ns <= 0 || (argmax && M % ns)) return TP3D_E_BADARG;
    if (!dA || !Y || !scale || !shift || !mean || !invstd || !dbeta || !dgamma || !dY || !workspace)
        return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t R = argmax ? M / ns : M;  // rows of the reduction domain
    if (int rc = bn_bwd_reduce(dA, argmax, Y, scale, shift, mean, invstd, slope, M, ns, C, training, dbeta, dgamma, nullptr,
                               nullptr, workspace, 0, s))
        return rc;
    if (argmax) {
        int split = (int)((262144 + R * C - 1) / (R * C));  // ~4 waves per SIMD worth of lanes
        split = split < 1 ? 1 : (split > ns ? ns : split);
        hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(grid_for(R * C * split)), dim3(RW_BLOCK), 0, s, dA, argmax, Y,
                           scale, shift, mean, invstd, dbeta, dgamma, slope, R, ns, C, split, training, dY);
    }
    else if ((C & 3) == 0)
        hipLaunchKernelGGL(bn_act_bwd_apply_kernel<4>, dim3(grid_for(M * C / 4)), dim3(RW_BLOCK), 0, s, dA, Y, scale,
                           shift, mean, invstd, dbeta, dgamma, slope, M, C, training, dY);
    else
        hipLaunchKernelGGL(bn_act_bwd_apply_kernel<1>, dim3(grid_for(M * C)), dim3(RW_BLOCK), 0, s, dA, Y, scale,
                           shift, mean, invstd, dbeta, dgamma, slope, M, C, training, dY);
    return check_launch();
}

TP3D_EXPORT int tp3d_interp_concat_fwd_f32(const float *feat_cl, const int64_t *idx, const float *weight,
                                           const float *skip_cl, int B, int m, int n, int C1, int C2, int ld,
                                           float *out, void *stream)
{
    if (B < 0 || m <= 0 || n < 0 || C1 < 0 || C2 < 0 || ld < C1 + C2) return TP3D_E_BADARG;
    const int64_t total = (int64_t)B * n * ld;
    if (total == 0) return TP3D_OK;
    if (!out || (C1 > 0 && (!feat_cl || !idx || !weight)) || (C2 > 0 && !skip_cl)) return TP3D_E_BADARG;
    if ((ld & 3) == 0 && (C1 & 3) == 0 && (((uintptr_t)feat_cl | (uintptr_t)out) & 15) == 0)
        hipLaunchKernelGGL(interp_concat_fwd4_kernel, dim3(grid_for(total / 4)), dim3(RW_BLOCK), 0, (hipStream_t)stream,
                           feat_cl, idx, weight, skip_cl, m, n, C1, C2, ld, total / 4, out);
    else
        hipLaunchKernelGGL(interp_concat_fwd_kernel, dim3(grid_for(total)), dim3(RW_BLOCK), 0, (hipStream_t)stream,
                           feat_cl, idx, weight, skip_cl, m, n, C1, C2, ld, total, out);
    return check_launch();
}

namespace tp3d {
// Geometric relation vector of Relation-Shape convolution (reference modules/RSConv/dense.py:86-101), one row per
// (centroid, neighbour) pair:  [ |d|, c_x c_y c_z, p_x p_y p_z, d_x d_y d_z, 0 .. ],  p = pos[b, idx], c = new_pos[b, j],
// d = p - c, |d| = sqrt((dx*dx + dy*dy) + dz*dz).  One thread per row, ld = 12: three 16-byte stores.
__global__ __launch_bounds__(RW_BLOCK) void relation_rows_kernel(const float *__restrict__ pos,
                                                                  const float *__restrict__ new_pos,
                                                                  const int64_t *__restrict__ idx, int N, int np, int ns,
                                                                  int ld, int64_t rows, float *__restrict__ out)
{
    for (int64_t row = (int64_t)blockIdx.x * RW_BLOCK + threadIdx.x; row < rows; row += (int64_t)gridDim.x * RW_BLOCK) {
        const int64_t bj = row / ns;
        const int b = (int)(bj / np);
        const int k = min(max((int)idx[row], 0), N - 1);
        const float *p = pos + ((size_t)b * N + k) * 3, *c = new_pos + bj * 3;
        const float px = p[0], py = p[1], pz = p[2], cx = c[0], cy = c[1], cz = c[2];
        const float dx = px - cx, dy = py - cy, dz = pz - cz;
        float v[12] = {sqrtf((dx * dx + dy * dy) + dz * dz), cx, cy, cz, px, py, pz, dx, dy, dz, 0.0f, 0.0f};
        float *o = out + row * ld;
        if (ld == 12 && ((uintptr_t)out & 15) == 0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) reinterpret_cast<float4 *>(o)[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        } else {
            for (int q = 0; q < ld; ++q) o[q] = q < 10 ? v[q] : 0.0f;
        }
    }
}
}  // namespace tp3d

TP3D_EXPORT int tp3d_relation_rows_f32(const float *pos, const float *new_pos, const int64_t *idx, int B, int N, int np,
                                       int ns, int ld, float *out, void *stream)
{
    if (B < 0 || N <= 0 || np < 0 || ns < 0 || ld < 10) return TP3D_E_BADARG;
    const int64_t rows = (int64_t)B * np * ns;
    if (rows == 0) return TP3D_OK;
    if (!pos || !new_pos || !idx || !out) return TP3D_E_BADARG;
    hipLaunchKernelGGL(relation_rows_kernel, dim3(grid_for(rows)), dim3(RW_BLOCK), 0, (hipStream_t)stream, pos, new_pos, idx,
                       N, np, ns, ld, rows, out);
    return check_launch();
}

TP3D_EXPORT int tp3d_idw_weights_f32(const float *dist, int64_t rows, float *weight, void *stream)
{
    if (rows < 0) return TP3D_E_BADARG;
    if (rows == 0) return TP3D_OK;
    if (!dist || !weight) return TP3D_E_BADARG;
    hipLaunchKernelGGL(idw_weights_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dist, rows, weight);
    return check_launch();
}
