// Uniform-grid radius search that keeps the reference's output order.
//
// ball_query's contract (reference core/spatial_ops/neighbour_finder.py:31-37,164) is
// "the first nsample hits in ascending index order" (sort=0) or "the nsample closest, closest first" (sort=1).
// A spatial grid visits points out of index order, so the search is split in two: (1) collect EVERY hit of the
// 27 cells around the query (cell edge >= 1.01 r), (2) rank the hits by index (or by (distance, index)) and emit
// ranks < nsample.  The membership test is the same fp32 expression as the brute-force kernel and the oracle, so
// the result is bit-identical -- only ~27/ncells of the cloud is tested (C2: 432 instead of 16384 tests/query).
//
// Build: one workgroup per cloud, all in LDS: bounding box -> cell histogram (LDS atomics) -> scan -> fill ->
// cell-ordered ids + a cell-ordered xyz copy for coalesced tests (order inside a cell is irrelevant: hits are ranked).
#include <cstdlib>

#include "grid.h"

namespace tp3d {

constexpr int GB_BLOCK = 1024;
constexpr int GQ_BLOCK = 256;     // 4 waves, one query per wave
constexpr int GQ_CAP = 640;       // hit slots per query (overflow -> exact in-order scan of the cloud); 5 KB of LDS per
                                  // wave = 32 resident waves per CU, the latency-bound kernel's only lever

// Cell edge and cell counts of one cloud from its bounding box (shared by both builders).
// cell > 0: at least 1.01 * cell (so +-1 cell covers a ball of that radius with margin for the fp32 rounding of the
// cell coordinate); cell <= 0: sized for `target` points per cell of the box volume.  Never more than G cells per axis.
__device__ __forceinline__ GridInfo make_grid_info(const float lo3[3], const float hi3[3], int L, float cell, float target,
                                                   int G)
{
    const float e0 = hi3[0] - lo3[0], e1 = hi3[1] - lo3[1], e2 = hi3[2] - lo3[2];
    const float ext = fmaxf(fmaxf(e0, e1), e2);
    float want = cell * 1.01f;
    if (!(cell > 0.0f)) {
        const float eps = ext * 1.0e-3f;
        const float vol = fmaxf(e0, eps) * fmaxf(e1, eps) * fmaxf(e2, eps);
        want = cbrtf(vol * target / (float)max(L, 1));
    }
    float cs = fmaxf(want, ext / (float)G * 1.0001f);
    if (!(cs > 0.0f)) cs = 1.0f;
    GridInfo gi;
    gi.minx = lo3[0];
    gi.miny = lo3[1];
    gi.minz = lo3[2];
    gi.inv_cs = 1.0f / cs;
    gi.gx = min(G, (int)floorf(e0 * gi.inv_cs) + 1);
    gi.gy = min(G, (int)floorf(e1 * gi.inv_cs) + 1);
    gi.gz = min(G, (int)floorf(e2 * gi.inv_cs) + 1);
    gi.pad = 0;
    return gi;
}

// Exclusive scan of a cell histogram in LDS by one GB_BLOCK workgroup (per-thread serial chunk + wave scan +
// cross-wave): cnt[k] and cs_out[k] become base + the start of bin k; cs_out[nused] = L (or `last` when given; not
// written when last < 0: a slab that is not the cloud's last one).  Ends without a barrier.
__device__ __forceinline__ void grid_scan_cells(int *cnt, int *cs_out, int nused, int L, int *s_scan, int base = 0,
                                                int last = 0)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (nused + GB_BLOCK - 1) / GB_BLOCK;
    const int k0 = min(tid * per, nused), k1 = min(k0 + per, nused);
    int sum = 0;
    for (int k = k0; k < k1; ++k) sum += cnt[k];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    int run = base + incl - sum;
#pragma unroll
    for (int w = 0; w < GB_BLOCK / 64; ++w) run += (w < wave) ? s_scan[w] : 0;  // the waves in front
    for (int k = k0; k < k1; ++k) {
        const int v = cnt[k];
        cnt[k] = run;
        cs_out[k] = run;  // start of bin k
        run += v;
    }
    if (tid == 0 && last >= 0) cs_out[nused] = last ? last : L;
}

// Bounding box of the PT points each thread holds -> GridInfo of the cloud in s_info / info[b] (ends with a barrier).
__device__ __forceinline__ void grid_box_to_info(float mn[3], float mx[3], int L, float radius, float target, int G,
                                                 float (*s_red)[GB_BLOCK / 64], GridInfo *s_info, GridInfo *out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s_red[a][wave] = mn[a];
            s_red[3 + a][wave] = mx[a];
        }
    __syncthreads();
    if (wave == 0) {  // the GB_BLOCK / 64 = 16 wave results, one per lane of a 16-lane row
        float lo3[3], hi3[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo3[a] = s_red[a][lane & (GB_BLOCK / 64 - 1)];
            hi3[a] = s_red[3 + a][lane & (GB_BLOCK / 64 - 1)];
#pragma unroll
            for (int off = GB_BLOCK / 128; off >= 1; off >>= 1) {
                lo3[a] = fminf(lo3[a], __shfl_xor(lo3[a], off));
                hi3[a] = fmaxf(hi3[a], __shfl_xor(hi3[a], off));
            }
        }
        if (L == 0) lo3[0] = lo3[1] = lo3[2] = hi3[0] = hi3[1] = hi3[2] = 0.0f;
        const GridInfo gi = make_grid_info(lo3, hi3, L, radius, target, G);
        if (lane == 0) {
            *s_info = gi;
            *out = gi;
        }
    }
    __syncthreads();
}

// seg == nullptr: dense (cloud b owns rows [b*N, (b+1)*N)); else rows [seg[b], seg[b+1]).
__global__ __launch_bounds__(GB_BLOCK) void grid_build_kernel(const float *__restrict__ x, const int64_t *__restrict__ seg,
                                                               int N, float radius, float target, int G,
                                                               GridInfo *__restrict__ info,
                                                               int *__restrict__ cell_start /*[B][G^3+1]*/,
                                                               float4 *__restrict__ sorted_pt /*[rows]*/)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float s_red[6][GB_BLOCK / 64];
    __shared__ int s_scan[32];
    __shared__ GridInfo s_info;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t lo = seg ? seg[b] : (int64_t)b * N;
    const int L = seg ? (int)(seg[b + 1] - seg[b]) : N;
    const float *p = x + lo * 3;
    const int nbins = G * G * G;
    int *cnt = reinterpret_cast<int *>(smem);
    unsigned short *ord = reinterpret_cast<unsigned short *>(smem + (((size_t)nbins * 4 + 15) & ~(size_t)15));
    int *cs_out = cell_start + (size_t)b * (nbins + 1);

    // ---- bounding box
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int j = tid; j < L; j += GB_BLOCK)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = p[(size_t)j * 3 + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    grid_box_to_info(mn, mx, L, radius, target, G, s_red, &s_info, &info[b]);
    const GridInfo gi = s_info;
    // only the gx*gy*gz cells the cloud actually spans are touched (the arrays are sized for the G^3 worst case)
    const int nused = gi.gx * gi.gy * gi.gz;
    for (int k = tid; k < nused; k += GB_BLOCK) cnt[k] = 0;
    __syncthreads();
    auto cell_of = [&](int j) {
        const int cx = cell_coord(p[(size_t)j * 3 + 0], gi.minx, gi.inv_cs, gi.gx);
        const int cy = cell_coord(p[(size_t)j * 3 + 1], gi.miny, gi.inv_cs, gi.gy);
        const int cz = cell_coord(p[(size_t)j * 3 + 2], gi.minz, gi.inv_cs, gi.gz);
        return (cz * gi.gy + cy) * gi.gx + cx;  // x fastest: the 3 x-neighbours of a cell are consecutive bins
    };
    for (int j = tid; j < L; j += GB_BLOCK) atomicAdd(&cnt[cell_of(j)], 1);
    __syncthreads();
    grid_scan_cells(cnt, cs_out, nused, L, s_scan);
    __syncthreads();
    for (int j = tid; j < L; j += GB_BLOCK) {
        const int pos = atomicAdd(&cnt[cell_of(j)], 1);  // cnt[k] ends as the END of bin k
        ord[pos] = (unsigned short)j;
    }
    __syncthreads();
    // (the order inside a cell is whatever the atomics produced: the query ranks its hits by index, so the
    //  output does not depend on it)
    for (int t = tid; t < L; t += GB_BLOCK) {
        const int j = ord[t];
        sorted_pt[lo + t] = make_float4(p[(size_t)j * 3 + 0], p[(size_t)j * 3 + 1], p[(size_t)j * 3 + 2], __int_as_float(j));
    }
}

// The same build for clouds of at most GB_BLOCK * PT points, by gridDim.y workgroups per cloud.  Every thread keeps its
// PT points in registers from one batch of loads (no second or third walk over the cloud, no id table in LDS); every
// workgroup of a cloud reads the whole cloud and derives the same box and cells, but owns one contiguous slab of the
// cell range: it histograms, scans and fills only the points of its slab, behind the count of the points in front of
// it.  One workgroup per cloud is bound by what one CU can load and scatter (20 us for 16384 points); the slabs spread
// that over the chip, and blockIdx.y-major launch order keeps the workgroups of a cloud on one XCD (shared L2).
template <int PT>
__global__ __launch_bounds__(GB_BLOCK) void grid_build_reg_kernel(const float *__restrict__ x, const int64_t *__restrict__ seg,
                                                                   int N, float radius, float target, int G,
                                                                   GridInfo *__restrict__ info, int *__restrict__ cell_start,
                                                                   float4 *__restrict__ sorted_pt)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float s_red[6][GB_BLOCK / 64];
    __shared__ int s_scan[32];
    __shared__ int s_before;
    __shared__ GridInfo s_info;
    const int b = blockIdx.x, slab = blockIdx.y, slabs = gridDim.y, tid = threadIdx.x, lane = tid & 63;
    const int64_t lo = seg ? seg[b] : (int64_t)b * N;
    const int L = seg ? (int)(seg[b + 1] - seg[b]) : N;
    const float *p = x + lo * 3;
    const int nbins = G * G * G;
    int *cnt = reinterpret_cast<int *>(smem);
    int *cs_out = cell_start + (size_t)b * (nbins + 1);

    float px[PT], py[PT], pz[PT];
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int i = 0; i < PT; ++i) px[i] = py[i] = pz[i] = 0.0f;
    if (L > 0) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {  // unconditional loads (clamped row), so that all 3 PT are in flight together
            const int j = min(tid + i * GB_BLOCK, L - 1);
            px[i] = p[(size_t)j * 3 + 0];
            py[i] = p[(size_t)j * 3 + 1];
            pz[i] = p[(size_t)j * 3 + 2];
        }
    }
#pragma unroll
    for (int i = 0; i < PT; ++i)
        if (tid + i * GB_BLOCK < L) {
            mn[0] = fminf(mn[0], px[i]), mx[0] = fmaxf(mx[0], px[i]);
            mn[1] = fminf(mn[1], py[i]), mx[1] = fmaxf(mx[1], py[i]);
            mn[2] = fminf(mn[2], pz[i]), mx[2] = fmaxf(mx[2], pz[i]);
        }
    if (tid == 0) s_before = 0;
    // (min / max do not depend on the order of the reduction: every workgroup of the cloud gets the same GridInfo)
    grid_box_to_info(mn, mx, L, radius, target, G, s_red, &s_info, &info[b]);
    const GridInfo gi = s_info;
    const int nused = gi.gx * gi.gy * gi.gz;
    const int c_lo = (int)((int64_t)nused * slab / slabs), c_hi = (int)((int64_t)nused * (slab + 1) / slabs);
    const int mine = c_hi - c_lo;
    for (int k = tid; k < mine; k += GB_BLOCK) cnt[k] = 0;
    __syncthreads();
    int cell[PT];
    int before = 0;  // this thread's points in the slabs in front
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int cx = cell_coord(px[i], gi.minx, gi.inv_cs, gi.gx);
        const int cy = cell_coord(py[i], gi.miny, gi.inv_cs, gi.gy);
        const int cz = cell_coord(pz[i], gi.minz, gi.inv_cs, gi.gz);
        const int c = (cz * gi.gy + cy) * gi.gx + cx;
        const bool live = tid + i * GB_BLOCK < L;
        cell[i] = (live && c >= c_lo && c < c_hi) ? c - c_lo : -1;
        before += (live && c < c_lo) ? 1 : 0;
        if (cell[i] >= 0) atomicAdd(&cnt[cell[i]], 1);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) before += __shfl_xor(before, off);
    if (lane == 0 && before) atomicAdd(&s_before, before);
    __syncthreads();
    const int base = s_before;
    grid_scan_cells(cnt, cs_out + c_lo, mine, 0, s_scan, base, slab + 1 == slabs ? L : -1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PT; ++i)
        if (cell[i] >= 0) {
            const int pos = atomicAdd(&cnt[cell[i]], 1);  // cnt[] holds the cloud-relative starts of the slab's cells
            sorted_pt[lo + pos] = make_float4(px[i], py[i], pz[i], __int_as_float(tid + i * GB_BLOCK));
        }
}

// A query with more hits than LDS slots (a very dense ball): exact in-order scan of its cloud by the wave (inlined: as a
// call it costs the common path register spills).
__device__ __forceinline__ void grid_query_overflow(const float *__restrict__ x, int64_t lo, int L, float qx, float qy, float qz,
                                                 float r2, int nsample, int sort, bool partial, int64_t goff,
                                                 int64_t *__restrict__ io, float *__restrict__ dd)
{
    const int lane = threadIdx.x & 63;
    const int64_t padv_empty = partial ? -1 : 0;
    // Unsorted: the first nsample hits in index order ARE the answer, so stop once nsample are found.
    // Sorted: keep every hit's rank bookkeeping simple by falling back to nsample smallest (d, id) via
    // repeated selection over the scan (rare path; correctness over speed).
    if (!sort) {
        int cnt = 0;
        int first = 0;
        for (int st = 0; st < L && cnt < nsample; st += 64) {
            const int k = st + lane;
            const bool valid = k < L;
            const int kk = valid ? k : 0;
            const float d = sqdist3(x[(lo + kk) * 3 + 0], x[(lo + kk) * 3 + 1], x[(lo + kk) * 3 + 2], qx, qy, qz);
            const bool hit = valid && d < r2;
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                if (cnt == 0) first = st + __builtin_ctzll(mask);
                const int slot = cnt + lanes_below(mask);
                if (hit && slot < nsample) {
                    io[slot] = goff + k;
                    dd[slot] = d;
                }
                cnt += __builtin_popcountll(mask);
            }
        }
        const int64_t padv = partial ? -1 : (cnt ? first : 0);
        for (int s = min(cnt, nsample) + lane; s < nsample; s += 64) {
            io[s] = padv;
            dd[s] = -1.0f;
        }
    } else {
        // selection of the nsample smallest (d, id): pass s finds the smallest pair greater than the previous
        float pd = -1.0f;
        int pi = -1;
        int emitted = 0;
        int64_t firstv = padv_empty;
        for (; emitted < nsample; ++emitted) {
            float bd = 3.0e38f;
            int bi = 0x7fffffff;
            for (int st = 0; st < L; st += 64) {
                const int k = st + lane;
                if (k < L) {
                    const float d = sqdist3(x[(lo + k) * 3 + 0], x[(lo + k) * 3 + 1], x[(lo + k) * 3 + 2], qx, qy, qz);
                    const bool after = d > pd || (d == pd && k > pi);
                    if (d < r2 && after && (d < bd || (d == bd && k < bi))) {
                        bd = d;
                        bi = k;
                    }
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float od = __shfl_xor(bd, off);
                const int oi = __shfl_xor(bi, off);
                if (od < bd || (od == bd && oi < bi)) {
                    bd = od;
                    bi = oi;
                }
            }
            if (bi == 0x7fffffff) break;
            if (lane == 0) {
                io[emitted] = goff + bi;
                dd[emitted] = bd;
            }
            if (emitted == 0) firstv = goff + bi;
            pd = bd;
            pi = bi;
        }
        const int64_t padv = partial ? -1 : firstv;
        for (int s = emitted + lane; s < nsample; s += 64) {
            io[s] = padv;
            dd[s] = -1.0f;
        }
    }
}

// One wave per query.  dense: cloud = q / np, indices cloud-local, pad = first hit (0 if none);
// partial: cloud = batch_y[q], indices global rows, pad = -1.
// BITMAP (index order, clouds of at most GQ_BM_POINTS points): the rank of a hit among the hits is the number of set bits
// below its index in a bitmap of the cloud (one LDS atomic OR per hit, one popcount pass of 8 words per lane, one wave
// scan) instead of a comparison against every other hit -- ~70 instead of ~300-500 VALU instructions for the 70-130 hits of a
// BASELINE ball.
constexpr int GQ_BM_POINTS = 16384, GQ_BM_WORDS = GQ_BM_POINTS / 32, GQ_BM_CAP = 384;
template <bool BITMAP>
__global__ __launch_bounds__(GQ_BLOCK) void grid_query_kernel(
    const float *__restrict__ x, const float *__restrict__ y, const int64_t *__restrict__ seg,
    const int64_t *__restrict__ batch_y, int64_t total_q, int N, int np, int num_clouds, float r2, int nsample, int sort,
    int G, const GridInfo *__restrict__ info, const int *__restrict__ cell_start,
    const float4 *__restrict__ sorted_pt, int64_t *__restrict__ idx, float *__restrict__ dist2)
{
    constexpr int CAP = BITMAP ? GQ_BM_CAP : GQ_CAP;
    __shared__ __attribute__((aligned(16))) int s_id[GQ_BLOCK / 64][CAP + 4];  // + the pad of the four-wide ranking reads
    __shared__ float s_d[GQ_BLOCK / 64][CAP];
    __shared__ __attribute__((aligned(16))) unsigned int s_bm[BITMAP ? GQ_BLOCK / 64 : 1][BITMAP ? GQ_BM_WORDS : 4];
    __shared__ __attribute__((aligned(16))) unsigned short s_pre[BITMAP ? GQ_BLOCK / 64 : 1][BITMAP ? GQ_BM_WORDS : 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * (GQ_BLOCK / 64) + wave;
    if (q >= total_q) return;  // wave-uniform, no workgroup barrier in this kernel
    int *cid = s_id[wave];
    float *cd = s_d[wave];
    const bool partial = seg != nullptr;
    const int64_t bq = partial ? batch_y[q] : q / np;
    int64_t *io = idx + q * nsample;
    float *dd = dist2 + q * nsample;
    const int64_t padv_empty = partial ? -1 : 0;
    if (bq < 0 || bq >= num_clouds) {  // a query whose cloud has no support points at all
        for (int s = lane; s < nsample; s += 64) {
            io[s] = padv_empty;
            dd[s] = -1.0f;
        }
        return;
    }
    const int64_t lo = partial ? seg[bq] : bq * N;
    const int L = partial ? (int)(seg[bq + 1] - seg[bq]) : N;
    const int64_t goff = partial ? lo : 0;  // partial-dense indices are global rows
    const float qx = y[q * 3 + 0], qy = y[q * 3 + 1], qz = y[q * 3 + 2];
    const GridInfo gi = info[bq];
    const int nbins = G * G * G;
    const int *cs = cell_start + (size_t)bq * (nbins + 1);

    // unclamped cell of the query; cells outside [-1, g] cannot touch the ball
    // (clamped in float first: a far-away query must not overflow the int conversion)
    const float far = (float)(G + 8);
    const int cx = (int)floorf(fminf(fmaxf((qx - gi.minx) * gi.inv_cs, -4.0f), far));
    const int cy = (int)floorf(fminf(fmaxf((qy - gi.miny) * gi.inv_cs, -4.0f), far));
    const int cz = (int)floorf(fminf(fmaxf((qz - gi.minz) * gi.inv_cs, -4.0f), far));
    int h = 0;  // hits so far (wave-uniform)
    bool overflow = false;
    unsigned int *bm = s_bm[BITMAP ? wave : 0];
    if (BITMAP) {  // this wave's bitmap: lane l owns the words [8l, 8l + 8)
        *reinterpret_cast<uint4 *>(&bm[lane * 8]) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4 *>(&bm[lane * 8 + 4]) = make_uint4(0u, 0u, 0u, 0u);
    }
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, gi.gx - 1);
    if (x0 <= x1) {
        // the nine (z, y) rows around the query are contiguous x-runs of <= 3 cells: lanes 0..8 fetch their slot
        // ranges at once (one load latency instead of nine), the loop reads them back with v_readlane
        int j0v = 0, j1v = 0;
        if (lane < 9) {
            const int zz = cz + lane / 3 - 1, yy = cy + lane % 3 - 1;
            if (zz >= 0 && zz < gi.gz && yy >= 0 && yy < gi.gy) {
                const int rowbase = (zz * gi.gy + yy) * gi.gx;
                j0v = cs[rowbase + x0];
                j1v = cs[rowbase + x1 + 1];
            }
        }
        // the first 64 points of all nine runs are requested before any is tested (a run is rarely longer): nine
        // independent 16-byte loads per lane instead of nine dependent load -> test -> append rounds
        float4 pre[9];
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
            const int j0 = __builtin_amdgcn_readlane(j0v, rr);
            pre[rr] = sorted_pt[lo + min(j0 + lane, L - 1)];
        }
#pragma unroll
        for (int rr = 0; rr < 9; ++rr) {
            const int j0 = __builtin_amdgcn_readlane(j0v, rr), j1 = __builtin_amdgcn_readlane(j1v, rr);
            for (int j = j0; j < j1 && !overflow; j += 64) {
                const int t = j + lane;
                const bool valid = t < j1;
                const float4 pt = (j == j0) ? pre[rr] : sorted_pt[lo + (valid ? t : j0)];
                const float d = sqdist3(pt.x, pt.y, pt.z, qx, qy, qz);
                const bool hit = valid && d < r2;
                const unsigned long long mask = __ballot(hit);
                if (mask) {
                    const int cntm = __builtin_popcountll(mask);
                    if (h + cntm > CAP) {
                        overflow = true;
                        break;
                    }
                    if (hit) {
                        const int slot = h + lanes_below(mask);
                        const int id = __float_as_int(pt.w);
                        cid[slot] = id;
                        cd[slot] = d;
                        if (BITMAP) atomicOr(&bm[id >> 5], 1u << (id & 31));
                    }
                    h += cntm;
                }
            }
        }
    }
    if (overflow) {
        grid_query_overflow(x, lo, L, qx, qy, qz, r2, nsample, sort, partial, goff, io, dd);
        return;
    }
    if (!sort && lane < 4) cid[h + lane] = 0x7fffffff;  // the index ranking reads four ids at a time
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- rank the h candidates: by index (sort=0) or by (distance, index) (sort=1); emit ranks < nsample
    int64_t firstv = padv_empty;
    if (BITMAP) {
        // exclusive count of set bits in front of every word: 8 words per lane, then a scan over the lanes
        unsigned short *pre = s_pre[wave];
        const uint4 w0 = *reinterpret_cast<const uint4 *>(&bm[lane * 8]), w1 = *reinterpret_cast<const uint4 *>(&bm[lane * 8 + 4]);
        const unsigned int wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
        int run[8], tot = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            run[u] = tot;
            tot += __builtin_popcount(wv[u]);
        }
        int incl = tot;  // inclusive scan of the lane totals (Hillis-Steele on the cross-lane network)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        const int base = incl - tot;
        *reinterpret_cast<uint4 *>(&pre[lane * 8]) =
            make_uint4((unsigned)(base + run[0]) | ((unsigned)(base + run[1]) << 16), (unsigned)(base + run[2]) | ((unsigned)(base + run[3]) << 16),
                       (unsigned)(base + run[4]) | ((unsigned)(base + run[5]) << 16), (unsigned)(base + run[6]) | ((unsigned)(base + run[7]) << 16));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int t0 = 0; t0 < h; t0 += 64) {  // (wave-uniform trip count: the ballot / broadcast below need every lane)
            const int t = t0 + lane;
            const bool have = t < h;
            const int id = have ? cid[t] : 0;
            const int rank = (int)pre[id >> 5] + __builtin_popcount(bm[id >> 5] & ((1u << (id & 31)) - 1u));
            if (have && rank < nsample) {
                io[rank] = goff + id;
                dd[rank] = cd[t];
            }
            const unsigned long long zero = __ballot(have && rank == 0);
            if (zero) firstv = goff + __shfl(id, __builtin_ctzll(zero));
        }
    } else if (!sort) {
        // two candidates per lane and four ids per LDS read: the kernel is bound by the VALU instructions it issues, and
        // a ball of 65..128 hits (the common case at nsample 64) would otherwise walk the list twice
        for (int t0 = 0; t0 < h; t0 += 128) {
            const int ta = t0 + lane, tb = ta + 64;
            const bool ha = ta < h, hb = tb < h, two = t0 + 64 < h;
            const int ma = ha ? cid[ta] : 0x7fffffff, mb = hb ? cid[tb] : 0x7fffffff;
            int ra = 0, rb = 0;
            if (two) {
                for (int u = 0; u < h; u += 4) {
                    const int4 v = *reinterpret_cast<const int4 *>(&cid[u]);
                    ra += (v.x < ma ? 1 : 0) + (v.y < ma ? 1 : 0) + (v.z < ma ? 1 : 0) + (v.w < ma ? 1 : 0);
                    rb += (v.x < mb ? 1 : 0) + (v.y < mb ? 1 : 0) + (v.z < mb ? 1 : 0) + (v.w < mb ? 1 : 0);
                }
            } else {
                for (int u = 0; u < h; u += 4) {
                    const int4 v = *reinterpret_cast<const int4 *>(&cid[u]);
                    ra += (v.x < ma ? 1 : 0) + (v.y < ma ? 1 : 0) + (v.z < ma ? 1 : 0) + (v.w < ma ? 1 : 0);
                }
            }
            if (ha && ra < nsample) {
                io[ra] = goff + ma;
                dd[ra] = cd[ta];
            }
            if (hb && rb < nsample) {
                io[rb] = goff + mb;
                dd[rb] = cd[tb];
            }
            const unsigned long long za = __ballot(ha && ra == 0), zb = __ballot(hb && rb == 0);
            if (za) firstv = goff + __shfl(ma, __builtin_ctzll(za));
            if (zb) firstv = goff + __shfl(mb, __builtin_ctzll(zb));
        }
    } else {
        for (int t = lane; t < ((h + 63) & ~63); t += 64) {
            const bool have = t < h;
            const int mi = have ? cid[t] : 0x7fffffff;
            const float md = have ? cd[t] : 3.0e38f;
            int rank = 0;
            for (int u = 0; u < h; ++u) {
                const float ud = cd[u];
                const int ui = cid[u];
                rank += (ud < md || (ud == md && ui < mi)) ? 1 : 0;
            }
            if (have && rank < nsample) {
                io[rank] = goff + mi;
                dd[rank] = md;
            }
            const unsigned long long zero = __ballot(have && rank == 0);
            if (zero) firstv = goff + __shfl(mi, __builtin_ctzll(zero));
        }
    }
    const int64_t padv = partial ? -1 : firstv;
    for (int s = min(h, nsample) + lane; s < nsample; s += 64) {
        io[s] = padv;
        dd[s] = -1.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Sort-based build for clouds that do not fit one workgroup's LDS (any size, any number of workgroups)
constexpr int GG_BLOCK = 256;
constexpr int GG_ROWS = GG_BLOCK * 16;  // rows per workgroup in the per-cloud passes

__device__ __forceinline__ int float_order(float f)  // order-preserving map float -> int (for atomicMin/Max)
{
    const int i = __float_as_int(f);
    return i ^ ((i >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float order_float(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7fffffff)); }

__global__ void gridg_init_kernel(int *__restrict__ bbox, int num_clouds)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < num_clouds * 6) bbox[t] = (t % 6) < 3 ? 0x7fffffff : (int)0x80000000;
}

// grid = (chunks, clouds)
__global__ __launch_bounds__(GG_BLOCK) void gridg_bbox_kernel(const float *__restrict__ x, const int64_t *__restrict__ seg,
                                                               int N, int *__restrict__ bbox)
{
    __shared__ float s_red[6][GG_BLOCK / 64];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t lo = seg ? seg[b] : (int64_t)b * N;
    const int64_t L = seg ? seg[b + 1] - seg[b] : N;
    const int64_t j0 = (int64_t)blockIdx.x * GG_ROWS;
    if (j0 >= L) return;  // workgroup-uniform
    const int64_t j1 = min(j0 + GG_ROWS, L);
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int64_t j = j0 + tid; j < j1; j += GG_BLOCK)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = x[(lo + j) * 3 + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s_red[a][wave] = mn[a];
            s_red[3 + a][wave] = mx[a];
        }
    __syncthreads();
    if (tid < 6) {
        float r = s_red[tid][0];
        for (int w = 1; w < GG_BLOCK / 64; ++w) r = tid < 3 ? fminf(r, s_red[tid][w]) : fmaxf(r, s_red[tid][w]);
        if (tid < 3) atomicMin(&bbox[b * 6 + tid], float_order(r));
        else atomicMax(&bbox[b * 6 + tid], float_order(r));
    }
}

__global__ void gridg_info_kernel(const int *__restrict__ bbox, const int64_t *__restrict__ seg, int N, int num_clouds,
                                  float cell, float target, int G, GridInfo *__restrict__ info)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= num_clouds) return;
    const int L = seg ? (int)(seg[b + 1] - seg[b]) : N;
    float lo3[3], hi3[3];
    for (int a = 0; a < 3; ++a) {
        lo3[a] = L ? order_float(bbox[b * 6 + a]) : 0.0f;
        hi3[a] = L ? order_float(bbox[b * 6 + 3 + a]) : 0.0f;
    }
    info[b] = make_grid_info(lo3, hi3, L, cell, target, G);
}

// grid = (chunks, clouds): key = cloud * G^3 + cell, value = global row
__global__ __launch_bounds__(GG_BLOCK) void gridg_key_kernel(const float *__restrict__ x, const int64_t *__restrict__ seg,
                                                              int N, int G, const GridInfo *__restrict__ info,
                                                              unsigned long long *__restrict__ keys,
                                                              unsigned int *__restrict__ vals)
{
    const int b = blockIdx.y;
    const int64_t lo = seg ? seg[b] : (int64_t)b * N;
    const int64_t L = seg ? seg[b + 1] - seg[b] : N;
    const int64_t j0 = (int64_t)blockIdx.x * GG_ROWS;
    if (j0 >= L) return;
    const int64_t j1 = min(j0 + GG_ROWS, L);
    const GridInfo gi = info[b];
    const unsigned long long base = (unsigned long long)b * (unsigned long long)(G * G * G);
    for (int64_t j = j0 + threadIdx.x; j < j1; j += GG_BLOCK) {
        const float *p = x + (lo + j) * 3;
        const int cx = cell_coord(p[0], gi.minx, gi.inv_cs, gi.gx);
        const int cy = cell_coord(p[1], gi.miny, gi.inv_cs, gi.gy);
        const int cz = cell_coord(p[2], gi.minz, gi.inv_cs, gi.gz);
        keys[lo + j] = base + (unsigned long long)((cz * gi.gy + cy) * gi.gx + cx);
        vals[lo + j] = (unsigned int)(lo + j);
    }
}

// Keys are cloud-major, so after the sort cloud b still owns slots [lo_b, lo_b + L_b).
__global__ __launch_bounds__(GG_BLOCK) void gridg_fill_kernel(const float *__restrict__ x, const int64_t *__restrict__ seg,
                                                               int N, int G, int64_t rows,
                                                               const unsigned long long *__restrict__ keys,
                                                               const unsigned int *__restrict__ vals,
                                                               float4 *__restrict__ sorted_pt)
{
    const int64_t t = (int64_t)blockIdx.x * GG_BLOCK + threadIdx.x;
    if (t >= rows) return;
    const int64_t row = vals[t];
    const int b = (int)(keys[t] / (unsigned long long)(G * G * G));
    const int64_t lo = seg ? seg[b] : (int64_t)b * N;
    sorted_pt[t] = make_float4(x[row * 3 + 0], x[row * 3 + 1], x[row * 3 + 2], __int_as_float((int)(row - lo)));
}

// grid = (cell chunks, clouds): cell_start[b][c] = first slot (cloud-relative) whose key is >= cell c
__global__ __launch_bounds__(GG_BLOCK) void gridg_cellstart_kernel(const int64_t *__restrict__ seg, int N, int G,
                                                                    const GridInfo *__restrict__ info,
                                                                    const unsigned long long *__restrict__ keys,
                                                                    int *__restrict__ cell_start)
{
    const int b = blockIdx.y;
    const GridInfo gi = info[b];
    const int nused = gi.gx * gi.gy * gi.gz;
    const int c = blockIdx.x * GG_BLOCK + threadIdx.x;
    if (c > nused) return;
    const int64_t lo = seg ? seg[b] : (int64_t)b * N;
    const int L = seg ? (int)(seg[b + 1] - seg[b]) : N;
    const unsigned long long want = (unsigned long long)b * (unsigned long long)(G * G * G) + (unsigned long long)c;
    int a = 0, z = L;  // lower bound in keys[lo, lo + L)
    while (a < z) {
        const int m = (a + z) >> 1;
        if (keys[lo + m] < want) a = m + 1;
        else z = m;
    }
    cell_start[(size_t)b * ((size_t)G * G * G + 1) + c] = a;
}

constexpr size_t GRID_LDS_BUDGET = 144 * 1024;

// largest grid edge whose histogram + u16 order array of an Lmax-point cloud fit LDS (0: cloud too large)
int grid_edge_for(int Lmax)
{
    if (Lmax > GRID_LDS_MAX_POINTS) return 0;
    const size_t left = GRID_LDS_BUDGET - (size_t)Lmax * 2 - 64;
    int G = 32;
    while (G > 1 && ((size_t)G * G * G * 4) > left) --G;
    return G;
}

// Clouds larger than this use the sort-based builder
static int grid_global_min_points() { return GRID_LDS_MAX_POINTS; }

GridPlan grid_plan(int Lmax)
{
    GridPlan p;
    p.global = Lmax > grid_global_min_points();
    if (!p.global) {
        p.G = grid_edge_for(Lmax);
    } else {
        // scans sample surfaces: the occupied cells are a thin subset of the box, so the table is sized for twice as
        // many cells as points (126^3 for 10^6 points) rather than for a volume-filling cloud
        int G = 16;
        while (G < GRID_GLOBAL_MAX_EDGE && (int64_t)G * G * G < 2 * (int64_t)Lmax) ++G;
        p.G = G;
    }
    return p;
}

GridWorkspace carve_grid_workspace(void *ws, int num_clouds, int64_t rows, GridPlan plan)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const int G = plan.G;
    GridWorkspace w;
    char *p = static_cast<char *>(ws);
    size_t off = 0;
    w.info = reinterpret_cast<GridInfo *>(p + off);
    off += up((size_t)num_clouds * sizeof(GridInfo));
    w.cell_start = reinterpret_cast<int *>(p + off);
    off += up((size_t)num_clouds * ((size_t)G * G * G + 1) * 4);
    w.sorted_pt = reinterpret_cast<float4 *>(p + off);
    off += up((size_t)rows * 16);
    w.bbox = nullptr;
    w.keys_in = w.keys_out = nullptr;
    w.vals_in = w.vals_out = nullptr;
    w.sort_tmp = nullptr;
    w.sort_tmp_bytes = 0;
    if (plan.global) {
        w.bbox = reinterpret_cast<int *>(p + off);
        off += up((size_t)num_clouds * 6 * 4);
        w.keys_in = reinterpret_cast<unsigned long long *>(p + off);
        off += up((size_t)rows * 8);
        w.keys_out = reinterpret_cast<unsigned long long *>(p + off);
        off += up((size_t)rows * 8);
        w.vals_in = reinterpret_cast<unsigned int *>(p + off);
        off += up((size_t)rows * 4);
        w.vals_out = reinterpret_cast<unsigned int *>(p + off);
        off += up((size_t)rows * 4);
        w.sort_tmp = p + off;
        w.sort_tmp_bytes = sort_pairs_tmp_bytes(rows);
        off += up(w.sort_tmp_bytes + 256);
    }
    w.bytes = off;
    return w;
}

int grid_build(const float *x, const int64_t *seg, int num_clouds, int64_t rows, int N, int Lmax, float cell, float target,
               GridPlan plan, const GridWorkspace &w, hipStream_t s)
{
    const int G = plan.G;
    if (G < 2 || num_clouds <= 0) return TP3D_E_TOOBIG;
    if (!plan.global) {
        const size_t lds = (((size_t)G * G * G * 4 + 15) & ~(size_t)15) + (size_t)Lmax * 2;
        static bool attr_set[64] = {false};
        allow_large_dynamic_lds(reinterpret_cast<const void *>(&grid_build_kernel), (int)GRID_LDS_BUDGET, attr_set);
        if (Lmax <= GB_BLOCK * 16) {
            const size_t lds_reg = (size_t)G * G * G * 4;
#define TP3D_BUILD_REG(PT)                                                                                            \
    do {                                                                                                              \
        static bool set_##PT[64] = {false};                                                                           \
        allow_large_dynamic_lds(reinterpret_cast<const void *>(&grid_build_reg_kernel<PT>), (int)GRID_LDS_BUDGET,     \
                                set_##PT);                                                                            \
        hipLaunchKernelGGL(grid_build_reg_kernel<PT>, dim3(num_clouds, 8), dim3(GB_BLOCK), lds_reg, s, x, seg, N, cell, \
                           target, G, w.info, w.cell_start, w.sorted_pt);                                             \
    } while (0)
            if (Lmax <= GB_BLOCK * 4)
                TP3D_BUILD_REG(4);
            else if (Lmax <= GB_BLOCK * 8)
                TP3D_BUILD_REG(8);
            else
                TP3D_BUILD_REG(16);
#undef TP3D_BUILD_REG
            return check_launch();
        }
        hipLaunchKernelGGL(grid_build_kernel, dim3(num_clouds), dim3(GB_BLOCK), lds, s, x, seg, N, cell, target, G, w.info,
                           w.cell_start, w.sorted_pt);
        return check_launch();
    }
    if (rows >= 0x7fffffff || num_clouds > 65535) return TP3D_E_TOOBIG;
    const unsigned chunks = (unsigned)(((int64_t)Lmax + GG_ROWS - 1) / GG_ROWS);
    hipLaunchKernelGGL(gridg_init_kernel, dim3((num_clouds * 6 + 255) / 256), dim3(256), 0, s, w.bbox, num_clouds);
    hipLaunchKernelGGL(gridg_bbox_kernel, dim3(chunks, num_clouds), dim3(GG_BLOCK), 0, s, x, seg, N, w.bbox);
    hipLaunchKernelGGL(gridg_info_kernel, dim3((num_clouds + 63) / 64), dim3(64), 0, s, w.bbox, seg, N, num_clouds, cell,
                       target, G, w.info);
    hipLaunchKernelGGL(gridg_key_kernel, dim3(chunks, num_clouds), dim3(GG_BLOCK), 0, s, x, seg, N, G, w.info, w.keys_in,
                       w.vals_in);
    if (int rc = check_launch()) return rc;
    unsigned bits = 1;
    const unsigned long long total = (unsigned long long)num_clouds * (unsigned long long)G * G * G;
    while (bits < 63 && (1ull << bits) < total) ++bits;
    if (int rc = sort_pairs_u64_u32(w.sort_tmp, w.sort_tmp_bytes, w.keys_in, w.keys_out, w.vals_in, w.vals_out, rows, bits, s))
        return rc;
    hipLaunchKernelGGL(gridg_fill_kernel, dim3((unsigned)((rows + GG_BLOCK - 1) / GG_BLOCK)), dim3(GG_BLOCK), 0, s, x, seg,
                       N, G, rows, w.keys_out, w.vals_out, w.sorted_pt);
    const unsigned cchunks = (unsigned)(((size_t)G * G * G + 1 + GG_BLOCK - 1) / GG_BLOCK);
    hipLaunchKernelGGL(gridg_cellstart_kernel, dim3(cchunks, num_clouds), dim3(GG_BLOCK), 0, s, seg, N, G, w.info,
                       w.keys_out, w.cell_start);
    return check_launch();
}

// Enqueue build + query. seg/batch_y null => dense layout.
int grid_ball_query(const float *x, const float *y, const int64_t *seg, const int64_t *batch_y, int num_clouds,
                    int64_t rows, int N, int np, int64_t total_q, int Lmax, float radius, int nsample, int sort,
                    int64_t *idx, float *dist2, void *workspace, size_t workspace_bytes, bool reuse_grid, hipStream_t s)
{
    const GridPlan plan = grid_plan(Lmax);
    if (plan.G < 2) return TP3D_E_TOOBIG;
    GridWorkspace w = carve_grid_workspace(workspace, num_clouds, rows, plan);
    if (workspace_bytes < w.bytes) return TP3D_E_BADARG;
    // reuse_grid: the workspace still holds the tables of the previous call with the same support, segments and radius
    if (!reuse_grid)
        if (int rc = grid_build(x, seg, num_clouds, rows, N, Lmax, radius, 2.0f, plan, w, s)) return rc;
    const int64_t blocks = (total_q + GQ_BLOCK / 64 - 1) / (GQ_BLOCK / 64);
    if (blocks > 0x7fffffff) return TP3D_E_TOOBIG;
    if (!sort && Lmax <= GQ_BM_POINTS)
        hipLaunchKernelGGL(grid_query_kernel<true>, dim3((unsigned)blocks), dim3(GQ_BLOCK), 0, s, x, y, seg, batch_y, total_q, N,
                           np, num_clouds, radius * radius, nsample, sort, plan.G, w.info, w.cell_start, w.sorted_pt,
                           idx, dist2);
    else
        hipLaunchKernelGGL(grid_query_kernel<false>, dim3((unsigned)blocks), dim3(GQ_BLOCK), 0, s, x, y, seg, batch_y, total_q, N,
                           np, num_clouds, radius * radius, nsample, sort, plan.G, w.info, w.cell_start, w.sorted_pt,
                           idx, dist2);
    return check_launch();
}

}  // namespace tp3d

TP3D_EXPORT size_t tp3d_ball_query_workspace_bytes(int num_clouds, int64_t rows, int max_cloud_points)
{
    if (num_clouds <= 0 || rows < 0 || max_cloud_points <= 0) return 0;
    const tp3d::GridPlan plan = tp3d::grid_plan(max_cloud_points);
    if (plan.G < 2) return 0;
    return tp3d::carve_grid_workspace(nullptr, num_clouds, rows, plan).bytes;
}
