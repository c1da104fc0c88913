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
#include "tp3d_common.h"

namespace tp3d {

constexpr int GB_BLOCK = 1024;
constexpr int GQ_BLOCK = 256;     // 4 waves, one query per wave
constexpr int GQ_CAP = 1024;      // candidate slots per query (overflow -> exact in-order scan of the cloud)

struct GridInfo {  // per cloud, 8 floats
    float minx, miny, minz, inv_cs;
    int gx, gy, gz, pad;
};

__device__ __forceinline__ int cell_coord(float x, float mn, float inv_cs, int g)
{
    int c = (int)floorf((x - mn) * inv_cs);
    return min(max(c, 0), g - 1);
}

// seg == nullptr: dense (cloud b owns rows [b*N, (b+1)*N)); else rows [seg[b], seg[b+1]).
__global__ __launch_bounds__(GB_BLOCK) void grid_build_kernel(const float *__restrict__ x, const int64_t *__restrict__ seg,
                                                               int N, float radius, int G, GridInfo *__restrict__ info,
                                                               int *__restrict__ cell_start /*[B][G^3+1]*/,
                                                               int *__restrict__ sorted_id /*[rows]*/,
                                                               float *__restrict__ sorted_xyz /*[rows][3]*/)
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
    if (tid == 0) {
        float lo3[3], hi3[3];
        for (int a = 0; a < 3; ++a) {
            lo3[a] = s_red[a][0];
            hi3[a] = s_red[3 + a][0];
            for (int w = 1; w < GB_BLOCK / 64; ++w) {
                lo3[a] = fminf(lo3[a], s_red[a][w]);
                hi3[a] = fmaxf(hi3[a], s_red[3 + a][w]);
            }
        }
        if (L == 0) lo3[0] = lo3[1] = lo3[2] = hi3[0] = hi3[1] = hi3[2] = 0.0f;
        const float ext = fmaxf(fmaxf(hi3[0] - lo3[0], hi3[1] - lo3[1]), hi3[2] - lo3[2]);
        // cell edge: at least 1.01 r (so +-1 cell covers the ball with margin for fp32 rounding of the cell
        // coordinate, |coordinate| <= 32), and coarse enough that no axis needs more than G cells
        float cs = fmaxf(radius * 1.01f, ext / (float)G * 1.0001f);
        if (!(cs > 0.0f)) cs = 1.0f;
        GridInfo gi;
        gi.minx = lo3[0];
        gi.miny = lo3[1];
        gi.minz = lo3[2];
        gi.inv_cs = 1.0f / cs;
        gi.gx = min(G, (int)floorf((hi3[0] - lo3[0]) * gi.inv_cs) + 1);
        gi.gy = min(G, (int)floorf((hi3[1] - lo3[1]) * gi.inv_cs) + 1);
        gi.gz = min(G, (int)floorf((hi3[2] - lo3[2]) * gi.inv_cs) + 1);
        gi.pad = 0;
        s_info = gi;
        info[b] = gi;
    }
    __syncthreads();
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
    // ---- exclusive scan of the histogram (per-thread serial chunk + wave scan + cross-wave)
    {
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
        if (tid == 0) {
            int run = 0;
            for (int w = 0; w < GB_BLOCK / 64; ++w) {
                const int v = s_scan[w];
                s_scan[w] = run;
                run += v;
            }
        }
        __syncthreads();
        int run = s_scan[wave] + incl - sum;
        for (int k = k0; k < k1; ++k) {
            const int v = cnt[k];
            cnt[k] = run;
            cs_out[k] = run;  // start of bin k
            run += v;
        }
        if (tid == 0) cs_out[nused] = L;
    }
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
        sorted_id[lo + t] = j;
        sorted_xyz[(lo + t) * 3 + 0] = p[(size_t)j * 3 + 0];
        sorted_xyz[(lo + t) * 3 + 1] = p[(size_t)j * 3 + 1];
        sorted_xyz[(lo + t) * 3 + 2] = p[(size_t)j * 3 + 2];
    }
}

// One wave per query.  dense: cloud = q / np, indices cloud-local, pad = first hit (0 if none);
// partial: cloud = batch_y[q], indices global rows, pad = -1.
__global__ __launch_bounds__(GQ_BLOCK) void grid_query_kernel(
    const float *__restrict__ x, const float *__restrict__ y, const int64_t *__restrict__ seg,
    const int64_t *__restrict__ batch_y, int64_t total_q, int N, int np, int num_clouds, float r2, int nsample, int sort,
    int G, const GridInfo *__restrict__ info, const int *__restrict__ cell_start, const int *__restrict__ sorted_id,
    const float *__restrict__ sorted_xyz, int64_t *__restrict__ idx, float *__restrict__ dist2)
{
    __shared__ int s_id[GQ_BLOCK / 64][GQ_CAP];
    __shared__ float s_d[GQ_BLOCK / 64][GQ_CAP];
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
    const int cx = (int)floorf(fminf(fmaxf((qx - gi.minx) * gi.inv_cs, -4.0f), 40.0f));
    const int cy = (int)floorf(fminf(fmaxf((qy - gi.miny) * gi.inv_cs, -4.0f), 40.0f));
    const int cz = (int)floorf(fminf(fmaxf((qz - gi.minz) * gi.inv_cs, -4.0f), 40.0f));
    int h = 0;  // hits so far (wave-uniform)
    bool overflow = false;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, gi.gx - 1);
    if (x0 <= x1) {
        for (int dz = -1; dz <= 1 && !overflow; ++dz) {
            const int zz = cz + dz;
            if (zz < 0 || zz >= gi.gz) continue;
            for (int dy = -1; dy <= 1 && !overflow; ++dy) {
                const int yy = cy + dy;
                if (yy < 0 || yy >= gi.gy) continue;
                const int rowbase = (zz * gi.gy + yy) * gi.gx;
                const int j0 = cs[rowbase + x0], j1 = cs[rowbase + x1 + 1];  // one contiguous run of <= 3 cells
                for (int j = j0; j < j1; j += 64) {
                    const int t = j + lane;
                    const bool valid = t < j1;
                    const int tt = valid ? t : j0;
                    const float d = sqdist3(sorted_xyz[(lo + tt) * 3 + 0], sorted_xyz[(lo + tt) * 3 + 1],
                                            sorted_xyz[(lo + tt) * 3 + 2], qx, qy, qz);
                    const bool hit = valid && d < r2;
                    const unsigned long long mask = __ballot(hit);
                    if (mask) {
                        const int cntm = __builtin_popcountll(mask);
                        if (h + cntm > GQ_CAP) {
                            overflow = true;
                            break;
                        }
                        if (hit) {
                            const int slot = h + lanes_below(mask);
                            cid[slot] = sorted_id[lo + tt];
                            cd[slot] = d;
                        }
                        h += cntm;
                    }
                }
            }
        }
    }
    if (overflow) {
        // more candidates than LDS slots (very dense ball): exact in-order scan of the cloud for this query.
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
        return;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- rank the h candidates: by index (sort=0) or by (distance, index) (sort=1); emit ranks < nsample
    int64_t firstv = padv_empty;
    for (int t = lane; t < ((h + 63) & ~63); t += 64) {
        const bool have = t < h;
        const int mi = have ? cid[t] : 0x7fffffff;
        const float md = have ? cd[t] : 3.0e38f;
        int rank = 0;
        if (!sort) {
            for (int u = 0; u < h; ++u) rank += (cid[u] < mi) ? 1 : 0;
        } else {
            for (int u = 0; u < h; ++u) {
                const float ud = cd[u];
                const int ui = cid[u];
                rank += (ud < md || (ud == md && ui < mi)) ? 1 : 0;
            }
        }
        if (have && rank < nsample) {
            io[rank] = goff + mi;
            dd[rank] = md;
        }
        const unsigned long long zero = __ballot(have && rank == 0);
        if (zero) firstv = goff + __shfl(mi, __builtin_ctzll(zero));
    }
    const int64_t padv = partial ? -1 : firstv;
    for (int s = min(h, nsample) + lane; s < nsample; s += 64) {
        io[s] = padv;
        dd[s] = -1.0f;
    }
}

constexpr size_t GRID_LDS_BUDGET = 144 * 1024;

// largest grid edge whose histogram + u16 order array of an Lmax-point cloud fit LDS (0: cloud too large)
int grid_edge_for(int Lmax)
{
    if (Lmax > 65536) return 0;
    const size_t left = GRID_LDS_BUDGET - (size_t)Lmax * 2 - 64;
    int G = 32;
    while (G > 1 && ((size_t)G * G * G * 4) > left) --G;
    return G;
}

struct GridWorkspace {
    GridInfo *info;
    int *cell_start;
    int *sorted_id;
    float *sorted_xyz;
    size_t bytes;
};

GridWorkspace carve_grid_workspace(void *ws, int num_clouds, int64_t rows, int G)
{
    auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    GridWorkspace w;
    char *p = static_cast<char *>(ws);
    size_t off = 0;
    w.info = reinterpret_cast<GridInfo *>(p + off);
    off += up((size_t)num_clouds * sizeof(GridInfo));
    w.cell_start = reinterpret_cast<int *>(p + off);
    off += up((size_t)num_clouds * ((size_t)G * G * G + 1) * 4);
    w.sorted_id = reinterpret_cast<int *>(p + off);
    off += up((size_t)rows * 4);
    w.sorted_xyz = reinterpret_cast<float *>(p + off);
    off += up((size_t)rows * 12);
    w.bytes = off;
    return w;
}

// Enqueue build + query. seg/batch_y null => dense layout.
int grid_ball_query(const float *x, const float *y, const int64_t *seg, const int64_t *batch_y, int num_clouds,
                    int64_t rows, int N, int np, int64_t total_q, int Lmax, float radius, int nsample, int sort,
                    int64_t *idx, float *dist2, void *workspace, size_t workspace_bytes, hipStream_t s)
{
    const int G = grid_edge_for(Lmax);
    if (G < 2) return TP3D_E_TOOBIG;
    GridWorkspace w = carve_grid_workspace(workspace, num_clouds, rows, G);
    if (workspace_bytes < w.bytes) return TP3D_E_BADARG;
    const size_t lds = (((size_t)G * G * G * 4 + 15) & ~(size_t)15) + (size_t)Lmax * 2;
    static bool attr_set[64] = {false};
    allow_large_dynamic_lds(reinterpret_cast<const void *>(&grid_build_kernel), (int)GRID_LDS_BUDGET, attr_set);
    hipLaunchKernelGGL(grid_build_kernel, dim3(num_clouds), dim3(GB_BLOCK), lds, s, x, seg, N, radius, G, w.info,
                       w.cell_start, w.sorted_id, w.sorted_xyz);
    if (int rc = check_launch()) return rc;
    const int64_t blocks = (total_q + GQ_BLOCK / 64 - 1) / (GQ_BLOCK / 64);
    if (blocks > 0x7fffffff) return TP3D_E_TOOBIG;
    hipLaunchKernelGGL(grid_query_kernel, dim3((unsigned)blocks), dim3(GQ_BLOCK), 0, s, x, y, seg, batch_y, total_q, N,
                       np, num_clouds, radius * radius, nsample, sort, G, w.info, w.cell_start, w.sorted_id,
                       w.sorted_xyz, idx, dist2);
    return check_launch();
}

}  // namespace tp3d

TP3D_EXPORT size_t tp3d_ball_query_workspace_bytes(int num_clouds, int64_t rows, int max_cloud_points)
{
    if (num_clouds <= 0 || rows < 0 || max_cloud_points <= 0) return 0;
    const int G = tp3d::grid_edge_for(max_cloud_points);
    if (G < 2) return 0;
    return tp3d::carve_grid_workspace(nullptr, num_clouds, rows, G).bytes;
}
