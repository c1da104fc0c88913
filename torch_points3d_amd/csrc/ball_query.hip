// ball_query (dense + partial_dense) -- radius neighbour search with ordered compaction.
//
// Reference contract: torch_points3d/core/spatial_ops/neighbour_finder.py:164 (dense, [0] used),
// :31-37 (partial_dense), core/losses/dirichlet_loss.py:52 (sort=True); semantics SURVEY.md 8a-H2/H3;
// oracle tpk_ref_ball_query_dense_f32 / tpk_ref_ball_query_partial_dense_f32.
//
// Unsorted (hot) path: a wave owns QPW queries; the support cloud streams through an LDS tile in
// struct-of-arrays form; every lane tests one support point per step against the wave's queries and a
// 64-bit __ballot + mbcnt prefix turns the hit mask into output slots, so hits land in ascending index
// order with no sort and no atomics.  A query that has its nsample hits stops testing; a workgroup whose
// queries are all full stops streaming.
#include "grid.h"

namespace tp3d {

constexpr int BQ_BLOCK = 256;              // 4 waves
constexpr int BQ_QPW = 4;                  // queries per wave
constexpr int BQ_QPB = BQ_QPW * BQ_BLOCK / kWave;  // 16 queries per workgroup
constexpr int BQ_TILE = 1024;              // support points per LDS tile (12 KiB)

__global__ __launch_bounds__(BQ_BLOCK) void ball_query_dense_kernel(const float *__restrict__ x,
                                                                     const float *__restrict__ y, int N, int np,
                                                                     float r2, int nsample,
                                                                     int64_t *__restrict__ idx,
                                                                     float *__restrict__ dist2)
{
    __shared__ float sx[BQ_TILE], sy[BQ_TILE], sz[BQ_TILE];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const float *xb = x + (size_t)b * N * 3;
    const int q0 = blockIdx.x * BQ_QPB + wave * BQ_QPW;

    float qx[BQ_QPW], qy[BQ_QPW], qz[BQ_QPW];
    int cnt[BQ_QPW], first[BQ_QPW];
    bool live[BQ_QPW];
#pragma unroll
    for (int q = 0; q < BQ_QPW; ++q) {
        live[q] = (q0 + q) < np;
        const float *yq = y + ((size_t)b * np + (live[q] ? q0 + q : 0)) * 3;
        qx[q] = yq[0];
        qy[q] = yq[1];
        qz[q] = yq[2];
        cnt[q] = live[q] ? 0 : nsample;  // queries past np count as full
        first[q] = 0;
    }

    for (int base = 0; base < N; base += BQ_TILE) {
        const int tcnt = min(BQ_TILE, N - base);
        for (int e = tid; e < tcnt * 3; e += BQ_BLOCK) {
            float v = xb[(size_t)base * 3 + e];
            int pnt = e / 3, c = e - pnt * 3;
            (c == 0 ? sx : (c == 1 ? sy : sz))[pnt] = v;
        }
        __syncthreads();
        bool wave_done = true;
#pragma unroll
        for (int q = 0; q < BQ_QPW; ++q) wave_done = wave_done && (cnt[q] >= nsample);
        if (!wave_done) {
            for (int st = 0; st < tcnt; st += kWave) {
                const int pnt = st + lane;
                const bool valid = pnt < tcnt;
                const float px = valid ? sx[pnt] : 0.0f;
                const float py = valid ? sy[pnt] : 0.0f;
                const float pz = valid ? sz[pnt] : 0.0f;
#pragma unroll
                for (int q = 0; q < BQ_QPW; ++q) {
                    if (cnt[q] < nsample) {  // wave-uniform
                        const float d = sqdist3(px, py, pz, qx[q], qy[q], qz[q]);
                        const bool hit = valid && d < r2;
                        const unsigned long long mask = __ballot(hit);
                        if (mask) {
                            if (cnt[q] == 0) first[q] = base + st + __builtin_ctzll(mask);
                            const int slot = cnt[q] + lanes_below(mask);
                            if (hit && slot < nsample) {
                                const size_t o = ((size_t)b * np + q0 + q) * nsample + slot;
                                idx[o] = base + pnt;
                                dist2[o] = d;
                            }
                            cnt[q] += __builtin_popcountll(mask);
                        }
                    }
                }
            }
        }
        bool done = true;
#pragma unroll
        for (int q = 0; q < BQ_QPW; ++q) done = done && (cnt[q] >= nsample);
        if (__syncthreads_and(done)) break;  // also fences the tile before it is overwritten
    }

    // padding: repeat the first hit (0 if the ball is empty), dist2 = -1
#pragma unroll
    for (int q = 0; q < BQ_QPW; ++q) {
        if (live[q]) {
            const int c = min(cnt[q], nsample);
            const size_t o = ((size_t)b * np + q0 + q) * nsample;
            for (int s = c + lane; s < nsample; s += kWave) {
                idx[o + s] = first[q];
                dist2[o + s] = -1.0f;
            }
        }
    }
}

// sort=True (rare path: reference core/losses/dirichlet_loss.py:52, nsample = 32): one thread per query
// keeps the nsample closest hits by insertion directly in its own output row (rows are thread-private).
__global__ void ball_query_sorted_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                         const int64_t *__restrict__ batch_x, const int64_t *__restrict__ batch_y,
                                         int64_t M, int64_t total_q, int N, int np, float r2, int nsample,
                                         int64_t pad_mode_shadow, int64_t *__restrict__ idx,
                                         float *__restrict__ dist2)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_q) return;
    int64_t lo, hi;
    if (batch_x) {  // partial dense: segment of x with the query's batch id (batch_x ascending)
        const int64_t bq = batch_y[t];
        int64_t a = 0, c = M;
        while (a < c) {
            int64_t m = (a + c) >> 1;
            if (batch_x[m] < bq) a = m + 1; else c = m;
        }
        lo = a;
        c = M;
        while (a < c) {
            int64_t m = (a + c) >> 1;
            if (batch_x[m] <= bq) a = m + 1; else c = m;
        }
        hi = a;
    } else {
        const int64_t b = t / np;
        lo = b * N;
        hi = lo + N;
    }
    const int64_t off = batch_x ? 0 : lo;  // dense indices are cloud-local
    const float qx = y[t * 3 + 0], qy = y[t * 3 + 1], qz = y[t * 3 + 2];
    int64_t *io = idx + t * nsample;
    float *dd = dist2 + t * nsample;
    int cnt = 0;
    for (int64_t k = lo; k < hi; ++k) {
        const float d = sqdist3(x[k * 3 + 0], x[k * 3 + 1], x[k * 3 + 2], qx, qy, qz);
        if (d < r2) {
            int n = cnt;
            if (n == nsample) {
                if (!(d < dd[n - 1])) continue;
                n = n - 1;
            }
            int pos = n;
            while (pos > 0 && d < dd[pos - 1]) {
                dd[pos] = dd[pos - 1];
                io[pos] = io[pos - 1];
                --pos;
            }
            dd[pos] = d;
            io[pos] = k - off;
            cnt = n + 1;
        }
    }
    const int64_t pad = pad_mode_shadow ? -1 : (cnt > 0 ? io[0] : 0);
    for (int s = cnt; s < nsample; ++s) {
        io[s] = pad;
        dd[s] = -1.0f;
    }
}

// partial_dense, unsorted: one wave per query, lanes stride over the query's cloud segment.
__global__ __launch_bounds__(BQ_BLOCK) void ball_query_partial_kernel(
    const float *__restrict__ x, const float *__restrict__ y, const int64_t *__restrict__ batch_x,
    const int64_t *__restrict__ batch_y, int64_t M, int64_t Nq, float r2, int nsample, int64_t *__restrict__ idx,
    float *__restrict__ dist2)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t j = (int64_t)blockIdx.x * (BQ_BLOCK / kWave) + threadIdx.x / kWave;
    if (j >= Nq) return;  // wave-uniform
    const int64_t bq = batch_y[j];
    int64_t a = 0, c = M;
    while (a < c) {
        int64_t m = (a + c) >> 1;
        if (batch_x[m] < bq) a = m + 1; else c = m;
    }
    const int64_t lo = a;
    c = M;
    while (a < c) {
        int64_t m = (a + c) >> 1;
        if (batch_x[m] <= bq) a = m + 1; else c = m;
    }
    const int64_t hi = a;
    const float qx = y[j * 3 + 0], qy = y[j * 3 + 1], qz = y[j * 3 + 2];
    int64_t *io = idx + j * nsample;
    float *dd = dist2 + j * nsample;
    int cnt = 0;
    for (int64_t st = lo; st < hi && cnt < nsample; st += kWave) {
        const int64_t k = st + lane;
        const bool valid = k < hi;
        const int64_t kk = valid ? k : lo;
        const float d = sqdist3(x[kk * 3 + 0], x[kk * 3 + 1], x[kk * 3 + 2], qx, qy, qz);
        const bool hit = valid && d < r2;
        const unsigned long long mask = __ballot(hit);
        if (mask) {
            const int slot = cnt + lanes_below(mask);
            if (hit && slot < nsample) {
                io[slot] = k;
                dd[slot] = d;
            }
            cnt += __builtin_popcountll(mask);
        }
    }
    for (int s = min(cnt, nsample) + lane; s < nsample; s += kWave) {
        io[s] = -1;
        dd[s] = -1.0f;
    }
}

}  // namespace tp3d

// Clouds at least this large go through the uniform grid (csrc/grid.hip) when a workspace is supplied; below it
// the whole cloud fits a few LDS tiles and the brute-force kernels win.
constexpr int BQ_GRID_MIN_POINTS = 2048;

TP3D_EXPORT int tp3d_ball_query_dense_f32(const float *x, const float *y, int B, int N, int np, float radius,
                                          int nsample, int sort, int64_t *idx, float *dist2, void *workspace,
                                          size_t workspace_bytes, void *stream)
{
    using namespace tp3d;
    if (B < 0 || N < 0 || np < 0 || nsample <= 0) return TP3D_E_BADARG;
    if (B == 0 || np == 0) return TP3D_OK;
    if (!y || !idx || !dist2 || (N > 0 && !x)) return TP3D_E_BADARG;
    if ((int64_t)N * 3 > INT32_MAX || B > 65535) return TP3D_E_TOOBIG;
    const float r2 = radius * radius;
    hipStream_t s = (hipStream_t)stream;
    if (workspace && N >= BQ_GRID_MIN_POINTS && grid_plan(N).G >= 2)
        return grid_ball_query(x, y, nullptr, nullptr, B, (int64_t)B * N, N, np, (int64_t)B * np, N, radius, nsample,
                               sort, idx, dist2, workspace, workspace_bytes, false, s);
    if (sort) {
        const int64_t total = (int64_t)B * np;
        const int block = 64;
        hipLaunchKernelGGL(ball_query_sorted_kernel, dim3((unsigned)((total + block - 1) / block)), dim3(block), 0,
                           s, x, y, (const int64_t *)nullptr, (const int64_t *)nullptr, (int64_t)0, total, N, np, r2,
                           nsample, (int64_t)0, idx, dist2);
    } else {
        dim3 grid((np + BQ_QPB - 1) / BQ_QPB, B);
        hipLaunchKernelGGL(ball_query_dense_kernel, grid, dim3(BQ_BLOCK), 0, s, x, y, N, np, r2, nsample, idx,
                           dist2);
    }
    return check_launch();
}

TP3D_EXPORT int tp3d_ball_query_partial_dense_f32(const float *x, const float *y, const int64_t *batch_x,
                                                  const int64_t *batch_y, int64_t M, int64_t Nq, float radius,
                                                  int nsample, int sort, int64_t *idx, float *dist2,
                                                  const int64_t *seg_x, int num_clouds, int max_cloud_points,
                                                  void *workspace, size_t workspace_bytes, int reuse_grid,
                                                  void *stream)
{
    using namespace tp3d;
    if (M < 0 || Nq < 0 || nsample <= 0) return TP3D_E_BADARG;
    if (Nq == 0) return TP3D_OK;
    if (!y || !batch_y || !idx || !dist2 || (M > 0 && (!x || !batch_x))) return TP3D_E_BADARG;
    const float r2 = radius * radius;
    hipStream_t s = (hipStream_t)stream;
    if (workspace && seg_x && num_clouds > 0 && max_cloud_points >= BQ_GRID_MIN_POINTS &&
        grid_plan(max_cloud_points).G >= 2)
        return grid_ball_query(x, y, seg_x, batch_y, num_clouds, M, 0, 0, Nq, max_cloud_points, radius, nsample, sort,
                               idx, dist2, workspace, workspace_bytes, reuse_grid != 0, s);
    if (sort) {
        const int block = 64;
        // batch_x must be non-null to select the partial-dense branch even when M == 0
        hipLaunchKernelGGL(ball_query_sorted_kernel, dim3((unsigned)((Nq + block - 1) / block)), dim3(block), 0, s,
                           x, y, batch_x ? batch_x : batch_y, batch_y, M, Nq, 0, 0, r2, nsample, (int64_t)1, idx,
                           dist2);
    } else {
        const int qpb = BQ_BLOCK / kWave;
        hipLaunchKernelGGL(ball_query_partial_kernel, dim3((unsigned)((Nq + qpb - 1) / qpb)), dim3(BQ_BLOCK), 0, s,
                           x, y, batch_x ? batch_x : batch_y, batch_y, M, Nq, r2, nsample, idx, dist2);
    }
    return check_launch();
}
