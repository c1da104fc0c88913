// Exact k nearest neighbours over the uniform grid (grid.h): the neighbour search behind KNNInterpolate / FPModule_PD
// (reference core/spatial_ops/interpolate.py:7-69, core/base_conv/partial_dense.py:103-146; k = 1 in the KPConv
// decoders, applications/conf/kpconv/unet_*.yaml `up_k`) and KNNNeighbourFinder (core/spatial_ops/neighbour_finder.py:
// 42-47; k = 16 in RandLA-Net).  The reference takes it from torch_cluster 1.5.9 `knn` (absent from the container);
// semantics here: the k support points of the query's own cloud with the smallest fp32 squared distance
// (dx*dx + dy*dy) + dz*dz, closest first, ties by lower index; slots beyond the cloud's size hold -1.
//
// One wave per query.  Ring R = 1, 2, ... : gather every support point of the (2R+1)^3 cells around the query's cell
// (x-runs of a (z, y) row are contiguous in the cell-ordered copy), select the k smallest (distance, index) pairs,
// and accept once the k-th distance is inside the largest ball that is certainly covered by the visited cells
// (distance to the nearest face of the block that still has cells behind it, shrunk by 0.1 % for the fp32 rounding
// of the cell coordinate).  Exact by construction: any point outside the block is farther than that ball.
// k == 1 never touches LDS (running wave arg-min).  2 <= k <= 64 keeps the 64 best pairs in registers (one per lane)
// and merges batches of 64 admitted candidates with a wave-wide bitonic network; the current k-th best is the
// admission threshold, so over-populated cells (surfaces scanned much denser than the grid assumes) cost a scan of
// their points and little else.  64 < k <= 128 streams through a 1024-slot LDS buffer compacted by successive
// selection; k > 128 falls back to a selection scan of the whole cloud.
#include "grid.h"

namespace tp3d {

constexpr int KQ_BLOCK = 256;  // 4 waves, one query per wave
constexpr int KQ_CAP = 1024;   // candidate slots per wave
constexpr int KQ_KMAX = 128;   // largest k served from the grid (beyond: selection scan of the whole cloud)

__device__ __forceinline__ bool pair_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }

__device__ __forceinline__ void wave_argmin(float &d, int &i)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float od = __shfl_xor(d, off);
        const int oi = __shfl_xor(i, off);
        if (pair_less(od, oi, d, i)) {
            d = od;
            i = oi;
        }
    }
}

// MODE 0: k == 1;  MODE 1: 2 <= k <= 64;  MODE 2: 64 < k (LDS candidate buffer).  The mode only sizes the LDS arrays
// (none / 64 staging slots / 1024 + 128 slots per wave), i.e. how many waves fit a CU to hide the load latencies.
template <int MODE>
__global__ __launch_bounds__(KQ_BLOCK) void grid_knn_kernel(
    const float *__restrict__ x, const float *__restrict__ y, const int64_t *__restrict__ seg,
    const int64_t *__restrict__ batch_y, int64_t total_q, int N, int np, int num_clouds, int k, int G,
    const GridInfo *__restrict__ info, const int *__restrict__ cell_start, const float4 *__restrict__ sorted_pt,
    int64_t *__restrict__ idx, float *__restrict__ dist2)
{
    constexpr int CAP = MODE == 2 ? KQ_CAP : (MODE == 1 ? 64 : 1);
    constexpr int STG = MODE == 2 ? KQ_KMAX : 1;
    __shared__ int s_id[KQ_BLOCK / 64][CAP];
    __shared__ float s_d[KQ_BLOCK / 64][CAP];
    __shared__ int s_si[KQ_BLOCK / 64][STG];
    __shared__ float s_sd[KQ_BLOCK / 64][STG];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * (KQ_BLOCK / 64) + wave;
    if (q >= total_q) return;  // wave-uniform; the kernel has no workgroup barrier
    int *cid = s_id[wave];
    float *cd = s_d[wave];
    int *si = s_si[wave];
    float *sd = s_sd[wave];
    const bool partial = seg != nullptr;
    const int64_t bq = partial ? batch_y[q] : q / np;
    int64_t *io = idx + q * k;
    float *dd = dist2 + q * k;
    int64_t lo = 0;
    int L = 0;
    if (bq >= 0 && bq < num_clouds) {
        lo = partial ? seg[bq] : bq * N;
        L = partial ? (int)(seg[bq + 1] - seg[bq]) : N;
    }
    if (L == 0) {
        for (int s = lane; s < k; s += 64) {
            io[s] = -1;
            dd[s] = -1.0f;
        }
        return;
    }
    const int64_t goff = partial ? lo : 0;
    const float qx = y[q * 3 + 0], qy = y[q * 3 + 1], qz = y[q * 3 + 2];
    const GridInfo gi = info[bq];
    const int *cs = cell_start + (size_t)bq * ((size_t)G * G * G + 1);
    const float cell = 1.0f / gi.inv_cs;
    // query position in cell units (same fp32 expression as the builders), clamped far outside the grid
    const float lim = 1.0e6f;
    const float ux = fminf(fmaxf((qx - gi.minx) * gi.inv_cs, -lim), lim);
    const float uy = fminf(fmaxf((qy - gi.miny) * gi.inv_cs, -lim), lim);
    const float uz = fminf(fmaxf((qz - gi.minz) * gi.inv_cs, -lim), lim);
    const int cx = (int)floorf(ux), cy = (int)floorf(uy), cz = (int)floorf(uz);
    // smallest ring whose block reaches the grid at all
    int R = max(1, max(max(max(-cx, cx - (gi.gx - 1)), max(-cy, cy - (gi.gy - 1))), max(-cz, cz - (gi.gz - 1))));
    const int kk = min(k, L);
    float tau = 3.0e38f;  // upper bound of the k-th distance once kk candidates are known

    for (; MODE != 2 || k <= KQ_KMAX; ++R) {
        const int x0 = max(cx - R, 0), x1 = min(cx + R, gi.gx - 1);
        const int y0 = max(cy - R, 0), y1 = min(cy + R, gi.gy - 1);
        const int z0 = max(cz - R, 0), z1 = min(cz + R, gi.gz - 1);
        const bool whole = x0 == 0 && y0 == 0 && z0 == 0 && x1 == gi.gx - 1 && y1 == gi.gy - 1 && z1 == gi.gz - 1;
        // radius of the ball around the query that the block certainly covers (faces with cells behind them only)
        float cover = 3.0e38f;
        if (x0 > 0) cover = fminf(cover, ux - (float)(cx - R));
        if (x1 < gi.gx - 1) cover = fminf(cover, (float)(cx + R + 1) - ux);
        if (y0 > 0) cover = fminf(cover, uy - (float)(cy - R));
        if (y1 < gi.gy - 1) cover = fminf(cover, (float)(cy + R + 1) - uy);
        if (z0 > 0) cover = fminf(cover, uz - (float)(cz - R));
        if (z1 < gi.gz - 1) cover = fminf(cover, (float)(cz + R + 1) - uz);
        cover = cover * cell * 0.999f;
        const float cover2 = cover * cover;
        // The block's (z, y) rows are contiguous x-runs of the cell-ordered copy.  Their slot ranges are fetched 64
        // rows at a time, one row per lane, and handed out by v_readlane: one load latency per ring instead of one
        // per row.
        auto for_each_run = [&](auto &&body) {
            const int ny = y1 - y0 + 1, nrows = (z1 - z0 + 1) * ny;
            for (int r0 = 0; r0 < nrows; r0 += 64) {
                const int r = r0 + lane;
                int j0v = 0, j1v = 0;
                if (r < nrows) {
                    const int rowbase = ((z0 + r / ny) * gi.gy + (y0 + r % ny)) * gi.gx;
                    j0v = cs[rowbase + x0];
                    j1v = cs[rowbase + x1 + 1];
                }
                const int cnt = min(64, nrows - r0);
                for (int rr = 0; rr < cnt; ++rr)
                    body(__builtin_amdgcn_readlane(j0v, rr), __builtin_amdgcn_readlane(j1v, rr));
            }
        };

        if (MODE == 0) {
            float bd = 3.0e38f;
            int bi = 0x7fffffff;
            for_each_run([&](int j0, int j1) {
                for (int j = j0 + lane; j < j1; j += 64) {
                    const float4 pt = sorted_pt[lo + j];
                    const float d = sqdist3(pt.x, pt.y, pt.z, qx, qy, qz);
                    const int id = __float_as_int(pt.w);
                    if (pair_less(d, id, bd, bi)) {
                        bd = d;
                        bi = id;
                    }
                }
            });
            wave_argmin(bd, bi);
            const bool found = bi != 0x7fffffff;
            if (whole || (found && bd <= cover2)) {  // `whole` always ends the search (found unless the input is NaN)
                if (lane == 0) {
                    io[0] = found ? goff + bi : -1;
                    dd[0] = found ? bd : -1.0f;
                }
                return;
            }
            continue;
        }

        if (MODE == 1) {
            // ---- 2 <= k <= 64: the best 64 pairs live in registers, one per lane, ascending.  Candidates that beat
            // the current k-th best (`tau`) are staged 64 at a time in LDS, bitonic-sorted across the wave and merged:
            // min(best[l], staged[63 - l]) holds the 64 smallest of the union as a bitonic sequence, six more
            // compare-exchange stages sort it.  After the first batch few points pass `tau`, so a query costs the
            // scan of its cells plus two or three merges.
            float bd = 3.0e38f;   // lane l: l-th best so far
            int bi = 0x7fffffff;
            int staged = 0, total = 0;
            auto exchange = [&](float &d, int &i, int stride, bool up) {
                const float od = __shfl_xor(d, stride);
                const int oi = __shfl_xor(i, stride);
                const bool lower = (lane & stride) == 0;           // this lane keeps the smaller pair when `up`
                const bool take = pair_less(od, oi, d, i) == (lower == up);
                if (take) {
                    d = od;
                    i = oi;
                }
            };
            auto flush = [&]() {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                float sdv = lane < staged ? cd[lane] : 3.0e38f;
                int siv = lane < staged ? cid[lane] : 0x7fffffff;
#pragma unroll
                for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
                    for (int stride = size >> 1; stride >= 1; stride >>= 1)
                        exchange(sdv, siv, stride, (lane & size) == 0);  // size == 64: every lane ascending
                const float rd = __shfl(sdv, 63 - lane);
                const int ri = __shfl(siv, 63 - lane);
                if (pair_less(rd, ri, bd, bi)) {
                    bd = rd;
                    bi = ri;
                }
#pragma unroll
                for (int stride = 32; stride >= 1; stride >>= 1) exchange(bd, bi, stride, true);
                total = min(total + staged, 64);
                staged = 0;
                if (total >= kk) tau = __shfl(bd, kk - 1);
                __builtin_amdgcn_wave_barrier();
            };
            for_each_run([&](int j0, int j1) {
                for (int j = j0; j < j1; j += 64) {
                    const int t = j + lane;
                    const bool valid = t < j1;
                    const int tt = valid ? t : j0;
                    const float4 pt = sorted_pt[lo + tt];
                    const float d = sqdist3(pt.x, pt.y, pt.z, qx, qy, qz);
                    const int id = __float_as_int(pt.w);
                    bool keep = valid && d <= tau;
                    unsigned long long mask = __ballot(keep);
                    if (!mask) continue;
                    if (staged + __builtin_popcountll(mask) > 64) {
                        flush();
                        keep = keep && d <= tau;
                        mask = __ballot(keep);
                        if (!mask) continue;
                    }
                    if (keep) {
                        const int slot = staged + lanes_below(mask);
                        cd[slot] = d;
                        cid[slot] = id;
                    }
                    staged += __builtin_popcountll(mask);
                }
            });
            if (staged) flush();
            if (total < kk && !whole) continue;
            const int emit = min(total, kk);
            const float worst = emit ? __shfl(bd, emit - 1) : 0.0f;
            if (!whole && !(worst <= cover2)) continue;  // the k-th neighbour may still lie outside the block: widen
            if (lane < emit) {
                io[lane] = goff + bi;
                dd[lane] = bd;
            } else if (lane < k) {
                io[lane] = -1;
                dd[lane] = -1.0f;
            }
            return;
        }

        // ---- k > 1: stream the block's points through the candidate buffer.  Once the buffer would overflow it is
        // compacted to its kk smallest pairs, whose largest distance `tau` bounds the final k-th distance from above:
        // from then on only points with d <= tau are appended, so dense cells cost a scan but never an overflow.
        int h = 0;
        auto compact = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            float pd = -1.0f;
            int pi = -1;
            const int emit = min(kk, h);
            for (int sidx = 0; sidx < emit; ++sidx) {  // pass s: the smallest pair after the previous one
                float bd = 3.0e38f;
                int bi = 0x7fffffff;
                for (int t = lane; t < h; t += 64) {
                    const float d = cd[t];
                    const int id = cid[t];
                    if ((d > pd || (d == pd && id > pi)) && pair_less(d, id, bd, bi)) {
                        bd = d;
                        bi = id;
                    }
                }
                wave_argmin(bd, bi);
                if (lane == 0) {
                    sd[sidx] = bd;
                    si[sidx] = bi;
                }
                pd = bd;
                pi = bi;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < emit; t += 64) {
                cd[t] = sd[t];
                cid[t] = si[t];
            }
            h = emit;
            if (emit == kk) tau = pd;  // kk real candidates exist within tau
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
        };
        for_each_run([&](int j0, int j1) {
            for (int j = j0; j < j1; j += 64) {
                const int t = j + lane;
                const bool valid = t < j1;
                const int tt = valid ? t : j0;
                const float4 pt = sorted_pt[lo + tt];
                const float d = sqdist3(pt.x, pt.y, pt.z, qx, qy, qz);
                const int id = __float_as_int(pt.w);
                bool keep = valid && d <= tau;
                unsigned long long mask = __ballot(keep);
                if (!mask) continue;
                if (h + __builtin_popcountll(mask) > KQ_CAP) {
                    compact();
                    keep = keep && d <= tau;
                    mask = __ballot(keep);
                    if (!mask) continue;
                }
                if (keep) {
                    const int slot = h + lanes_below(mask);
                    cd[slot] = d;
                    cid[slot] = id;
                }
                h += __builtin_popcountll(mask);
            }
        });
        if (h < kk && !whole) continue;
        compact();  // the kk (or all h) smallest pairs, sorted, at the front of the buffer
        const int emit = h;
        const float worst = emit ? cd[emit - 1] : 0.0f;
        if (!whole && !(worst <= cover2)) continue;  // the k-th neighbour may still lie outside the block: widen
        for (int sidx = lane; sidx < emit; sidx += 64) {
            io[sidx] = goff + cid[sidx];
            dd[sidx] = cd[sidx];
        }
        for (int sidx = emit + lane; sidx < k; sidx += 64) {
            io[sidx] = -1;
            dd[sidx] = -1.0f;
        }
        return;
    }

    // ---- k beyond the grid path's staging size: successive selection over the whole cloud
    float pd = -1.0f;
    int pi = -1;
    for (int s = 0; s < kk; ++s) {
        float bd = 3.0e38f;
        int bi = 0x7fffffff;
        for (int j = lane; j < L; j += 64) {
            const float d = sqdist3(x[(lo + j) * 3 + 0], x[(lo + j) * 3 + 1], x[(lo + j) * 3 + 2], qx, qy, qz);
            if ((d > pd || (d == pd && j > pi)) && pair_less(d, j, bd, bi)) {
                bd = d;
                bi = j;
            }
        }
        wave_argmin(bd, bi);
        if (lane == 0) {
            io[s] = goff + bi;
            dd[s] = bd;
        }
        pd = bd;
        pi = bi;
    }
    for (int s = kk + lane; s < k; s += 64) {
        io[s] = -1;
        dd[s] = -1.0f;
    }
}

// knn_interpolate (torch_geometric 1.7.2 nn/unpool/knn_interpolate.py, called at core/spatial_ops/interpolate.py:69) fused
// with FPModule_PD's skip concatenation (core/base_conv/partial_dense.py:139-140):
//   w_j = 1 / max(d2_j, 1e-16);  out[i, 0:C] = (sum_j x[idx[i,j]] * w_j) / (sum_j w_j), both sums in slot order;
//   out[i, C:C+C2] = skip[i];  columns up to ld zero.  wnorm[i,j] = w_j / sum_j w_j is kept for the backward pass.
__global__ __launch_bounds__(256) void knn_interpolate_kernel(const float *__restrict__ x, const int64_t *__restrict__ idx,
                                                               const float *__restrict__ dist2,
                                                               const float *__restrict__ skip, int64_t Nq, int k, int C,
                                                               int C2, int ld, float *__restrict__ out,
                                                               float *__restrict__ wnorm)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= Nq * ld) return;
    const int64_t i = t / ld;
    const int c = (int)(t - i * ld);
    float v = 0.0f;
    if (c < C) {
        float num = 0.0f, den = 0.0f;
        for (int j = 0; j < k; ++j) {
            const int64_t m = idx[i * k + j];
            if (m < 0) continue;
            const float w = 1.0f / fmaxf(dist2[i * k + j], 1.0e-16f);
            num += x[m * C + c] * w;
            den += w;
        }
        v = num / den;  // den == 0 only when the query's cloud has no support point: 0/0 = NaN, as in the reference
        if (c == 0 && wnorm)
            for (int j = 0; j < k; ++j) {
                const bool real = idx[i * k + j] >= 0;
                wnorm[i * k + j] = real ? (1.0f / fmaxf(dist2[i * k + j], 1.0e-16f)) / den : 0.0f;
            }
    } else if (c < C + C2) {
        v = skip[i * C2 + (c - C)];
    }
    out[t] = v;
}

int grid_knn(const float *x, const float *y, const int64_t *seg, const int64_t *batch_y, int num_clouds, int64_t rows,
             int N, int np, int64_t total_q, int Lmax, int k, float cell, int64_t *idx, float *dist2, void *workspace,
             size_t workspace_bytes, hipStream_t s)
{
    const GridPlan plan = grid_plan(Lmax);
    if (plan.G < 2) return TP3D_E_TOOBIG;
    GridWorkspace w = carve_grid_workspace(workspace, num_clouds, rows, plan);
    if (workspace_bytes < w.bytes) return TP3D_E_BADARG;
    // automatic cell edge: about k/4 points per cell, so the 27-cell block usually holds the k-th neighbour
    const float target = fmaxf(2.0f, (float)k * 0.25f);
    if (int rc = grid_build(x, seg, num_clouds, rows, N, Lmax, cell, target, plan, w, s)) return rc;
    const int64_t blocks = (total_q + KQ_BLOCK / 64 - 1) / (KQ_BLOCK / 64);
    if (blocks > 0x7fffffff) return TP3D_E_TOOBIG;
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(KQ_BLOCK), 0, s, x, y, seg, batch_y, total_q, N, np,
                           num_clouds, k, plan.G, w.info, w.cell_start, w.sorted_pt, idx, dist2);
    };
    if (k == 1) launch(grid_knn_kernel<0>);
    else if (k <= 64) launch(grid_knn_kernel<1>);
    else launch(grid_knn_kernel<2>);
    return check_launch();
}

}  // namespace tp3d

using namespace tp3d;

TP3D_EXPORT size_t tp3d_knn_workspace_bytes(int num_clouds, int64_t rows, int max_cloud_points)
{
    if (num_clouds <= 0 || rows < 0 || max_cloud_points <= 0) return 0;
    const GridPlan plan = grid_plan(max_cloud_points);
    if (plan.G < 2) return 0;
    return carve_grid_workspace(nullptr, num_clouds, rows, plan).bytes;
}

TP3D_EXPORT int tp3d_knn_partial_dense_f32(const float *x, const float *y, const int64_t *batch_y, const int64_t *seg_x,
                                           int num_clouds, int max_cloud_points, int64_t M, int64_t Nq, int k, float cell,
                                           int64_t *idx, float *dist2, void *workspace, size_t workspace_bytes,
                                           void *stream)
{
    if (M < 0 || Nq < 0 || k <= 0 || num_clouds <= 0 || max_cloud_points < 0) return TP3D_E_BADARG;
    if (Nq == 0) return TP3D_OK;
    if (!y || !batch_y || !seg_x || !idx || !dist2 || !workspace) return TP3D_E_BADARG;
    if (M > 0 && !x) return TP3D_E_BADARG;
    if (max_cloud_points == 0) max_cloud_points = 1;  // every cloud empty: the kernel only writes the -1 padding
    return grid_knn(x, y, seg_x, batch_y, num_clouds, M, 0, 0, Nq, max_cloud_points, k, cell, idx, dist2, workspace,
                    workspace_bytes, (hipStream_t)stream);
}

TP3D_EXPORT int tp3d_knn_dense_f32(const float *x, const float *y, int B, int N, int np, int k, float cell, int64_t *idx,
                                   float *dist2, void *workspace, size_t workspace_bytes, void *stream)
{
    if (B < 0 || N <= 0 || np < 0 || k <= 0) return TP3D_E_BADARG;
    if (B == 0 || np == 0) return TP3D_OK;
    if (!x || !y || !idx || !dist2 || !workspace) return TP3D_E_BADARG;
    return grid_knn(x, y, nullptr, nullptr, B, (int64_t)B * N, N, np, (int64_t)B * np, N, k, cell, idx, dist2, workspace,
                    workspace_bytes, (hipStream_t)stream);
}

TP3D_EXPORT int tp3d_knn_interpolate_fwd_f32(const float *x, const int64_t *idx, const float *dist2, const float *skip,
                                             int64_t Nq, int k, int C, int C2, int ld, float *out, float *wnorm,
                                             void *stream)
{
    if (Nq < 0 || k <= 0 || C <= 0 || C2 < 0 || ld < C + C2) return TP3D_E_BADARG;
    if (Nq == 0) return TP3D_OK;
    if (!x || !idx || !dist2 || !out || (C2 > 0 && !skip)) return TP3D_E_BADARG;
    const int64_t blocks = (Nq * ld + 255) / 256;
    if (blocks > 0x7fffffff) return TP3D_E_TOOBIG;
    hipLaunchKernelGGL(knn_interpolate_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, idx, dist2,
                       skip, Nq, k, C, C2, ld, out, wnorm);
    return check_launch();
}
