// Voxel clustering of a partial-dense cloud: the device form of GridSampling3D
// (reference torch_points3d/core/data_transform/grid_transform.py:33-141) -- the sampler of every strided KPConv block
// (modules/KPConv/blocks.py:60-61,79).
//
// Reference pipeline (all third-party, none of it in the container: torch_cluster 1.5.9 grid_cluster,
// torch_geometric 1.7.2 voxel_grid / consecutive_cluster, torch_scatter 2.0.8 scatter_mean / scatter_add):
//     coords  = round(pos / size)                                   half-to-even, fp32 true division
//     key     = sum_d (coords_d - min_d) * stride_d,  d = x, y, z, batch;  stride = running product of extents
//     cluster, perm = unique(key, sorted, inverse);  perm[c] = LAST point index of cluster c
//     mean mode: scatter_mean in ascending point order;  labels: one-hot scatter_add + argmax (ties -> lowest label)
// Keys order lexicographically by (batch, z, y, x), so cluster ids are the ranks of the occupied voxels in that order.
//
// Device form: one stable LSD radix sort (rocPRIM) of (key, point index) over exactly the bits the key needs, boundary
// flags + inclusive scan -> consecutive ids, one scatter pass.  Inside a cluster the sorted order IS ascending point
// index (stable sort of iota), which is what makes the fp32 means reproduce a sequential scatter_add bit for bit.
// The host reads back 7 ints (bounding box of coords) before the sort and 1 int64 (cluster count) after it: the same
// two device->host waits torch.unique costs the reference.
#include <cstring>  // rocPRIM's texture iterator calls memset unqualified

#include <rocprim/rocprim.hpp>

#include "grid.h"

namespace tp3d {

constexpr int VX_BLOCK = 256;
constexpr int VX_COORD_LIMIT = 1 << 24;  // beyond this fp32 coordinates stop being exact integers (reference breaks too)

__device__ __forceinline__ int voxel_coord(float p, float size)
{
    const float c = rintf(p / size);  // v_rndne_f32 == torch.round; '/' is IEEE (hipcc's default correctly-rounded divide)
    return (int)fminf(fmaxf(c, -(float)VX_COORD_LIMIT), (float)VX_COORD_LIMIT);
}

__global__ void voxel_bounds_init_kernel(int *bounds)
{
    const int t = threadIdx.x;
    if (t < 3) bounds[t] = 0x7fffffff;
    else if (t < 7) bounds[t] = (int)0x80000000;
    else if (t == 7) bounds[t] = 0;
}

// bounds = [min x,y,z | max x,y,z | max batch | bad-input flag]
__global__ __launch_bounds__(VX_BLOCK) void voxel_bounds_kernel(const float *__restrict__ pos,
                                                                 const int64_t *__restrict__ batch, int64_t N, float size,
                                                                 int *__restrict__ bounds)
{
    __shared__ int s_red[8][VX_BLOCK / 64];
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    int mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    int mb = (int)0x80000000, bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * VX_BLOCK + threadIdx.x; i < N; i += (int64_t)gridDim.x * VX_BLOCK) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int c = voxel_coord(pos[i * 3 + a], size);
            mn[a] = min(mn[a], c);
            mx[a] = max(mx[a], c);
            bad |= (c <= -VX_COORD_LIMIT || c >= VX_COORD_LIMIT) ? 1 : 0;
        }
        if (batch) {
            const int64_t b = batch[i];
            bad |= (b < 0 || b >= (1 << 30)) ? 1 : 0;
            mb = max(mb, (int)b);
        }
    }
    int v[8] = {mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], mb, bad};
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int o = __shfl_xor(v[k], off);
            v[k] = k < 3 ? min(v[k], o) : max(v[k], o);
        }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 8; ++k) s_red[k][wave] = v[k];
    __syncthreads();
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        int r = s_red[k][0];
        for (int w = 1; w < VX_BLOCK / 64; ++w) r = k < 3 ? min(r, s_red[k][w]) : max(r, s_red[k][w]);
        if (k < 3) atomicMin(&bounds[k], r);
        else atomicMax(&bounds[k], r);
    }
}

__global__ __launch_bounds__(VX_BLOCK) void voxel_key_kernel(const float *__restrict__ pos, const int64_t *__restrict__ batch,
                                                              int64_t N, float size, int minx, int miny, int minz,
                                                              int64_t ex, int64_t ey, int64_t ez,
                                                              unsigned long long *__restrict__ keys,
                                                              unsigned int *__restrict__ vals)
{
    const int64_t i = (int64_t)blockIdx.x * VX_BLOCK + threadIdx.x;
    if (i >= N) return;
    const int64_t cx = voxel_coord(pos[i * 3 + 0], size) - minx;
    const int64_t cy = voxel_coord(pos[i * 3 + 1], size) - miny;
    const int64_t cz = voxel_coord(pos[i * 3 + 2], size) - minz;
    const int64_t b = batch ? batch[i] : 0;
    keys[i] = (unsigned long long)(((b * ez + cz) * ey + cy) * ex + cx);
    vals[i] = (unsigned int)i;
}

__global__ __launch_bounds__(VX_BLOCK) void voxel_flag_kernel(const unsigned long long *__restrict__ keys, int64_t N,
                                                               int *__restrict__ flags)
{
    const int64_t i = (int64_t)blockIdx.x * VX_BLOCK + threadIdx.x;
    if (i >= N) return;
    flags[i] = (i > 0 && keys[i] != keys[i - 1]) ? 1 : 0;
}

// cid[i] = cluster of sorted slot i (inclusive scan of the boundary flags)
__global__ __launch_bounds__(VX_BLOCK) void voxel_scatter_kernel(const unsigned int *__restrict__ vals,
                                                                  const int *__restrict__ cid, int64_t N,
                                                                  int64_t *__restrict__ cluster, int64_t *__restrict__ order,
                                                                  int64_t *__restrict__ cluster_start,
                                                                  int64_t *__restrict__ last,
                                                                  const int64_t *__restrict__ batch, int64_t nb,
                                                                  unsigned long long *__restrict__ meta)
{
    const int64_t i = (int64_t)blockIdx.x * VX_BLOCK + threadIdx.x;
    if (i >= N) return;
    const int64_t p = vals[i];
    const int c = cid[i];
    cluster[p] = c;
    order[i] = p;
    if (i == 0 || cid[i - 1] != c) cluster_start[c] = i;
    if (i == N - 1 || cid[i + 1] != c) last[c] = p;  // highest point index of the voxel (stable sort)
    // clusters are numbered cloud by cloud: the last slot of cloud b records how many clusters clouds 0..b hold
    const int64_t b = batch ? batch[p] : 0;
    // (b < nb: the caller's extent may be a hint that the host verifies afterwards; never write past meta)
    if ((i == N - 1 || (batch && batch[vals[i + 1]] != b)) && b >= 0 && b < nb) meta[1 + b] = (unsigned long long)c + 1;
    if (i == N - 1) {
        cluster_start[c + 1] = N;
        meta[0] = (unsigned long long)c + 1;
    }
}

// out[c][ch] = (sum over the members of c, ascending point index, of x[p][ch]) / count      (scatter_mean)
__global__ __launch_bounds__(VX_BLOCK) void cluster_mean_kernel(const float *__restrict__ x, const int64_t *__restrict__ order,
                                                                 const int64_t *__restrict__ cluster_start, int64_t K,
                                                                 int C, float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * VX_BLOCK + threadIdx.x;
    if (t >= K * C) return;
    const int64_t c = t / C;
    const int ch = (int)(t - c * C);
    const int64_t j0 = cluster_start[c], j1 = cluster_start[c + 1];
    float sum = 0.0f;
    for (int64_t j = j0; j < j1; ++j) sum += x[order[j] * C + ch];
    out[t] = sum / (float)(j1 - j0);
}

// Majority label per cluster: argmax over (label - min_label) of the member count, ties -> lowest label
// (torch.argmax over the one-hot sums, grid_transform.py:73-76).  One wave per cluster.
constexpr int VX_HIST = 2048;  // label range handled by the per-wave LDS histogram
__global__ __launch_bounds__(VX_BLOCK) void cluster_majority_kernel(const int64_t *__restrict__ labels,
                                                                     const int64_t *__restrict__ order,
                                                                     const int64_t *__restrict__ cluster_start, int64_t K,
                                                                     int64_t min_label, int num_classes,
                                                                     int64_t *__restrict__ out)
{
    __shared__ int s_hist[VX_BLOCK / 64][VX_HIST];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * (VX_BLOCK / 64) + wave;
    if (c >= K) return;  // wave-uniform; no workgroup barrier below
    const int64_t j0 = cluster_start[c], j1 = cluster_start[c + 1];
    int best_cnt = 0;
    int64_t best_lab = 0x7fffffffffffffffLL;
    if (num_classes <= VX_HIST) {
        int *h = s_hist[wave];
        for (int k = lane; k < num_classes; k += 64) h[k] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int64_t j = j0 + lane; j < j1; j += 64) atomicAdd(&h[(int)(labels[order[j]] - min_label)], 1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < num_classes; k += 64) {
            const int n = h[k];
            if (n > best_cnt) {  // ascending k per lane: the first maximum is kept
                best_cnt = n;
                best_lab = k;
            }
        }
    } else {
        // label range too wide for LDS: count each member's label against all members (rare: instance ids)
        for (int64_t j = j0 + lane; j < j1; j += 64) {
            const int64_t lab = labels[order[j]] - min_label;
            int n = 0;
            for (int64_t u = j0; u < j1; ++u) n += (labels[order[u]] - min_label == lab) ? 1 : 0;
            if (n > best_cnt || (n == best_cnt && lab < best_lab)) {
                best_cnt = n;
                best_lab = lab;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const int oc = __shfl_xor(best_cnt, off);
        const int64_t ol = __shfl_xor(best_lab, off);
        if (oc > best_cnt || (oc == best_cnt && ol < best_lab)) {
            best_cnt = oc;
            best_lab = ol;
        }
    }
    if (lane == 0) out[c] = best_lab + min_label;
}

struct VoxelWorkspace {
    unsigned long long *keys_in, *keys_out;
    unsigned int *vals_in, *vals_out;
    int *flags, *cid;
    void *tmp;
    size_t tmp_bytes, bytes;
};

static int hip_rc(hipError_t e)
{
    if (e == hipSuccess) return TP3D_OK;
    set_last_hip_error(e);
    return TP3D_E_LAUNCH;
}

// The library's one radix-sort instantiation (also used by the sort-based grid build, grid.hip).
size_t sort_pairs_tmp_bytes(int64_t n)
{
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                    (const unsigned int *)nullptr, (unsigned int *)nullptr, (size_t)n, 0u, 64u,
                                    (hipStream_t)0);
    return bytes;
}

int sort_pairs_u64_u32(void *tmp, size_t tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                       const unsigned int *vals_in, unsigned int *vals_out, int64_t n, unsigned bits, hipStream_t s)
{
    size_t tb = tmp_bytes;
    return hip_rc(rocprim::radix_sort_pairs(tmp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, bits, s));
}

static size_t voxel_tmp_bytes(int64_t N)
{
    size_t sort_bytes = sort_pairs_tmp_bytes(N), scan_bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, scan_bytes, (const int *)nullptr, (int *)nullptr, (size_t)N,
                                  rocprim::plus<int>(), (hipStream_t)0);
    return sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
}

static VoxelWorkspace carve_voxel_workspace(void *ws, int64_t N)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    VoxelWorkspace w;
    char *p = static_cast<char *>(ws);
    size_t off = 0;
    w.keys_in = reinterpret_cast<unsigned long long *>(p + off);
    off += up((size_t)N * 8);
    w.keys_out = reinterpret_cast<unsigned long long *>(p + off);
    off += up((size_t)N * 8);
    w.vals_in = reinterpret_cast<unsigned int *>(p + off);
    off += up((size_t)N * 4);
    w.vals_out = reinterpret_cast<unsigned int *>(p + off);
    off += up((size_t)N * 4);
    w.flags = reinterpret_cast<int *>(p + off);
    off += up((size_t)N * 4);
    w.cid = reinterpret_cast<int *>(p + off);
    off += up((size_t)N * 4);
    w.tmp = p + off;
    w.tmp_bytes = voxel_tmp_bytes(N);
    off += up(w.tmp_bytes + 256);
    w.bytes = off;
    return w;
}

}  // namespace tp3d

using namespace tp3d;

TP3D_EXPORT int tp3d_voxel_bounds_f32(const float *pos, const int64_t *batch, int64_t N, float size, int32_t *bounds,
                                      void *stream)
{
    if (N < 0 || N >= 0x7fffffff || !(size > 0.0f) || !bounds) return TP3D_E_BADARG;
    if (N > 0 && !pos) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(voxel_bounds_init_kernel, dim3(1), dim3(64), 0, s, bounds);
    if (int rc = check_launch()) return rc;
    if (N == 0) return TP3D_OK;
    int64_t blocks = (N + VX_BLOCK - 1) / VX_BLOCK;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(voxel_bounds_kernel, dim3((unsigned)blocks), dim3(VX_BLOCK), 0, s, pos, batch, N, size, bounds);
    return check_launch();
}

TP3D_EXPORT size_t tp3d_voxel_workspace_bytes(int64_t N)
{
    if (N <= 0 || N >= 0x7fffffff) return 0;
    return carve_voxel_workspace(nullptr, N).bytes;
}

TP3D_EXPORT int tp3d_voxel_cluster_f32(const float *pos, const int64_t *batch, int64_t N, float size,
                                       const int32_t *bounds_host, int64_t *cluster, int64_t *order,
                                       int64_t *cluster_start, int64_t *last, int64_t *meta, void *workspace,
                                       size_t workspace_bytes, void *stream)
{
    if (N <= 0 || N >= 0x7fffffff || !(size > 0.0f) || !pos || !bounds_host || !cluster || !order || !cluster_start ||
        !last || !meta || !workspace)
        return TP3D_E_BADARG;
    if (bounds_host[7] != 0) return TP3D_E_TOOBIG;  // a coordinate beyond +-2^24 voxels or a batch id out of range
    const int64_t ex = (int64_t)bounds_host[3] - bounds_host[0] + 1;
    const int64_t ey = (int64_t)bounds_host[4] - bounds_host[1] + 1;
    const int64_t ez = (int64_t)bounds_host[5] - bounds_host[2] + 1;
    const int64_t nb = batch ? (int64_t)bounds_host[6] + 1 : 1;
    if (ex <= 0 || ey <= 0 || ez <= 0 || nb <= 0) return TP3D_E_BADARG;
    // number of key bits: extents are < 2^26 each and nb <= 2^30, so use 128-bit products to detect overflow
    const unsigned __int128 total = (unsigned __int128)ex * (unsigned __int128)ey * (unsigned __int128)ez * (unsigned __int128)nb;
    if (total >> 63) return TP3D_E_TOOBIG;
    unsigned bits = 1;
    while (bits < 63 && ((unsigned __int128)1 << bits) < total) ++bits;
    VoxelWorkspace w = carve_voxel_workspace(workspace, N);
    if (workspace_bytes < w.bytes) return TP3D_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((N + VX_BLOCK - 1) / VX_BLOCK);
    hipLaunchKernelGGL(voxel_key_kernel, dim3(blocks), dim3(VX_BLOCK), 0, s, pos, batch, N, size, bounds_host[0],
                       bounds_host[1], bounds_host[2], ex, ey, ez, w.keys_in, w.vals_in);
    if (int rc = check_launch()) return rc;
    if (int rc = sort_pairs_u64_u32(w.tmp, w.tmp_bytes, w.keys_in, w.keys_out, w.vals_in, w.vals_out, N, bits, s))
        return rc;
    hipLaunchKernelGGL(voxel_flag_kernel, dim3(blocks), dim3(VX_BLOCK), 0, s, w.keys_out, N, w.flags);
    if (int rc = check_launch()) return rc;
    size_t tb = w.tmp_bytes;
    if (int rc = hip_rc(rocprim::inclusive_scan(w.tmp, tb, (const int *)w.flags, w.cid, (size_t)N, rocprim::plus<int>(), s)))
        return rc;
    if (int rc = zero_async(meta, (size_t)(1 + nb) * sizeof(int64_t), s)) return rc;
    hipLaunchKernelGGL(voxel_scatter_kernel, dim3(blocks), dim3(VX_BLOCK), 0, s, w.vals_out, w.cid, N, cluster, order,
                       cluster_start, last, batch, nb, reinterpret_cast<unsigned long long *>(meta));
    return check_launch();
}

TP3D_EXPORT int tp3d_cluster_mean_f32(const float *x, const int64_t *order, const int64_t *cluster_start, int64_t K, int C,
                                      float *out, void *stream)
{
    if (K < 0 || C <= 0) return TP3D_E_BADARG;
    if (K == 0) return TP3D_OK;
    if (!x || !order || !cluster_start || !out) return TP3D_E_BADARG;
    const int64_t blocks = (K * C + VX_BLOCK - 1) / VX_BLOCK;
    if (blocks > 0x7fffffff) return TP3D_E_TOOBIG;
    hipLaunchKernelGGL(cluster_mean_kernel, dim3((unsigned)blocks), dim3(VX_BLOCK), 0, (hipStream_t)stream, x, order,
                       cluster_start, K, C, out);
    return check_launch();
}

TP3D_EXPORT int tp3d_cluster_majority_i64(const int64_t *labels, const int64_t *order, const int64_t *cluster_start,
                                          int64_t K, int64_t min_label, int64_t num_classes, int64_t *out, void *stream)
{
    if (K < 0 || num_classes <= 0) return TP3D_E_BADARG;
    if (K == 0) return TP3D_OK;
    if (!labels || !order || !cluster_start || !out) return TP3D_E_BADARG;
    const int64_t blocks = (K + VX_BLOCK / 64 - 1) / (VX_BLOCK / 64);
    if (blocks > 0x7fffffff) return TP3D_E_TOOBIG;
    const int nc = num_classes > 0x7fffffff ? 0x7fffffff : (int)num_classes;
    hipLaunchKernelGGL(cluster_majority_kernel, dim3((unsigned)blocks), dim3(VX_BLOCK), 0, (hipStream_t)stream, labels,
                       order, cluster_start, K, min_label, nc, out);
    return check_launch();
}
