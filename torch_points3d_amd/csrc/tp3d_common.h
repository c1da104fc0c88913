// Shared device/host helpers for libtp3d_hip.so (gfx950 only, wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tp3d_hip.h"

#define TP3D_EXPORT extern "C" __attribute__((visibility("default")))

namespace tp3d {

constexpr int kWave = 64;

void set_last_hip_error(hipError_t e);

// Checks the launch that was just enqueued; called by every entry point.
inline int check_launch()
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return TP3D_E_LAUNCH;
    }
    return TP3D_OK;
}

// Kernels that need more than 64 KiB of dynamic LDS must opt in once per (function, device).
inline void allow_large_dynamic_lds(const void *func, int bytes, bool *done_per_device /*[64]*/)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!done_per_device[dev]) {
        (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done_per_device[dev] = true;
    }
}

// Zero-fills a device buffer on the stream (used by the scatter-add backward entry points).
inline int zero_async(void *ptr, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return TP3D_OK;
    hipError_t e = hipMemsetAsync(ptr, 0, bytes, s);
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return TP3D_E_LAUNCH;
    }
    return TP3D_OK;
}

// Squared distance in the one evaluation order shared with oracle/tpk_ref_cpu.c.
// The translation units are built with -ffp-contract=off: no v_fma may be formed here.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = ax - bx;
    float dy = ay - by;
    float dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

// csr.hip: inverse index of a neighbour table + the gather-sum that replaces scatter-add atomics
struct ScatterWorkspace {
    int *start;      // B*(nbins+1)
    int *order;      // B*L
    int *scratch;    // B*L (only touched when a cloud's tables do not fit LDS)
    float *wsorted;  // B*L or null
    int *merge_tmp;  // B*L: second buffer of the run merge that sorts bins of more than 1024 slots (flat inversion)
    int *hubs;       // 1 + B*nbins: count, then the ids b*nbins + k of the destinations with long runs (rows.hip)
    size_t bytes;
};
ScatterWorkspace carve_scatter_workspace(void *ws, int B, int L, int nbins, bool with_weights);
int csr_transpose(const int64_t *idx, int B, int L, int nbins, int div, const float *weight, int *start, int *order,
                  float *wsorted, int *scratch_ord, hipStream_t s);
int gather_sum(const float *rows, const int *start, const int *order, const float *wsorted, int B, int C, int nbins,
               int Lrow, int Lslots, float *out, hipStream_t s);

// kpconv.hip: multi-workgroup inverse of an index table (cnt, cursor: M ints; start: M + 1; order: slots).
// per_cloud_slots > 0: idx is (clouds, per_cloud_slots) with values clamped to [0, per_cloud_bins); bin = cloud *
// per_cloud_bins + value, M = clouds * per_cloud_bins (one flat table over the batch).
int invert_table(const int64_t *idx, int64_t slots, int64_t M, int *cnt, int *start, int *cursor, int *order,
                 hipStream_t s, int64_t per_cloud_slots = 0, int64_t per_cloud_bins = 0, int *merge_tmp = nullptr);
// csr.hip: does the one-workgroup-per-cloud transpose fit LDS?
bool csr_fits_lds(int L, int nbins);

// gemm_tn.hip: out[e] = sum over the row splits of partial[s][e], in ascending split order (reproducible)
// dbeta, dgamma (, c1, c2) from chunk partials [chunks][2][C] (rows.hip)
int bn_bwd_finalize_launch(const float *partial, int chunks, int C, float *dbeta, float *dgamma, const float *invstd, int64_t M,
                           int training, float *c1, float *c2, hipStream_t s);
int tn_reduce_splits(const float *partial, int splits, int64_t NK, float *out, hipStream_t s);

// grid.hip: uniform-grid radius search (build + query); seg/batch_y null => dense layout (more in grid.h)
int grid_ball_query(const float *x, const float *y, const int64_t *seg, const int64_t *batch_y, int num_clouds,
                    int64_t rows, int N, int np, int64_t total_q, int Lmax, float radius, int nsample, int sort,
                    int64_t *idx, float *dist2, void *workspace, size_t workspace_bytes, bool reuse_grid, hipStream_t s);

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}


// Bitonic sort of one bin (n <= 64 * NU slot ids) by one wave: element i lives in lane i & 63, register i >> 6.
template <int NU, typename OrdT>
__device__ __forceinline__ void wave_sort_bin(OrdT *__restrict__ bin, int n, int lane)
{
    int e[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) e[u] = (lane + 64 * u < n) ? (int)bin[lane + 64 * u] : 0x7fffffff;
#pragma unroll
    for (int size = 2; size <= 64 * NU; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            if (stride >= 64) {  // partner in the same lane, another register
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int pu = u ^ (stride >> 6);
                    if (pu > u) {
                        const bool up = (((u << 6) | lane) & size) == 0;
                        const int a = e[u], b = e[pu];
                        const bool swap = up ? a > b : a < b;
                        e[u] = swap ? b : a;
                        e[pu] = swap ? a : b;
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int other = __shfl_xor(e[u], stride);
                    const bool up = (((u << 6) | lane) & size) == 0;
                    const bool lower = (lane & stride) == 0;
                    e[u] = (lower == up) ? min(e[u], other) : max(e[u], other);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();  // (the loads above all happened before the first exchange)
#pragma unroll
    for (int u = 0; u < NU; ++u)
        if (lane + 64 * u < n) bin[lane + 64 * u] = (OrdT)e[u];
}

}  // namespace tp3d
