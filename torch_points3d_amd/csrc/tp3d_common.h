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
    size_t bytes;
};
ScatterWorkspace carve_scatter_workspace(void *ws, int B, int L, int nbins, bool with_weights);
int csr_transpose(const int64_t *idx, int B, int L, int nbins, int div, const float *weight, int *start, int *order,
                  float *wsorted, int *scratch_ord, hipStream_t s);
int gather_sum(const float *rows, const int *start, const int *order, const float *wsorted, int B, int C, int nbins,
               int Lrow, int Lslots, float *out, hipStream_t s);

// kpconv.hip: multi-workgroup inverse of an index table (cnt, cursor: M ints; start: M + 1; order: slots)
int invert_table(const int64_t *idx, int64_t slots, int64_t M, int *cnt, int *start, int *cursor, int *order,
                 hipStream_t s);

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

}  // namespace tp3d
