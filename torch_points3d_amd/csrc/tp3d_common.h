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

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int lanes_below(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

}  // namespace tp3d
