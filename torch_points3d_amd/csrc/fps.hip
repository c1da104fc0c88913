// furthest_point_sample -- one workgroup per cloud, the cloud and its running min-distance held in
// registers (P points per lane), per-step argmax = in-lane scan -> 64-lane butterfly -> one LDS
// exchange between the waves.  The winner's coordinates travel with the (value, index) pair so no
// global re-read sits on the serial chain.
//
// Reference contract: torch_points3d/core/spatial_ops/sampling.py:100 (DenseFPSSampler.sample) ->
// tp.furthest_point_sample(pos, npoint); semantics SURVEY.md 8a-H1; oracle tpk_ref_fps_f32.
#include "tp3d_common.h"

namespace tp3d {

struct Cand {
    float v;  // running min squared distance (the arg-max key); -1 for lanes/points past N
    int i;    // point index
};

// larger value wins; equal values -> lower index wins (the oracle scans j ascending with strict '>')
__device__ __forceinline__ bool beats(float ov, int oi, float v, int i) { return ov > v || (ov == v && oi < i); }

__device__ __forceinline__ Cand wave_argmax(Cand c)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        float ov = __shfl_xor(c.v, off);
        int oi = __shfl_xor(c.i, off);
        if (beats(ov, oi, c.v, c.i)) {
            c.v = ov;
            c.i = oi;
        }
    }
    return c;
}

// BLOCK threads, P points per thread (point j = t + k*BLOCK lives in thread t, slot k).
template <int BLOCK, int P>
__global__ __launch_bounds__(BLOCK) void fps_reg_kernel(const float *__restrict__ xyz, int N, int npoint,
                                                         int64_t *__restrict__ out)
{
    constexpr int NW = BLOCK / kWave;
    __shared__ float s_v[2][NW];
    __shared__ int s_i[2][NW];

    const int b = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & (kWave - 1);
    const int wave = t / kWave;
    const float *p = xyz + (size_t)b * N * 3;
    int64_t *o = out + (size_t)b * npoint;

    float px[P], py[P], pz[P], md[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        int j = t + k * BLOCK;
        bool ok = j < N;
        px[k] = ok ? p[(size_t)j * 3 + 0] : 0.0f;
        py[k] = ok ? p[(size_t)j * 3 + 1] : 0.0f;
        pz[k] = ok ? p[(size_t)j * 3 + 2] : 0.0f;
        md[k] = ok ? 1e10f : -1.0f;  // -1 never exceeds the scan's initial best of -1
    }

    if (t == 0 && npoint > 0) o[0] = 0;
    float lx = p[0], ly = p[1], lz = p[2];

    for (int it = 1; it < npoint; ++it) {
        Cand c;
        c.v = -1.0f;
        c.i = 0;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            float d = sqdist3(px[k], py[k], pz[k], lx, ly, lz);
            float m = md[k] < d ? md[k] : d;
            md[k] = m;
            if (m > c.v) {
                c.v = m;
                c.i = t + k * BLOCK;
            }
        }
        c = wave_argmax(c);
        const int buf = it & 1;
        if (NW > 1) {
            if (lane == 0) {
                s_v[buf][wave] = c.v;
                s_i[buf][wave] = c.i;
            }
            __syncthreads();
            // every wave re-reduces the NW partials; lanes >= NW replicate entry (lane % NW)
            Cand r;
            r.v = s_v[buf][lane % NW];
            r.i = s_i[buf][lane % NW];
#pragma unroll
            for (int off = NW / 2; off >= 1; off >>= 1) {
                float ov = __shfl_xor(r.v, off);
                int oi = __shfl_xor(r.i, off);
                if (beats(ov, oi, r.v, r.i)) {
                    r.v = ov;
                    r.i = oi;
                }
            }
            c = r;
        }
        const int last = __builtin_amdgcn_readfirstlane(c.i);
        lx = p[(size_t)last * 3 + 0];
        ly = p[(size_t)last * 3 + 1];
        lz = p[(size_t)last * 3 + 2];
        if (t == 0) o[it] = last;
    }
}

// Any N: running min-distance kept in caller scratch (B*N floats), cloud re-read from L2 each step.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void fps_generic_kernel(const float *__restrict__ xyz, int N, int npoint,
                                                             float *__restrict__ scratch,
                                                             int64_t *__restrict__ out)
{
    constexpr int NW = BLOCK / kWave;
    __shared__ float s_v[2][NW];
    __shared__ int s_i[2][NW];
    const int b = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & (kWave - 1);
    const int wave = t / kWave;
    const float *p = xyz + (size_t)b * N * 3;
    float *md = scratch + (size_t)b * N;
    int64_t *o = out + (size_t)b * npoint;

    for (int j = t; j < N; j += BLOCK) md[j] = 1e10f;
    if (t == 0 && npoint > 0) o[0] = 0;
    float lx = p[0], ly = p[1], lz = p[2];

    for (int it = 1; it < npoint; ++it) {
        Cand c;
        c.v = -1.0f;
        c.i = 0;
        for (int j = t; j < N; j += BLOCK) {  // each thread only ever touches its own md[j]
            float d = sqdist3(p[(size_t)j * 3 + 0], p[(size_t)j * 3 + 1], p[(size_t)j * 3 + 2], lx, ly, lz);
            float m = md[j];
            m = m < d ? m : d;
            md[j] = m;
            if (m > c.v) {
                c.v = m;
                c.i = j;
            }
        }
        c = wave_argmax(c);
        const int buf = it & 1;
        if (lane == 0) {
            s_v[buf][wave] = c.v;
            s_i[buf][wave] = c.i;
        }
        __syncthreads();
        Cand r;
        r.v = s_v[buf][lane % NW];
        r.i = s_i[buf][lane % NW];
#pragma unroll
        for (int off = NW / 2; off >= 1; off >>= 1) {
            float ov = __shfl_xor(r.v, off);
            int oi = __shfl_xor(r.i, off);
            if (beats(ov, oi, r.v, r.i)) {
                r.v = ov;
                r.i = oi;
            }
        }
        const int last = __builtin_amdgcn_readfirstlane(r.i);
        lx = p[(size_t)last * 3 + 0];
        ly = p[(size_t)last * 3 + 1];
        lz = p[(size_t)last * 3 + 2];
        if (t == 0) o[it] = last;
    }
}

template <int BLOCK, int P>
static void launch_reg(const float *xyz, int B, int N, int npoint, int64_t *out, hipStream_t s)
{
    hipLaunchKernelGGL((fps_reg_kernel<BLOCK, P>), dim3(B), dim3(BLOCK), 0, s, xyz, N, npoint, out);
}

}  // namespace tp3d

TP3D_EXPORT int tp3d_fps_f32(const float *xyz, int B, int N, int npoint, float *scratch, int64_t *out_idx,
                             void *stream)
{
    using namespace tp3d;
    if (B < 0 || N <= 0 || npoint < 0 || npoint > N) return TP3D_E_BADARG;
    if (B == 0 || npoint == 0) return TP3D_OK;
    if (!xyz || !out_idx) return TP3D_E_BADARG;
    if ((int64_t)N * 3 > INT32_MAX) return TP3D_E_TOOBIG;
    hipStream_t s = (hipStream_t)stream;
    if (N <= 64) launch_reg<64, 1>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 256) launch_reg<256, 1>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 512) launch_reg<256, 2>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 1024) launch_reg<256, 4>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 2048) launch_reg<1024, 2>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 4096) launch_reg<1024, 4>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 8192) launch_reg<1024, 8>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 16384) launch_reg<1024, 16>(xyz, B, N, npoint, out_idx, s);
    else if (N <= TP3D_FPS_MAX_REG_POINTS) launch_reg<1024, 32>(xyz, B, N, npoint, out_idx, s);
    else {
        if (!scratch) return TP3D_E_BADARG;
        hipLaunchKernelGGL((fps_generic_kernel<1024>), dim3(B), dim3(1024), 0, s, xyz, N, npoint, scratch,
                           out_idx);
    }
    return check_launch();
}
