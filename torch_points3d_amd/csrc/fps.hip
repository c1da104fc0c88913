// furthest_point_sample -- one workgroup per cloud, the cloud and its running min-distance held in
// registers (P points per lane).  Per step: in-lane scan -> wave arg-max on the DPP cross-lane network
// (value max, then lowest index among the lanes holding it) -> the owning lane's coordinates are read out
// of the register array with a wave-uniform index -> ONE LDS exchange + ONE barrier between the waves.
// The winner's coordinates travel with the (value, index) pair, so no memory access sits on the serial chain.
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

// ---- DPP helpers: all-reduce inside each row of 16 lanes, then combine the 4 rows through SGPRs ----
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ float row_allmax(float x)
{
    x = fmaxf(x, dpp_f<0xB1>(x));   // quad_perm [1,0,3,2]
    x = fmaxf(x, dpp_f<0x4E>(x));   // quad_perm [2,3,0,1]
    x = fmaxf(x, dpp_f<0x124>(x));  // row_ror:4
    x = fmaxf(x, dpp_f<0x128>(x));  // row_ror:8
    return x;
}
__device__ __forceinline__ int row_allmin(int x)
{
    x = min(x, dpp_i<0xB1>(x));
    x = min(x, dpp_i<0x4E>(x));
    x = min(x, dpp_i<0x124>(x));
    x = min(x, dpp_i<0x128>(x));
    return x;
}
__device__ __forceinline__ float wave_allmax(float x)
{
    x = row_allmax(x);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 32));
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ int wave_allmin(int x)
{
    x = row_allmin(x);
    const int a = __builtin_amdgcn_readlane(x, 0), b = __builtin_amdgcn_readlane(x, 16);
    const int c = __builtin_amdgcn_readlane(x, 32), d = __builtin_amdgcn_readlane(x, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ float readlane_f(float x, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}

typedef float f2 __attribute__((ext_vector_type(2)));

// BLOCK threads, P (even) points per thread: point j = t + k*BLOCK lives in thread t, slot k; slots are held
// as P/2 register PAIRS so the distance arithmetic issues as packed fp32 (v_pk_add/mul_f32, 2 points per
// instruction -- the kernel is VALU-issue bound, not memory bound).
template <int BLOCK, int P>
__global__ __launch_bounds__(BLOCK) void fps_reg_kernel(const float *__restrict__ xyz, int N, int npoint,
                                                         int64_t *__restrict__ out)
{
    constexpr int NW = BLOCK / kWave;
    constexpr int H = P / 2;
    static_assert(P % 2 == 0, "points are processed in packed pairs");
    static_assert(NW == 1 || NW == 4 || NW == 8 || NW == 16, "cross-wave exchange assumes the partials fit one DPP row");
    __shared__ float s_v[2][NW], s_x[2][NW], s_y[2][NW], s_z[2][NW];
    __shared__ int s_i[2][NW];

    const int b = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & (kWave - 1);
    const int wave = t / kWave;
    const float *p = xyz + (size_t)b * N * 3;
    int64_t *o = out + (size_t)b * npoint;

    f2 px[H], py[H], pz[H], md[H];  // element e of pair h is slot k = 2h + e
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int j = t + k * BLOCK;
        const bool ok = j < N;
        px[k / 2][k & 1] = ok ? p[(size_t)j * 3 + 0] : 0.0f;
        py[k / 2][k & 1] = ok ? p[(size_t)j * 3 + 1] : 0.0f;
        pz[k / 2][k & 1] = ok ? p[(size_t)j * 3 + 2] : 0.0f;
        md[k / 2][k & 1] = ok ? 1e10f : -1.0f;  // -1 never wins: every real point has min-distance >= 0
    }

    if (t == 0 && npoint > 0) o[0] = 0;
    float lx = p[0], ly = p[1], lz = p[2];

    for (int it = 1; it < npoint; ++it) {
        // distance update + per-lane max VALUE only (the index is resolved once per wave, below); the maxima are kept
        // per group of 8 slots so that the resolution can skip the groups that do not hold the wave's maximum
        constexpr int NG = P >= 16 ? P / 8 : 1;  // slot groups
        constexpr int GS = P / NG;               // slots per group
        float gmax[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) gmax[g] = -1.0f;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const f2 dx = px[h] - lx, dy = py[h] - ly, dz = pz[h] - lz;
            const f2 d = (dx * dx + dy * dy) + dz * dz;  // same order as sqdist3; contraction is off
            md[h][0] = fminf(md[h][0], d[0]);
            md[h][1] = fminf(md[h][1], d[1]);
            gmax[(2 * h) / GS] = fmaxf(fmaxf(gmax[(2 * h) / GS], md[h][0]), md[h][1]);
        }
        float best = gmax[0];
#pragma unroll
        for (int g = 1; g < NG; ++g) best = fmaxf(best, gmax[g]);
        // wave arg-max: max value, then the lowest index among the (lane, slot) pairs that hold it.  Only the groups in
        // which some lane holds the maximum are scanned (in-kernel clocks: scanning all 32 slots in every lane cost
        // 0.37 us of a 1.30 us step, more than the distance update itself)
        const float wv = wave_allmax(best);
        int bi = 0x7fffffff;
#pragma unroll
        for (int g = NG - 1; g >= 0; --g) {  // descending: a lower slot overwrites, the lowest slot wins
            if (NG == 1 || __ballot(gmax[g] == wv)) {
#pragma unroll
                for (int k = (g + 1) * GS - 1; k >= g * GS; --k) bi = (md[k / 2][k & 1] == wv) ? t + k * BLOCK : bi;
            }
        }
        const int wi = wave_allmin(bi);
        // owner of wi inside this wave: thread wi % BLOCK (same wave by construction), slot wi / BLOCK
        const int wslot = (wi == 0x7fffffff) ? 0 : wi / BLOCK;
        const int wlane = (wi == 0x7fffffff) ? 0 : (wi % BLOCK) & (kWave - 1);
        // wave-uniform register index (s_set_gpr_idx) + v_readlane: no memory access for the coordinates
        const int wh = wslot >> 1;
        const bool wodd = wslot & 1;
        float ox = readlane_f(wodd ? px[wh][1] : px[wh][0], wlane);
        float oy = readlane_f(wodd ? py[wh][1] : py[wh][0], wlane);
        float oz = readlane_f(wodd ? pz[wh][1] : pz[wh][0], wlane);
        int last;
        if (NW > 1) {
            const int buf = it & 1;
            if (lane == 0) {
                s_v[buf][wave] = wv;
                s_i[buf][wave] = wi;
                s_x[buf][wave] = ox;
                s_y[buf][wave] = oy;
                s_z[buf][wave] = oz;
            }
            __syncthreads();
            // every wave reduces the NW partials itself; lanes >= NW replicate entry lane % NW
            const int e = lane % NW;
            const float ev = s_v[buf][e];
            const int ei = s_i[buf][e];
            const float ex = s_x[buf][e], ey = s_y[buf][e], ez = s_z[buf][e];
            const float gv = readlane_f(row_allmax(ev), 0);
            const int gi = __builtin_amdgcn_readlane(row_allmin(ev == gv ? ei : 0x7fffffff), 0);
            const unsigned long long who = __ballot(ei == gi && ev == gv);
            const int wl = __builtin_ctzll(who);  // a lane holding the winning wave's entry
            last = gi;
            lx = readlane_f(ex, wl);
            ly = readlane_f(ey, wl);
            lz = readlane_f(ez, wl);
        } else {
            last = wi;
            lx = ox;
            ly = oy;
            lz = oz;
        }
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
    // Block size per cloud size, measured on MI355X (tools/microbench.py fps): a step is a serial chain
    // (scan -> DPP arg-max -> LDS exchange -> barrier) of ~0.55 us plus ~0.06 ns per point; 8 waves (512 lanes,
    // 2 per SIMD) beat 16 waves for every N <= 16384 because the barrier and the VALU issue queue are shorter.
    if (N <= 128) launch_reg<64, 2>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 512) launch_reg<256, 2>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 1024) launch_reg<256, 4>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 2048) launch_reg<512, 4>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 4096) launch_reg<512, 8>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 8192) launch_reg<512, 16>(xyz, B, N, npoint, out_idx, s);
    else if (N <= 16384) launch_reg<512, 32>(xyz, B, N, npoint, out_idx, s);
    else if (N <= TP3D_FPS_MAX_REG_POINTS) launch_reg<1024, 32>(xyz, B, N, npoint, out_idx, s);
    else {
        if (!scratch) return TP3D_E_BADARG;
        hipLaunchKernelGGL((fps_generic_kernel<1024>), dim3(B), dim3(1024), 0, s, xyz, N, npoint, scratch,
                           out_idx);
    }
    return check_launch();
}
