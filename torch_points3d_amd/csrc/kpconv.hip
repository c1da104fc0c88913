// KPConv rigid kernel-point convolution, stage 1: kernel-point weighted neighbourhood features.
//
// Reference: torch_points3d/modules/KPConv/convolution_ops.py:19-107 (KPConv_ops) with the gather of
// core/common_modules/gathering.py:1-33 (index -1 = shadow neighbour: point at 1e6, zero feature).
//   wf[q, k, :] = sum_n h(| (s[nbr[q,n]] - q) - K_k |) * x[nbr[q,n], :]          (convolution_ops.py:49-98)
//   out[q, :]   = sum_k wf[q, k, :] @ W[k]     == (Nq, KP*Cin) @ (KP*Cin, Cout)    (:101-105, one library GEMM)
// The reference materialises (Nq, Mn, KP, 3) differences and (Nq, KP, Mn) weights in HBM; here one wave owns a
// query: the KP x Mn influence weights live in LDS, neighbour feature rows are read as coalesced row segments
// and KP accumulators per lane stay in registers, so HBM sees the neighbour rows once and wf once.
#include "tp3d_common.h"

namespace tp3d {

constexpr int KP_BLOCK = 256;  // 4 waves, one query per wave
constexpr int KP_MAX = 16;     // kernel points (15 in every reference config)
constexpr int KP_NCH = 48;     // neighbours per LDS pass (covers every max_num_neighbors of the reference configs)

__global__ __launch_bounds__(KP_BLOCK) void kpconv_weighted_kernel(
    const float *__restrict__ query, const float *__restrict__ support, const int64_t *__restrict__ nbr,
    const float *__restrict__ feat, const float *__restrict__ kpts, int64_t Nq, int64_t M, int Mn, int Cin, int KP,
    float extent, int influence, int closest, float *__restrict__ wf)
{
    __shared__ __attribute__((aligned(16))) float s_w[KP_BLOCK / 64][KP_NCH][KP_MAX];
    __shared__ float s_d[KP_BLOCK / 64][KP_NCH][KP_MAX];
    __shared__ int s_id[KP_BLOCK / 64][KP_NCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t qraw = (int64_t)blockIdx.x * (KP_BLOCK / 64) + wave;
    const bool live = qraw < Nq;  // idle waves keep walking the barriers
    const int64_t q = live ? qraw : Nq - 1;
    const float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
    float(*w)[KP_MAX] = s_w[wave];
    float(*dd)[KP_MAX] = s_d[wave];
    int *ids = s_id[wave];
    const float sigma = extent * 0.3f;
    const float gden = 2.0f * sigma * sigma + 1e-9f;

    const bool single_pass = Mn <= KP_NCH;  // then the weights of phase A serve every channel chunk
    for (int c0 = 0; c0 < Cin; c0 += 64) {
        float acc[KP_MAX];
#pragma unroll
        for (int k = 0; k < KP_MAX; ++k) acc[k] = 0.0f;
        const int c = c0 + lane;
        for (int n0 = 0; n0 < Mn; n0 += KP_NCH) {
            const int cnt = min(KP_NCH, Mn - n0);
            // ---- phase A: influence weight of every (neighbour, kernel point) pair of this chunk
            if (!(single_pass && c0 > 0))
            for (int p = lane; p < cnt * KP_MAX; p += 64) {
                const int n = p / KP_MAX, k = p % KP_MAX;
                const int64_t id = nbr[q * Mn + n0 + n];
                const bool shadow = id < 0 || id >= M;
                if (k == 0) ids[n] = shadow ? -1 : (int)id;
                float wv = 0.0f, d2 = 3.0e38f;
                if (!shadow && k < KP) {
                    const float cx = support[id * 3 + 0] - qx, cy = support[id * 3 + 1] - qy,
                                cz = support[id * 3 + 2] - qz;
                    const float dx = cx - kpts[k * 3 + 0], dy = cy - kpts[k * 3 + 1], dz = cz - kpts[k * 3 + 2];
                    d2 = (dx * dx + dy * dy) + dz * dz;
                    if (influence == 0) wv = 1.0f;
                    else if (influence == 1) wv = fmaxf(1.0f - sqrtf(d2) / extent, 0.0f);
                    else wv = expf(-d2 / gden);
                }
                w[n][k] = wv;
                dd[n][k] = d2;
            }
            __syncthreads();
            if (closest && !(single_pass && c0 > 0)) {  // only the closest kernel point keeps its influence
                if (lane < cnt) {
                    int kb = 0;
                    float best = dd[lane][0];
                    for (int k = 1; k < KP; ++k)
                        if (dd[lane][k] < best) {
                            best = dd[lane][k];
                            kb = k;
                        }
                    for (int k = 0; k < KP; ++k)
                        if (k != kb) w[lane][k] = 0.0f;
                }
            }
            __syncthreads();
            // ---- phase B: accumulate the neighbour rows into the KP accumulators (lanes over channels)
            if (c < Cin) {
                for (int n = 0; n < cnt; ++n) {
                    const int id = ids[n];
                    if (id < 0) continue;  // wave-uniform
                    const float v = feat[(size_t)id * Cin + c];
                    const float4 w0 = *reinterpret_cast<const float4 *>(&w[n][0]);
                    const float4 w1 = *reinterpret_cast<const float4 *>(&w[n][4]);
                    const float4 w2 = *reinterpret_cast<const float4 *>(&w[n][8]);
                    const float4 w3 = *reinterpret_cast<const float4 *>(&w[n][12]);
                    acc[0] += w0.x * v;  acc[1] += w0.y * v;  acc[2] += w0.z * v;  acc[3] += w0.w * v;
                    acc[4] += w1.x * v;  acc[5] += w1.y * v;  acc[6] += w1.z * v;  acc[7] += w1.w * v;
                    acc[8] += w2.x * v;  acc[9] += w2.y * v;  acc[10] += w2.z * v; acc[11] += w2.w * v;
                    acc[12] += w3.x * v; acc[13] += w3.y * v; acc[14] += w3.z * v; acc[15] += w3.w * v;
                }
            }
            __syncthreads();
        }
        if (live && c < Cin) {
#pragma unroll
            for (int k = 0; k < KP_MAX; ++k)
                if (k < KP) wf[((size_t)q * KP + k) * Cin + c] = acc[k];
        }
    }
}

}  // namespace tp3d

using namespace tp3d;

TP3D_EXPORT int tp3d_kpconv_weighted_f32(const float *query, const float *support, const int64_t *neighbors,
                                         const float *features, const float *k_points, int64_t Nq, int64_t M,
                                         int Mn, int Cin, int KP, float extent, int influence, int closest,
                                         float *weighted, void *stream)
{
    if (Nq < 0 || M < 0 || Mn < 0 || Cin <= 0 || KP <= 0 || influence < 0 || influence > 2) return TP3D_E_BADARG;
    if (KP > KP_MAX) return TP3D_E_TOOBIG;
    if (Nq == 0) return TP3D_OK;
    if (!query || !weighted || !k_points || (Mn > 0 && (!neighbors || !support || !features))) return TP3D_E_BADARG;
    const int64_t blocks = (Nq + KP_BLOCK / 64 - 1) / (KP_BLOCK / 64);
    if (blocks > 0x7fffffff) return TP3D_E_TOOBIG;
    hipLaunchKernelGGL(kpconv_weighted_kernel, dim3((unsigned)blocks), dim3(KP_BLOCK), 0, (hipStream_t)stream, query,
                       support, neighbors, features, k_points, Nq, M, Mn, Cin, KP, extent, influence, closest,
                       weighted);
    return check_launch();
}
