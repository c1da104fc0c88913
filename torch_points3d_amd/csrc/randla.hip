// RandLA-Net local feature aggregation on a fixed-k neighbour table.
//
// Reference: torch_points3d/modules/RandLANet/modules.py:9-54 (RandlaKernel.message / update) run by torch_geometric's
// MessagePassing over an edge list.  With the (Nq, k) table of libtp3d_hip.so's exact kNN every query owns k
// consecutive edges, so the two edge-wise pieces that are not MLPs become row kernels:
//   * relative position encoding  [pos_i, pos_j, pos_i - pos_j, |pos_i - pos_j|]          (modules.py:36-41)
//   * attentive pooling           out[q] = sum_k softmax_c(g[e, :]) * f[e, :],  e = q*k + n  (modules.py:46-52 + aggr="add")
// Both are HBM-bound: the pooling reads g and f once (2 * E * C * 4 B) and writes Nq * C * 4 B; the reference's
// softmax / mul / scatter-add chain moves the same rows seven times.
#include "tp3d_common.h"

namespace tp3d {

// one thread per edge; rows of 12 floats (10 used) so that the MLP's GEMM reads aligned float4
__global__ __launch_bounds__(256) void randla_relpos_kernel(const float *__restrict__ q_pos,
                                                           const float *__restrict__ s_pos,
                                                           const int64_t *__restrict__ nbr, int64_t E, int k, int64_t M,
                                                           float4 *__restrict__ out)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int64_t q = e / k, j = nbr[e];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a;
    if (j >= 0 && j < M) {
        const float qx = q_pos[q * 3 + 0], qy = q_pos[q * 3 + 1], qz = q_pos[q * 3 + 2];
        const float sx = s_pos[j * 3 + 0], sy = s_pos[j * 3 + 1], sz = s_pos[j * 3 + 2];
        const float dx = qx - sx, dy = qy - sy, dz = qz - sz;
        const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
        a = make_float4(qx, qy, qz, sx);
        b = make_float4(sy, sz, dx, dy);
        c = make_float4(dz, d, 0.f, 0.f);
    }
    out[e * 3 + 0] = a;
    out[e * 3 + 1] = b;
    out[e * 3 + 2] = c;
}

// Lane layout of the pooling kernels: one wave per query.  P = lanes per edge (power of two >= min(C, 64)), 64 / P
// edges per pass, R = ceil(C / P) channels per lane (channel c = r * P + lane % P).
template <int R>
struct AttnRow {
    float g[R], f[R];
};

__device__ __forceinline__ float seg_max(float v, int P)
{
    for (int off = 1; off < P; off <<= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float seg_sum(float v, int P)
{
    for (int off = 1; off < P; off <<= 1) v += __shfl_xor(v, off);
    return v;
}

// softmax over the channels of one edge row held across a P-lane segment; returns s[r], leaves row.f untouched
template <int R>
__device__ __forceinline__ void edge_softmax(const AttnRow<R> &row, const bool (&live)[R], int P, float (&s)[R])
{
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (live[r]) m = fmaxf(m, row.g[r]);
    m = seg_max(m, P);
    float z = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = live[r] ? expf(row.g[r] - m) : 0.0f;
        z += s[r];
    }
    z = seg_sum(z, P);
    const float inv = 1.0f / z;
#pragma unroll
    for (int r = 0; r < R; ++r) s[r] *= inv;
}

template <int R>
__global__ __launch_bounds__(256) void attn_pool_fwd_kernel(const float *__restrict__ g, const float *__restrict__ f,
                                                            const int64_t *__restrict__ nbr, int64_t Nq, int k, int C,
                                                            int ldg, int ldf, int P, float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Nq) return;  // whole waves leave together
    const int cl = lane & (P - 1), slot = lane / P, epp = 64 / P;
    bool live[R];
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        live[r] = r * P + cl < C;
        acc[r] = 0.0f;
    }
    for (int n0 = 0; n0 < k; n0 += epp) {
        const int n = n0 + slot;
        const int64_t e = q * k + (n < k ? n : 0);
        // idle slots and missing neighbours (-1: cloud smaller than k) run the shuffles on a neutral row
        const bool have = n < k && (!nbr || nbr[e] >= 0);
        AttnRow<R> row;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ld_ = have && live[r];
            row.g[r] = ld_ ? g[e * ldg + r * P + cl] : 0.0f;
            row.f[r] = ld_ ? f[e * ldf + r * P + cl] : 0.0f;
        }
        float s[R];
        edge_softmax<R>(row, live, P, s);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] += s[r] * row.f[r];  // f = 0 in idle slots
    }
    // sum the edge slots (lanes that share cl): fixed butterfly order
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float v = acc[r];
        for (int off = P; off < 64; off <<= 1) v += __shfl_xor(v, off);
        if (slot == 0 && live[r]) out[q * C + r * P + cl] = v;
    }
}

// msg = s * f, s = softmax(g):   df = s * dout;   dg = s * (f * dout - sum_c s * f * dout)
template <int R>
__global__ __launch_bounds__(256) void attn_pool_bwd_kernel(const float *__restrict__ g, const float *__restrict__ f,
                                                            const float *__restrict__ dout,
                                                            const int64_t *__restrict__ nbr, int64_t Nq, int k, int C,
                                                            int ldg, int ldf, int P, float *__restrict__ dg,
                                                            float *__restrict__ df)
{
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Nq) return;
    const int cl = lane & (P - 1), slot = lane / P, epp = 64 / P;
    bool live[R];
    float go[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        live[r] = r * P + cl < C;
        go[r] = live[r] ? dout[q * C + r * P + cl] : 0.0f;
    }
    for (int n0 = 0; n0 < k; n0 += epp) {
        const int n = n0 + slot;
        const int64_t e = q * k + (n < k ? n : 0);
        const bool have = n < k && (!nbr || nbr[e] >= 0);
        AttnRow<R> row;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ld_ = have && live[r];
            row.g[r] = ld_ ? g[e * ldg + r * P + cl] : 0.0f;
            row.f[r] = ld_ ? f[e * ldf + r * P + cl] : 0.0f;
        }
        float s[R];
        edge_softmax<R>(row, live, P, s);
        float dot = 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) dot += s[r] * (row.f[r] * go[r]);
        dot = seg_sum(dot, P);
        if (n < k) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int c = r * P + cl;
                if (live[r]) {
                    dg[e * ldg + c] = have ? s[r] * (row.f[r] * go[r] - dot) : 0.0f;
                    df[e * ldf + c] = have ? s[r] * go[r] : 0.0f;
                } else if (c < ldf) {
                    df[e * ldf + c] = 0.0f;  // padding columns of the feature rows
                }
            }
        }
    }
}

struct AttnPlan {
    int P, R;
};
static AttnPlan attn_plan(int C, int ldf)
{
    AttnPlan p;
    p.P = 4;
    while (p.P < 64 && p.P < C) p.P <<= 1;
    const int span = ldf > C ? ldf : C;  // the backward also clears the padding columns of df
    p.R = (span + p.P - 1) / p.P;
    return p;
}

}  // namespace tp3d

using namespace tp3d;

TP3D_EXPORT int tp3d_randla_relpos_f32(const float *q_pos, const float *s_pos, const int64_t *nbr, int64_t Nq, int k,
                                       int64_t M, float *out, void *stream)
{
    if (Nq < 0 || k <= 0 || M < 0) return TP3D_E_BADARG;
    const int64_t E = Nq * k;
    if (E == 0) return TP3D_OK;
    if (!q_pos || !s_pos || !nbr || !out || M == 0) return TP3D_E_BADARG;
    if ((E + 255) / 256 > 0x7fffffffLL) return TP3D_E_TOOBIG;
    hipLaunchKernelGGL(randla_relpos_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, (hipStream_t)stream, q_pos,
                       s_pos, nbr, E, k, M, reinterpret_cast<float4 *>(out));
    return check_launch();
}

#define TP3D_ATTN_DISPATCH(KERNEL, ...)                                                                              \
    switch (p.R) {                                                                                                   \
    case 1: hipLaunchKernelGGL(KERNEL<1>, grid, dim3(256), 0, s, __VA_ARGS__); break;                                \
    case 2: hipLaunchKernelGGL(KERNEL<2>, grid, dim3(256), 0, s, __VA_ARGS__); break;                                \
    case 3: hipLaunchKernelGGL(KERNEL<3>, grid, dim3(256), 0, s, __VA_ARGS__); break;                                \
    default: hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), 0, s, __VA_ARGS__); break;                               \
    }

static int attn_check(int64_t Nq, int k, int C, int ldg, int ldf)
{
    if (Nq < 0 || k <= 0 || C <= 0 || ldg < C || ldf < C) return TP3D_E_BADARG;
    if (C > 256 || ldf > 256) return TP3D_E_TOOBIG;
    if ((Nq + 3) / 4 > 0x7fffffffLL) return TP3D_E_TOOBIG;
    return TP3D_OK;
}

TP3D_EXPORT int tp3d_attn_pool_fwd_f32(const float *g, const float *f, const int64_t *nbr, int64_t Nq, int k, int C,
                                       int ldg, int ldf, float *out, void *stream)
{
    if (int rc = attn_check(Nq, k, C, ldg, ldf)) return rc;
    if (Nq == 0) return TP3D_OK;
    if (!g || !f || !out) return TP3D_E_BADARG;
    AttnPlan p = attn_plan(C, C);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((Nq + 3) / 4));
    TP3D_ATTN_DISPATCH(attn_pool_fwd_kernel, g, f, nbr, Nq, k, C, ldg, ldf, p.P, out)
    return check_launch();
}

TP3D_EXPORT int tp3d_attn_pool_bwd_f32(const float *g, const float *f, const float *dout, const int64_t *nbr, int64_t Nq,
                                       int k, int C, int ldg, int ldf, float *dg, float *df, void *stream)
{
    if (int rc = attn_check(Nq, k, C, ldg, ldf)) return rc;
    if (Nq == 0) return TP3D_OK;
    if (!g || !f || !dout || !dg || !df) return TP3D_E_BADARG;
    AttnPlan p = attn_plan(C, ldf);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((Nq + 3) / 4));
    TP3D_ATTN_DISPATCH(attn_pool_bwd_kernel, g, f, dout, nbr, Nq, k, C, ldg, ldf, p.P, dg, df)
    return check_launch();
}
