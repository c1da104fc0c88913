// three_interpolate forward / backward.
//
// Reference contract: torch_points3d/core/base_conv/dense.py:140 (weights built at :137-139);
// semantics SURVEY.md 8a-H9; oracle tpk_ref_three_interpolate_{fwd,bwd}_f32.
//
// HBM-bound, write-dominated: one lane per unknown point i loads its 3 (idx, weight) pairs once and
// walks a chunk of channels, so the (B,C,n) output is written in full coalesced rows while the gathers
// hit a (m-float) feature row that stays in L1/L2.  Backward: csr.hip (transpose + gather-sum, no atomics).
#include "tp3d_common.h"

namespace tp3d {

constexpr int TI_BLOCK = 256;
constexpr int TI_CCHUNK = 16;  // channels per workgroup

__global__ __launch_bounds__(TI_BLOCK) void three_interpolate_fwd_kernel(const float *__restrict__ feat,
                                                                          const int64_t *__restrict__ idx,
                                                                          const float *__restrict__ w, int C,
                                                                          int m, int n, float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int i = blockIdx.x * TI_BLOCK + threadIdx.x;
    if (i >= n) return;
    const size_t o = ((size_t)b * n + i) * 3;
    const int k0 = min(max((int)idx[o + 0], 0), m - 1);
    const int k1 = min(max((int)idx[o + 1], 0), m - 1);
    const int k2 = min(max((int)idx[o + 2], 0), m - 1);
    const float w0 = w[o + 0], w1 = w[o + 1], w2 = w[o + 2];
    const int c0 = blockIdx.y * TI_CCHUNK;
    const int c1 = min(c0 + TI_CCHUNK, C);
    for (int c = c0; c < c1; ++c) {
        const float *f = feat + ((size_t)b * C + c) * m;
        const float a0 = w0 * f[k0];
        const float a1 = w1 * f[k1];
        const float a2 = w2 * f[k2];
        out[((size_t)b * C + c) * n + i] = (a0 + a1) + a2;
    }
}

// Known sets of up to 32768 points: the feature rows of CC channels (m floats each) are staged in LDS and the three
// gathers per output come from there (from memory they touch up to 64 different cache lines per wave-instruction: the
// kernel above wrote its coalesced output at 1.4 TB/s).  One workgroup per (cloud, CC channels), four unknown points per
// thread: their 12 (index, weight) pairs once, then per channel 12 LDS reads and one float4 store.
constexpr int TL_BLOCK = 1024;
constexpr int TL_LDS_FLOATS = 32768;  // 128 KiB of feature rows per workgroup

template <int CC>
__global__ __launch_bounds__(TL_BLOCK) void three_interpolate_fwd_lds_kernel(const float *__restrict__ feat,
                                                                              const int64_t *__restrict__ idx,
                                                                              const float *__restrict__ w, int C, int m,
                                                                              int n, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float s_feat[];  // [CC][m]
    const int b = blockIdx.y, c0 = blockIdx.x * CC, tid = threadIdx.x;
    const int nc = min(CC, C - c0);
    const float *fb = feat + ((size_t)b * C + c0) * m;
    for (int e = tid; e < nc * m; e += TL_BLOCK) s_feat[e] = fb[e];
    __syncthreads();
    const int64_t *ib = idx + (size_t)b * n * 3;
    const float *wb = w + (size_t)b * n * 3;
    float *ob = out + ((size_t)b * C + c0) * n;
    const bool vec = (n & 3) == 0 && ((uintptr_t)ob & 15) == 0;
    for (int i0 = tid * 4; i0 < n; i0 += TL_BLOCK * 4) {
        int k[4][3];
        float wt[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u, n - 1);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                k[u][t] = min(max((int)ib[(size_t)i * 3 + t], 0), m - 1);
                wt[u][t] = wb[(size_t)i * 3 + t];
            }
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            if (cc >= nc) break;
            const float *f = s_feat + cc * m;
            float r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a0 = wt[u][0] * f[k[u][0]];
                const float a1 = wt[u][1] * f[k[u][1]];
                const float a2 = wt[u][2] * f[k[u][2]];
                r[u] = (a0 + a1) + a2;  // the oracle's order
            }
            float *o = ob + (size_t)cc * n + i0;
            if (vec) {
                *reinterpret_cast<float4 *>(o) = make_float4(r[0], r[1], r[2], r[3]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (i0 + u < n) o[u] = r[u];
            }
        }
    }
}

template <int CC>
static void launch_interp_lds(const float *features, const int64_t *idx, const float *weight, int B, int C, int m, int n,
                              float *out, hipStream_t s)
{
    static bool attr_set[64] = {false};
    allow_large_dynamic_lds(reinterpret_cast<const void *>(&three_interpolate_fwd_lds_kernel<CC>), TL_LDS_FLOATS * 4,
                            attr_set);
    hipLaunchKernelGGL(three_interpolate_fwd_lds_kernel<CC>, dim3((C + CC - 1) / CC, B), dim3(TL_BLOCK), (size_t)CC * m * 4, s,
                       features, idx, weight, C, m, n, out);
}

}  // namespace tp3d

TP3D_EXPORT int tp3d_three_interpolate_fwd_f32(const float *features, const int64_t *idx, const float *weight,
                                               int B, int C, int m, int n, float *out, void *stream)
{
    using namespace tp3d;
    if (B < 0 || C < 0 || m <= 0 || n < 0) return TP3D_E_BADARG;
    if (B == 0 || C == 0 || n == 0) return TP3D_OK;
    if (!features || !idx || !weight || !out) return TP3D_E_BADARG;
    if (B > 65535 || (C + TI_CCHUNK - 1) / TI_CCHUNK > 65535) return TP3D_E_TOOBIG;
    if (m <= TL_LDS_FLOATS && n >= 2048 && C <= 65535) {
        // as many channel rows per workgroup as fit 128 KiB (at most 16), but enough workgroups to fill the chip
        int cc = TL_LDS_FLOATS / m;
        cc = cc >= 16 ? 16 : (cc >= 8 ? 8 : (cc >= 4 ? 4 : (cc >= 2 ? 2 : 1)));
        while (cc > 1 && (int64_t)B * ((C + cc - 1) / cc) < 512) cc >>= 1;
        hipStream_t s = (hipStream_t)stream;
        switch (cc) {
        case 16: launch_interp_lds<16>(features, idx, weight, B, C, m, n, out, s); break;
        case 8: launch_interp_lds<8>(features, idx, weight, B, C, m, n, out, s); break;
        case 4: launch_interp_lds<4>(features, idx, weight, B, C, m, n, out, s); break;
        case 2: launch_interp_lds<2>(features, idx, weight, B, C, m, n, out, s); break;
        default: launch_interp_lds<1>(features, idx, weight, B, C, m, n, out, s); break;
        }
        return check_launch();
    }
    dim3 grid((n + TI_BLOCK - 1) / TI_BLOCK, (C + TI_CCHUNK - 1) / TI_CCHUNK, B);
    hipLaunchKernelGGL(three_interpolate_fwd_kernel, grid, dim3(TI_BLOCK), 0, (hipStream_t)stream, features, idx,
                       weight, C, m, n, out);
    return check_launch();
}

// backward: grad_features[b,c,k] = sum over the slots (i,t) with idx[b,i,t] == k of w[b,i,t]*grad_out[b,c,i],
// ascending (i,t) (csr.hip: transpose the (B, 3n) table once, then one weighted gather-sum per destination).
TP3D_EXPORT int tp3d_three_interpolate_bwd_f32(const float *grad_out, const int64_t *idx, const float *weight,
                                               int B, int C, int m, int n, float *grad_features, void *workspace,
                                               size_t workspace_bytes, void *stream)
{
    using namespace tp3d;
    if (B < 0 || C < 0 || m <= 0 || n < 0) return TP3D_E_BADARG;
    if (B == 0 || C == 0) return TP3D_OK;
    if (!grad_features) return TP3D_E_BADARG;
    if ((int64_t)n * 3 > INT32_MAX / 4 || B > 65535 || C > 65535 * 4) return TP3D_E_TOOBIG;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return zero_async(grad_features, (size_t)B * C * m * sizeof(float), s);
    if (!grad_out || !idx || !weight || !workspace) return TP3D_E_BADARG;
    const int L = n * 3;
    ScatterWorkspace w = carve_scatter_workspace(workspace, B, L, m, true);
    if (workspace_bytes < w.bytes) return TP3D_E_BADARG;
    if (int rc = csr_transpose(idx, B, L, m, 3, weight, w.start, w.order, w.wsorted, w.scratch, s)) return rc;
    return gather_sum(grad_out, w.start, w.order, w.wsorted, B, C, m, n, L, grad_features, s);
}
