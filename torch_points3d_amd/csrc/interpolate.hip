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

}  // namespace tp3d

TP3D_EXPORT int tp3d_three_interpolate_fwd_f32(const float *features, const int64_t *idx, const float *weight,
                                               int B, int C, int m, int n, float *out, void *stream)
{
    using namespace tp3d;
    if (B < 0 || C < 0 || m <= 0 || n < 0) return TP3D_E_BADARG;
    if (B == 0 || C == 0 || n == 0) return TP3D_OK;
    if (!features || !idx || !weight || !out) return TP3D_E_BADARG;
    if (B > 65535 || (C + TI_CCHUNK - 1) / TI_CCHUNK > 65535) return TP3D_E_TOOBIG;
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
