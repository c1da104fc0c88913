// grouping_operation forward / backward.
//
// Reference contract: torch_points3d/modules/pointnet2/dense.py:38,45 (PointNetMSGDown._prepare_features);
// semantics SURVEY.md 8a-H5; oracle tpk_ref_group_{fwd,bwd}_f32.
//
// HBM-bound copy: one lane per (centroid, sample) slot loads its int64 index once and walks a chunk of
// channels; the (B,C,np,ns) output is written in coalesced rows, the gathers hit an N-float feature row.
#include "tp3d_common.h"

namespace tp3d {

constexpr int GR_BLOCK = 256;
constexpr int GR_CCHUNK = 16;

__global__ __launch_bounds__(GR_BLOCK) void group_fwd_kernel(const float *__restrict__ feat,
                                                              const int64_t *__restrict__ idx, int C, int N,
                                                              int L, float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int l = blockIdx.x * GR_BLOCK + threadIdx.x;
    if (l >= L) return;
    const int k = min(max((int)idx[(size_t)b * L + l], 0), N - 1);
    const int c0 = blockIdx.y * GR_CCHUNK;
    const int c1 = min(c0 + GR_CCHUNK, C);
    for (int c = c0; c < c1; ++c) out[((size_t)b * C + c) * L + l] = feat[((size_t)b * C + c) * N + k];
}

}  // namespace tp3d

TP3D_EXPORT int tp3d_group_fwd_f32(const float *features, const int64_t *idx, int B, int C, int N, int np, int ns,
                                   float *out, void *stream)
{
    using namespace tp3d;
    if (B < 0 || C < 0 || N <= 0 || np < 0 || ns < 0) return TP3D_E_BADARG;
    const int64_t L64 = (int64_t)np * ns;
    if (B == 0 || C == 0 || L64 == 0) return TP3D_OK;
    if (!features || !idx || !out) return TP3D_E_BADARG;
    if (L64 > INT32_MAX || B > 65535 || (C + GR_CCHUNK - 1) / GR_CCHUNK > 65535) return TP3D_E_TOOBIG;
    const int L = (int)L64;
    dim3 grid((L + GR_BLOCK - 1) / GR_BLOCK, (C + GR_CCHUNK - 1) / GR_CCHUNK, B);
    hipLaunchKernelGGL(group_fwd_kernel, grid, dim3(GR_BLOCK), 0, (hipStream_t)stream, features, idx, C, N, L, out);
    return check_launch();
}

// backward: grad_features[b,c,k] = sum over the slots l with idx[b,l] == k of grad_out[b,c,l], ascending l
// (csr.hip: transpose the table once, then one gather-sum per destination -- no atomics, reproducible).
TP3D_EXPORT int tp3d_group_bwd_f32(const float *grad_out, const int64_t *idx, int B, int C, int N, int np, int ns,
                                   float *grad_features, void *workspace, size_t workspace_bytes, void *stream)
{
    using namespace tp3d;
    if (B < 0 || C < 0 || N <= 0 || np < 0 || ns < 0) return TP3D_E_BADARG;
    if (B == 0 || C == 0) return TP3D_OK;
    if (!grad_features) return TP3D_E_BADARG;
    const int64_t L64 = (int64_t)np * ns;
    if (L64 > INT32_MAX / 4 || B > 65535 || C > 65535 * 4) return TP3D_E_TOOBIG;
    hipStream_t s = (hipStream_t)stream;
    const int L = (int)L64;
    if (L == 0) return zero_async(grad_features, (size_t)B * C * N * sizeof(float), s);
    if (!grad_out || !idx || !workspace) return TP3D_E_BADARG;
    ScatterWorkspace w = carve_scatter_workspace(workspace, B, L, N, false);
    if (workspace_bytes < w.bytes) return TP3D_E_BADARG;
    if (int rc = csr_transpose(idx, B, L, N, 1, nullptr, w.start, w.order, nullptr, w.scratch, s)) return rc;
    return gather_sum(grad_out, w.start, w.order, nullptr, B, C, N, L, L, grad_features, s);
}
