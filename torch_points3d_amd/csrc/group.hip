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

// Clouds of up to 32768 points: the feature row of a channel (N floats) fits LDS, and gathering from LDS costs a few
// cycles per wave where the same gather from memory touches 64 different cache lines per wave-instruction (the kernel above
// wrote its coalesced output at 0.9 TB/s on 16384-point clouds).  One workgroup per (cloud, CC channels): stage the CC rows
// (coalesced), then walk the L slots four per thread -- the indices as two 16-byte loads, one float4 store per channel.
constexpr int GL_BLOCK = 1024;
constexpr int GL_LDS_FLOATS = 32768;  // 128 KiB of feature rows per workgroup

template <int CC>
__global__ __launch_bounds__(GL_BLOCK) void group_fwd_lds_kernel(const float *__restrict__ feat,
                                                                  const int64_t *__restrict__ idx, int C, int N, int L,
                                                                  float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float s_rows[];  // [CC][N]
    const int b = blockIdx.y, c0 = blockIdx.x * CC, tid = threadIdx.x;
    const int nc = min(CC, C - c0);
    const float *fb = feat + ((size_t)b * C + c0) * N;
    for (int e = tid; e < nc * N; e += GL_BLOCK) s_rows[e] = fb[e];
    __syncthreads();
    const int64_t *ib = idx + (size_t)b * L;
    float *ob = out + ((size_t)b * C + c0) * L;
    const int L4 = L & ~3;
    const bool vec = (((uintptr_t)ob | (uintptr_t)ib) & 15) == 0 && (L & 3) == 0;  // every channel row 16-byte aligned
    if (vec) {
        for (int l = tid * 4; l < L4; l += GL_BLOCK * 4) {
            const longlong2 i01 = *reinterpret_cast<const longlong2 *>(ib + l);
            const longlong2 i23 = *reinterpret_cast<const longlong2 *>(ib + l + 2);
            const int k0 = min(max((int)i01.x, 0), N - 1), k1 = min(max((int)i01.y, 0), N - 1);
            const int k2 = min(max((int)i23.x, 0), N - 1), k3 = min(max((int)i23.y, 0), N - 1);
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
                if (cc < nc) {
                    const float *r = s_rows + cc * N;
                    *reinterpret_cast<float4 *>(ob + (size_t)cc * L + l) = make_float4(r[k0], r[k1], r[k2], r[k3]);
                }
        }
    } else {
        for (int l = tid; l < L; l += GL_BLOCK) {
            const int k = min(max((int)ib[l], 0), N - 1);
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
                if (cc < nc) ob[(size_t)cc * L + l] = s_rows[cc * N + k];
        }
    }
}

template <int CC>
static void launch_group_lds(const float *features, const int64_t *idx, int B, int C, int N, int L, float *out, hipStream_t s)
{
    static bool attr_set[64] = {false};
    allow_large_dynamic_lds(reinterpret_cast<const void *>(&group_fwd_lds_kernel<CC>), GL_LDS_FLOATS * 4, attr_set);
    hipLaunchKernelGGL(group_fwd_lds_kernel<CC>, dim3((C + CC - 1) / CC, B), dim3(GL_BLOCK), (size_t)CC * N * 4, s, features,
                       idx, C, N, L, out);
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
    if (N <= GL_LDS_FLOATS && L >= 4096 && C <= 65535) {
        // as many channel rows per workgroup as fit 128 KiB, but enough workgroups to fill the chip
        int cc = GL_LDS_FLOATS / N;
        cc = cc >= 8 ? 8 : (cc >= 4 ? 4 : (cc >= 2 ? 2 : 1));
        while (cc > 1 && (int64_t)B * ((C + cc - 1) / cc) < 512) cc >>= 1;
        hipStream_t s = (hipStream_t)stream;
        if (cc == 8) launch_group_lds<8>(features, idx, B, C, N, L, out, s);
        else if (cc == 4) launch_group_lds<4>(features, idx, B, C, N, L, out, s);
        else if (cc == 2) launch_group_lds<2>(features, idx, B, C, N, L, out, s);
        else launch_group_lds<1>(features, idx, B, C, N, L, out, s);
        return check_launch();
    }
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
