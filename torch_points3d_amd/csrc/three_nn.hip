// three_nn -- for every unknown point the three nearest known points.
//
// Reference contract: torch_points3d/core/base_conv/dense.py:136 (DenseFPModule.conv); the returned
// distance is consumed as Euclidean (1/(dist+1e-8), dense.py:137).  Semantics SURVEY.md 8a-H8;
// oracle tpk_ref_three_nn_f32.
//
// One lane per unknown point; the known cloud streams through an LDS tile that every lane reads at the
// same address (broadcast, conflict-free).  Strict '<' insertion keeps the lowest index on ties.
#include "tp3d_common.h"

namespace tp3d {

constexpr int NN_BLOCK = 256;
constexpr int NN_TILE = 1024;  // known points per LDS tile (16 KiB as float4)

__global__ __launch_bounds__(NN_BLOCK) void three_nn_kernel(const float *__restrict__ unknown,
                                                             const float *__restrict__ known, int n, int m,
                                                             float *__restrict__ dist, int64_t *__restrict__ idx)
{
    __shared__ float4 sk[NN_TILE];
    const int b = blockIdx.y;
    const int i = blockIdx.x * NN_BLOCK + threadIdx.x;
    const bool ok = i < n;
    const float *kb = known + (size_t)b * m * 3;
    const size_t u = ((size_t)b * n + (ok ? i : 0)) * 3;
    const float ux = unknown[u + 0], uy = unknown[u + 1], uz = unknown[u + 2];

    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int base = 0; base < m; base += NN_TILE) {
        const int tcnt = min(NN_TILE, m - base);
        for (int e = threadIdx.x; e < tcnt; e += NN_BLOCK) {
            const float *kp = kb + (size_t)(base + e) * 3;
            sk[e] = make_float4(kp[0], kp[1], kp[2], 0.0f);
        }
        __syncthreads();
        for (int k = 0; k < tcnt; ++k) {
            const float4 p = sk[k];
            const float d = sqdist3(p.x, p.y, p.z, ux, uy, uz);
            if (d < b3) {
                const int kk = base + k;
                if (d < b1) {
                    b3 = b2; i3 = i2;
                    b2 = b1; i2 = i1;
                    b1 = d;  i1 = kk;
                } else if (d < b2) {
                    b3 = b2; i3 = i2;
                    b2 = d;  i2 = kk;
                } else {
                    b3 = d;  i3 = kk;
                }
            }
        }
        __syncthreads();
    }
    if (ok) {
        const size_t o = ((size_t)b * n + i) * 3;
        dist[o + 0] = __fsqrt_rn(b1);
        dist[o + 1] = __fsqrt_rn(b2);
        dist[o + 2] = __fsqrt_rn(b3);
        idx[o + 0] = i1;
        idx[o + 1] = i2;
        idx[o + 2] = i3;
    }
}

}  // namespace tp3d

TP3D_EXPORT int tp3d_three_nn_f32(const float *unknown, const float *known, int B, int n, int m, float *dist,
                                  int64_t *idx, void *stream)
{
    using namespace tp3d;
    if (B < 0 || n < 0 || m < 3) return TP3D_E_BADARG;
    if (B == 0 || n == 0) return TP3D_OK;
    if (!unknown || !known || !dist || !idx) return TP3D_E_BADARG;
    if ((int64_t)n * 3 > INT32_MAX || (int64_t)m * 3 > INT32_MAX || B > 65535) return TP3D_E_TOOBIG;
    dim3 grid((n + NN_BLOCK - 1) / NN_BLOCK, B);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(NN_BLOCK), 0, (hipStream_t)stream, unknown, known, n, m, dist,
                       idx);
    return check_launch();
}
