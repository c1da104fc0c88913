// three_nn -- for every unknown point the three nearest known points.
//
// Reference contract: torch_points3d/core/base_conv/dense.py:136 (DenseFPModule.conv); the returned
// distance is consumed as Euclidean (1/(dist+1e-8), dense.py:137).  Semantics SURVEY.md 8a-H8;
// oracle tpk_ref_three_nn_f32.
//
// One lane per unknown point; the known cloud streams through an LDS tile (three coordinate arrays) that every lane
// reads at the same address (broadcast, conflict-free).  FOUR known points per step: their squared distances are
// formed together (the compiler packs the pairs into v_pk_* instructions; each value is still (dx*dx + dy*dy) + dz*dz
// in fp32, so results stay bit-identical) and ONE comparison of their minimum against the current third-best decides
// whether the ordered insertion runs at all -- after the first few dozen points it almost never does, so a test costs
// ~6 instructions instead of ~15 (distance + compare + divergent branch per point: 117 us at n = 16384, m = 512, B = 32).
// Strict '<' insertion in ascending index order keeps the lowest index on ties.
#include "tp3d_common.h"

namespace tp3d {

constexpr int NN_BLOCK = 256;
constexpr int NN_TILE = 1024;  // known points per LDS tile (3 x 4 KiB)

__device__ __forceinline__ void nn_insert(float d, int kk, float &b1, float &b2, float &b3, int &i1, int &i2, int &i3)
{
    if (d < b3) {
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

__global__ __launch_bounds__(NN_BLOCK) void three_nn_kernel(const float *__restrict__ unknown,
                                                             const float *__restrict__ known, int n, int m,
                                                             float *__restrict__ dist, int64_t *__restrict__ idx)
{
    __shared__ __attribute__((aligned(16))) float sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
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
        const int tpad = (tcnt + 3) & ~3;
        for (int e = threadIdx.x; e < tpad; e += NN_BLOCK) {
            // slots past the cloud hold +inf: their distance is +inf, never below the running third-best
            const float *kp = kb + (size_t)(base + min(e, tcnt - 1)) * 3;
            const bool in = e < tcnt;
            sx[e] = in ? kp[0] : INFINITY;
            sy[e] = in ? kp[1] : INFINITY;
            sz[e] = in ? kp[2] : INFINITY;
        }
        __syncthreads();
        for (int k = 0; k < tpad; k += 4) {
            const float4 X = *reinterpret_cast<const float4 *>(&sx[k]);
            const float4 Y = *reinterpret_cast<const float4 *>(&sy[k]);
            const float4 Z = *reinterpret_cast<const float4 *>(&sz[k]);
            const float d0 = sqdist3(X.x, Y.x, Z.x, ux, uy, uz), d1 = sqdist3(X.y, Y.y, Z.y, ux, uy, uz);
            const float d2 = sqdist3(X.z, Y.z, Z.z, ux, uy, uz), d3 = sqdist3(X.w, Y.w, Z.w, ux, uy, uz);
            if (fminf(fminf(d0, d1), fminf(d2, d3)) < b3) {
                const int kk = base + k;
                nn_insert(d0, kk + 0, b1, b2, b3, i1, i2, i3);
                nn_insert(d1, kk + 1, b1, b2, b3, i1, i2, i3);
                nn_insert(d2, kk + 2, b1, b2, b3, i1, i2, i3);
                nn_insert(d3, kk + 3, b1, b2, b3, i1, i2, i3);
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
