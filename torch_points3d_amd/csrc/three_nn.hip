// three_nn -- for every unknown point the three nearest known points.
//
// Reference contract: torch_points3d/core/base_conv/dense.py:136 (DenseFPModule.conv); the returned
// distance is consumed as Euclidean (1/(dist+1e-8), dense.py:137).  Semantics SURVEY.md 8a-H8;
// oracle tpk_ref_three_nn_f32: the three smallest fp32 squared distances (dx*dx + dy*dy) + dz*dz, closest first,
// the lower index first among equal distances.
//
// Two kernels.
//  * three_nn_kernel: one lane per unknown point, the known cloud streams through an LDS tile that every lane reads at
//    the same address (broadcast).  Every (unknown, known) pair is tested: ~17 instructions per pair, 117 us at
//    n = 16384, m = 512, B = 32.  Serves small and very large known clouds.
//  * three_nn_grid_kernel (64 <= m <= 896 known points, the decoder shapes of PointNet++): each workgroup first bins
//    the known cloud of its batch element into a uniform grid IN LDS (bounding box -> ~2 points per cell -> histogram
//    -> scan -> cell-ordered float4 lists; a few microseconds for 512 points), then every lane walks the 3x3x3 cells
//    around its own unknown point instead of the whole cloud: ~34 tests instead of 512.  The search is exact: a lane
//    stops only when its third-best distance lies inside the largest ball certainly covered by the visited block
//    (distance to the nearest block face that still has cells behind it, shrunk by 0.1 % for the fp32 rounding of the
//    cell coordinate); otherwise it visits a wider block, then the whole grid.  Candidates arrive out of index order,
//    so the running top three are kept as 64-bit keys (distance bits << 32 | index): one unsigned compare orders by
//    (distance, index) -- squared distances are non-negative floats, whose bit patterns order like the values -- and
//    the result is bit-identical to the scan.
#include "grid.h"
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
            if (d < b3) {  // strict '<' in ascending index order keeps the lowest index on ties
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
        dist[o + 0] = sqrtf(b1);  // correctly rounded (the fast __fsqrt_rn form is 1 ulp off the oracle's sqrtf)
        dist[o + 1] = sqrtf(b2);
        dist[o + 2] = sqrtf(b3);
        idx[o + 0] = i1;
        idx[o + 1] = i2;
        idx[o + 2] = i3;
    }
}

#ifndef TP3D_NG_BLOCK  // tuning knobs of tools/exp_three_nn.py; the defaults are the measured best
#define TP3D_NG_BLOCK 1024
#define TP3D_NG_QPT 2
#define TP3D_NG_TARGET 2.0f
#endif
constexpr int NG_BLOCK = TP3D_NG_BLOCK;
constexpr int NG_QPT = TP3D_NG_QPT;  // unknown points per thread: a workgroup serves NG_BLOCK * NG_QPT with one grid build
constexpr int NG_MAX_KNOWN = 896;    // 9 copies of every known point as float4 + the cell tables must fit 160 KiB of LDS
constexpr int NG_MIN_KNOWN = 64;
constexpr int NG_MAX_CELLS = 1024;
constexpr int NG_AXIS = 16;
constexpr float NG_TARGET = TP3D_NG_TARGET;  // known points per cell of the bounding box
constexpr unsigned long long NG_EMPTY = 0x7f8000007fffffffull;  // (+inf, INT_MAX)

struct NNGrid {
    float minx, miny, minz, cs, inv_cs;
    int gx, gy, gz;
};

__host__ __device__ constexpr size_t nng_lds_bytes(int m) { return ((size_t)9 * m + 1) * 16 + (size_t)(3 * NG_MAX_CELLS + 4) * 4; }

__device__ __forceinline__ void nn_push(unsigned long long key, unsigned long long &k1, unsigned long long &k2,
                                        unsigned long long &k3)
{
    const bool c1 = key < k1, c2 = key < k2, c3 = key < k3;
    k3 = c2 ? k2 : (c3 ? key : k3);
    k2 = c1 ? k1 : (c2 ? key : k2);
    k1 = c1 ? key : k1;
}

// Known points per workgroup are binned into "super rows": the list of cell (cx, cy, cz) holds, ordered by cx, every
// known point of the nine grid rows (cy-1..cy+1, cz-1..cz+1) whose x cell is cx.  The 27 cells around a query are then
// ONE contiguous run [start(cx-1, cy, cz), start(cx+2, cy, cz)): a lane's walk is a single loop with no per-row set-up,
// and a wave runs as long as its longest lane's 27-cell population instead of the sum of nine per-row maxima
// (51 instead of 73 iterations at m = 512).  Price: nine copies of the known cloud in LDS (72 KiB for 512 points).
__global__ __launch_bounds__(NG_BLOCK) void three_nn_grid_kernel(const float *__restrict__ unknown,
                                                                  const float *__restrict__ known, int n, int m,
                                                                  float *__restrict__ dist, int64_t *__restrict__ idx)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *spt = reinterpret_cast<float4 *>(smem);                           // [9 m + 1] super-row entries, .w = index bits
    int *scnt = reinterpret_cast<int *>(smem + ((size_t)9 * m + 1) * 16);    // [NG_MAX_CELLS] points per cell
    int *sstart = scnt + NG_MAX_CELLS;                                        // [NG_MAX_CELLS + 1] first entry of a list
    int *scur = sstart + NG_MAX_CELLS + 4;                                    // [NG_MAX_CELLS] fill cursors
    __shared__ float sred[6][NG_BLOCK / 64];
    __shared__ NNGrid sgrid;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *kb = known + (size_t)b * m * 3;

    // ---- bounding box of the known cloud
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int j = tid; j < m; j += NG_BLOCK)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = kb[(size_t)j * 3 + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off));
        }
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            sred[a][wave] = mn[a];
            sred[3 + a][wave] = mx[a];
        }
    for (int k = tid; k < NG_MAX_CELLS; k += NG_BLOCK) scnt[k] = 0;
    __syncthreads();
    if (tid == 0) {
        float lo[3], e[3];
        for (int a = 0; a < 3; ++a) {
            float l = sred[a][0], h = sred[3 + a][0];
            for (int w = 1; w < NG_BLOCK / 64; ++w) {
                l = fminf(l, sred[a][w]);
                h = fmaxf(h, sred[3 + a][w]);
            }
            lo[a] = l;
            e[a] = h - l;
        }
        const float ext = fmaxf(fmaxf(e[0], e[1]), e[2]);
        const float eps = ext * 1.0e-3f;
        const float vol = fmaxf(e[0], eps) * fmaxf(e[1], eps) * fmaxf(e[2], eps);
        float cs = fmaxf(cbrtf(vol * NG_TARGET / (float)m), ext / (float)NG_AXIS * 1.0001f);
        if (!(cs > 0.0f)) cs = 1.0f;  // every known point at the same place: one cell
        NNGrid g;
        for (;;) {
            g.inv_cs = 1.0f / cs;
            g.gx = min(NG_AXIS, (int)floorf(e[0] * g.inv_cs) + 1);
            g.gy = min(NG_AXIS, (int)floorf(e[1] * g.inv_cs) + 1);
            g.gz = min(NG_AXIS, (int)floorf(e[2] * g.inv_cs) + 1);
            if (g.gx * g.gy * g.gz <= NG_MAX_CELLS) break;
            cs *= 1.26f;
        }
        g.minx = lo[0];
        g.miny = lo[1];
        g.minz = lo[2];
        g.cs = cs;
        sgrid = g;
    }
    __syncthreads();
    const NNGrid g = sgrid;
    const int ncells = g.gx * g.gy * g.gz;
    auto cell_of = [&](int j, int &cx, int &cy, int &cz) {
        cx = cell_coord(kb[(size_t)j * 3 + 0], g.minx, g.inv_cs, g.gx);
        cy = cell_coord(kb[(size_t)j * 3 + 1], g.miny, g.inv_cs, g.gy);
        cz = cell_coord(kb[(size_t)j * 3 + 2], g.minz, g.inv_cs, g.gz);
    };
    for (int j = tid; j < m; j += NG_BLOCK) {
        int cx, cy, cz;
        cell_of(j, cx, cy, cz);
        atomicAdd(&scnt[(cz * g.gy + cy) * g.gx + cx], 1);
    }
    __syncthreads();
    for (int c = tid; c < ncells; c += NG_BLOCK) {  // population of the super-row list of cell c
        const int cx = c % g.gx, cy = (c / g.gx) % g.gy, cz = c / (g.gx * g.gy);
        int sum = 0;
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy) {
                const int zz = cz + dz, yy = cy + dy;
                if (zz >= 0 && zz < g.gz && yy >= 0 && yy < g.gy) sum += scnt[(zz * g.gy + yy) * g.gx + cx];
            }
        scur[c] = sum;
    }
    __syncthreads();
    if (wave == 0) {  // exclusive scan of <= 1024 counters by one wave
        const int per = (ncells + 63) / 64;
        const int k0 = min(lane * per, ncells), k1 = min(k0 + per, ncells);
        int sum = 0;
        for (int k = k0; k < k1; ++k) sum += scur[k];
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        int run = incl - sum;
        for (int k = k0; k < k1; ++k) {
            const int v = scur[k];
            scur[k] = run;
            sstart[k] = run;
            run += v;
        }
        if (lane == 63) sstart[ncells] = incl;
    }
    __syncthreads();
    for (int j = tid; j < m; j += NG_BLOCK) {
        int cx, cy, cz;
        cell_of(j, cx, cy, cz);
        const float4 pt = make_float4(kb[(size_t)j * 3 + 0], kb[(size_t)j * 3 + 1], kb[(size_t)j * 3 + 2], __int_as_float(j));
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy) {
                const int zz = cz + dz, yy = cy + dy;
                if (zz >= 0 && zz < g.gz && yy >= 0 && yy < g.gy)  // order inside a list segment: whatever the atomics
                    spt[atomicAdd(&scur[(zz * g.gy + yy) * g.gx + cx], 1)] = pt;  // give -- the keys rank the hits
            }
    }
    __syncthreads();

    // ---- queries
#pragma unroll 1
    for (int qq = 0; qq < NG_QPT; ++qq) {
        const int i = (blockIdx.x * NG_QPT + qq) * NG_BLOCK + tid;
        const bool ok = i < n;
        const size_t u = ((size_t)b * n + (ok ? i : 0)) * 3;
        const float ux = unknown[u + 0], uy = unknown[u + 1], uz = unknown[u + 2];
        const int cx = cell_coord(ux, g.minx, g.inv_cs, g.gx);
        const int cy = cell_coord(uy, g.miny, g.inv_cs, g.gy);
        const int cz = cell_coord(uz, g.minz, g.inv_cs, g.gz);
        unsigned long long k1 = NG_EMPTY, k2 = NG_EMPTY, k3 = NG_EMPTY;
        {   // the 3x3x3 block: one run of the super-row list
            const int rowbase = (cz * g.gy + cy) * g.gx;
            const int j0 = sstart[rowbase + max(cx - 1, 0)];
            const int j1 = ok ? sstart[rowbase + min(cx + 1, g.gx - 1) + 1] : j0;  // lanes past the cloud walk nothing
            float4 p = spt[j0];
            for (int j = j0; j < j1; ++j) {
                const float4 nxt = spt[j + 1];  // requested before this point's arithmetic: the LDS latency overlaps it
                const float d = sqdist3(p.x, p.y, p.z, ux, uy, uz);
                nn_push(((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w), k1, k2, k3);
                p = nxt;
            }
        }
        // Is the third-best certainly inside the 3x3x3 block?  Distance to the nearest face that still has cells behind it.
        float fd = 3.0e38f;
        if (cx - 1 > 0) fd = fminf(fd, ux - (g.minx + (float)(cx - 1) * g.cs));
        if (cx + 1 < g.gx - 1) fd = fminf(fd, (g.minx + (float)(cx + 2) * g.cs) - ux);
        if (cy - 1 > 0) fd = fminf(fd, uy - (g.miny + (float)(cy - 1) * g.cs));
        if (cy + 1 < g.gy - 1) fd = fminf(fd, (g.miny + (float)(cy + 2) * g.cs) - uy);
        if (cz - 1 > 0) fd = fminf(fd, uz - (g.minz + (float)(cz - 1) * g.cs));
        if (cz + 1 < g.gz - 1) fd = fminf(fd, (g.minz + (float)(cz + 2) * g.cs) - uz);
        fd = fmaxf(fd, 0.0f);  // 3.0e38^2 = +inf: a block that covers the grid always passes
        const bool open = ok && !(__uint_as_float((unsigned)(k3 >> 32)) < fd * fd * 0.998f);
        // The few lanes that fail (2 in 10^4 when the known points are a furthest-point subset) are served one after the
        // other by the WHOLE wave: 64 known points per step straight from the cloud, then a butterfly merge of the 64
        // partial top-threes.  A lane walking wider blocks on its own would hold its wave -- and the kernel -- for tens
        // of microseconds.
        unsigned long long todo = __ballot(open);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const float qx = __shfl(ux, src), qy = __shfl(uy, src), qz = __shfl(uz, src);
            unsigned long long t1 = NG_EMPTY, t2 = NG_EMPTY, t3 = NG_EMPTY;
            for (int j = lane; j < m; j += 64) {
                const float d = sqdist3(kb[(size_t)j * 3 + 0], kb[(size_t)j * 3 + 1], kb[(size_t)j * 3 + 2], qx, qy, qz);
                nn_push(((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j, t1, t2, t3);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {  // keys are unique per known point: no duplicates to guard against
                const unsigned long long o1 = __shfl_xor(t1, off), o2 = __shfl_xor(t2, off), o3 = __shfl_xor(t3, off);
                nn_push(o1, t1, t2, t3);
                nn_push(o2, t1, t2, t3);
                nn_push(o3, t1, t2, t3);
            }
            if (lane == src) {
                k1 = t1;
                k2 = t2;
                k3 = t3;
            }
        }
        if (ok) {
            const size_t o = ((size_t)b * n + i) * 3;
            dist[o + 0] = sqrtf(__uint_as_float((unsigned)(k1 >> 32)));
            dist[o + 1] = sqrtf(__uint_as_float((unsigned)(k2 >> 32)));
            dist[o + 2] = sqrtf(__uint_as_float((unsigned)(k3 >> 32)));
            idx[o + 0] = (int64_t)(unsigned)k1;
            idx[o + 1] = (int64_t)(unsigned)k2;
            idx[o + 2] = (int64_t)(unsigned)k3;
        }
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
    if (m >= NG_MIN_KNOWN && m <= NG_MAX_KNOWN && n >= 4 * m && n >= 1024) {
        // enough unknown points per known one to pay for binning the known cloud in every workgroup
        static bool attr_set[64] = {false};
        allow_large_dynamic_lds(reinterpret_cast<const void *>(&three_nn_grid_kernel), (int)nng_lds_bytes(NG_MAX_KNOWN),
                                attr_set);
        dim3 grid((n + NG_BLOCK * NG_QPT - 1) / (NG_BLOCK * NG_QPT), B);
        hipLaunchKernelGGL(three_nn_grid_kernel, grid, dim3(NG_BLOCK), nng_lds_bytes(m), (hipStream_t)stream, unknown,
                           known, n, m, dist, idx);
        return check_launch();
    }
    dim3 grid((n + NN_BLOCK - 1) / NN_BLOCK, B);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(NN_BLOCK), 0, (hipStream_t)stream, unknown, known, n, m, dist,
                       idx);
    return check_launch();
}
