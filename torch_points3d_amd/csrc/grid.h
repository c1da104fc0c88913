// Uniform grid over the support points of each cloud: shared by the radius search (grid.hip) and the exact kNN
// (knn.hip).  Two builders fill the same tables:
//   * clouds of <= 65536 points: one workgroup per cloud, everything in LDS (grid.hip: grid_build_kernel);
//   * larger clouds: bounding box with atomics -> (cloud, cell) keys -> one rocPRIM radix sort -> binary-searched
//     cell starts (grid.hip: gridg_* kernels); any cloud size, any number of workgroups.
#pragma once
#include "tp3d_common.h"

namespace tp3d {

struct GridInfo {  // per cloud, 8 words
    float minx, miny, minz, inv_cs;
    int gx, gy, gz, pad;
};

__device__ __forceinline__ int cell_coord(float x, float mn, float inv_cs, int g)
{
    int c = (int)floorf((x - mn) * inv_cs);
    return min(max(c, 0), g - 1);
}

constexpr int GRID_LDS_MAX_POINTS = 65536;  // u16 slots of the in-LDS build
constexpr int GRID_GLOBAL_MAX_EDGE = 128;

struct GridPlan {
    int G;          // cells per axis the tables are sized for (0: no grid possible)
    bool global;    // sort-based build
};
GridPlan grid_plan(int Lmax);

struct GridWorkspace {
    GridInfo *info;    // [clouds]
    int *cell_start;   // [clouds][G^3 + 1], offsets relative to the cloud's first row
    float4 *sorted_pt; // [rows] cell-ordered copy of the points: (x, y, z, cloud-local point id as int bits) --
                       // one 16-byte load per candidate in the query kernels
    // sort-based build only
    int *bbox;                              // [clouds][6] order-preserving int images of min/max
    unsigned long long *keys_in, *keys_out; // [rows]
    unsigned int *vals_in, *vals_out;       // [rows]
    void *sort_tmp;
    size_t sort_tmp_bytes;
    size_t bytes;
};
GridWorkspace carve_grid_workspace(void *ws, int num_clouds, int64_t rows, GridPlan plan);

// Enqueue the build.  seg == nullptr: dense layout (cloud b = rows [b*N, (b+1)*N)).  `cell` is the smallest cell
// edge wanted (the search radius for ball queries); <= 0 lets the kernel pick one for `target` points per cell.
int grid_build(const float *x, const int64_t *seg, int num_clouds, int64_t rows, int N, int Lmax, float cell, float target,
               GridPlan plan, const GridWorkspace &w, hipStream_t s);

// voxel.hip: the one rocPRIM radix sort instantiation of the library (u64 keys, u32 values, stable)
size_t sort_pairs_tmp_bytes(int64_t n);
int sort_pairs_u64_u32(void *tmp, size_t tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                       const unsigned int *vals_in, unsigned int *vals_out, int64_t n, unsigned bits, hipStream_t s);

int grid_knn(const float *x, const float *y, const int64_t *seg, const int64_t *batch_y, int num_clouds, int64_t rows,
             int N, int np, int64_t total_q, int Lmax, int k, float cell, int64_t *idx, float *dist2, void *workspace,
             size_t workspace_bytes, hipStream_t s);

}  // namespace tp3d
