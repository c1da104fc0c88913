/*
 * tp3d_cpu.h -- C-ABI of libtp3d_cpu.so: the HOST-side neighbour searches torch-points3d's data transforms and dataset
 * builders import as `torch_points_kernels.points_cpu` (reference core/data_transform/transforms.py:16,805,853,887,
 * 890,919,1044; datasets/registration/utils.py:8,150-166,286; datasets/registration/base_siamese_dataset.py:133-135;
 * datasets/registration/basetest.py:33,361).  In the reference these are torch-points-kernels 0.7.0's nanoflann
 * KD-tree searches; here a uniform grid over the support cloud serves both.  Plain C, no GPU runtime, no global state,
 * no persistent threads (worker threads are created and joined inside a call), so the library is safe to use in forked
 * DataLoader workers (datasets/base_dataset.py:251-263).
 *
 * Conventions: row-major float32 xyz triples, int64 indices, squared distances evaluated as (dx*dx + dy*dy) + dz*dz
 * without fused multiply-add; a point is inside a ball when d2 < radius*radius (strict).  Host memory is owned by the
 * caller; the only object the library allocates is the grid handle.  Return 0 / negative TP3D_CPU_E_* code.
 */
#ifndef TP3D_CPU_H
#define TP3D_CPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TP3D_CPU_OK 0
#define TP3D_CPU_E_BADARG (-1)
#define TP3D_CPU_E_NOMEM (-2)

int tp3d_cpu_abi_version(void);

/* Uniform grid over `n` support points with cells of about `cell` (> 0) edge; the points are borrowed and must stay
 * valid while the handle lives. */
void *tp3d_cpu_grid_build(const float *points, int64_t n, float cell);
void tp3d_cpu_grid_free(void *grid);

/* Radius search, two passes so that the caller allocates the result:
 *   count: counts[q] = number of support points with d2 < radius^2 around query q;
 *   fill : per query its hits in ascending support index (sorted == 0) or ascending (d2, index) (sorted != 0), at most
 *          `limit` of them (limit <= 0: all); hit j of query q goes to slot offsets[q] + j of idx / dist2.  Slots
 *          offsets[q] + hits .. offsets[q+1] - 1 (padding of the matrix layout) receive -1 / -1.0f.
 * threads <= 0: one. */
int tp3d_cpu_ball_count(const void *grid, const float *query, int64_t nq, float radius, int64_t *counts, int threads);
int tp3d_cpu_ball_fill(const void *grid, const float *query, int64_t nq, float radius, int limit, int sorted,
                       const int64_t *offsets, int64_t *idx, float *dist2, int threads);

/* Exact k nearest neighbours of every query: idx / dist2 (nq, k), ascending (d2, index); slots a cloud of fewer than k
 * points cannot fill receive -1 / -1.0f. */
int tp3d_cpu_knn(const void *grid, const float *query, int64_t nq, int k, int64_t *idx, float *dist2, int threads);

/* Region growing over a fixed-width neighbour table (rows end at their first -1): the clusters reached from the lowest
 * unvisited point through the table's edges, members in discovery order, clusters of fewer than min_size points dropped.
 * Replaces the numba loop behind torch_points_kernels.region_grow (reference models/panoptic/pointgroup.py:101-115).
 * members: n slots, cluster_start: n + 1 slots.  Returns the number of clusters kept, or a negative error. */
int64_t tp3d_cpu_grow_clusters(const int64_t *neighbours, int64_t n, int width, int64_t min_size, int64_t *members,
                               int64_t *cluster_start);

#ifdef __cplusplus
}
#endif
#endif /* TP3D_CPU_H */
