/*
 * tp3d_hip.h -- C-ABI of libtp3d_hip.so, the MI355X (gfx950) implementation of the
 * torch_points_kernels hot path used by torch-points3d.
 *
 * The reference has no FFI table for this path: it binds the third-party Python package
 * `torch_points_kernels` by import name (reference torch_points3d/core/spatial_ops/sampling.py:7,
 * core/spatial_ops/neighbour_finder.py:5, core/base_conv/dense.py:19, modules/pointnet2/dense.py:4).
 * Each entry point below is what that package's Python function would bind for one call; the
 * Python wrapper lives in torch_points3d_amd/torchpoints.py and INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *    (the library never allocates, frees or synchronises);
 *  - all tensors are dense, row-major, contiguous; float = IEEE fp32; indices are int64 (the
 *    reference feeds them to torch.gather, core/base_conv/dense.py:75-76);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); work is enqueued, not awaited;
 *  - return 0 on success, a negative TP3D_E_* code otherwise (no exception crosses the boundary);
 *    tp3d_strerror() gives text, tp3d_last_hip_error() the hipError_t of a failed launch;
 *  - squared distances are evaluated as (dx*dx + dy*dy) + dz*dz, fp32, no fused multiply-add, so that
 *    index outputs are bit-exact against the CPU oracle (oracle/tpk_ref_cpu.c).
 */
#ifndef TP3D_HIP_H
#define TP3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TP3D_OK 0
#define TP3D_E_BADARG (-1)   /* negative / inconsistent sizes, null pointer with non-empty work */
#define TP3D_E_LAUNCH (-2)   /* hipGetLastError() != hipSuccess after the launch */
#define TP3D_E_UNSORTED (-3) /* reserved: batch vector not sorted (checked by the host wrapper) */
#define TP3D_E_TOOBIG (-4)   /* size exceeds what the kernel's index arithmetic supports */

#define TP3D_ABI_VERSION 36

int tp3d_abi_version(void);
const char *tp3d_strerror(int code);
int tp3d_last_hip_error(void);

/*
 * furthest_point_sample(xyz, npoint)            [reference call: core/spatial_ops/sampling.py:100]
 *   xyz (B,N,3) f32 -> out_idx (B,npoint) int64.  sel[0]=0, then argmax of the running min squared
 *   distance, ties -> lowest index.  scratch: B*N floats (used only when N > TP3D_FPS_MAX_REG_POINTS).
 */
#define TP3D_FPS_MAX_REG_POINTS 32768
int tp3d_fps_f32(const float *xyz, int B, int N, int npoint, float *scratch, int64_t *out_idx, void *stream);

/*
 * ball_query(radius, nsample, x, y, mode="dense", sort)
 *                                               [reference call: core/spatial_ops/neighbour_finder.py:164,
 *                                                core/losses/dirichlet_loss.py:52]
 *   x (B,N,3), y (B,np,3) -> idx (B,np,nsample) int64, dist2 (B,np,nsample) f32.
 *   sort=0: hits (d2 < r*r, strict) in ascending index order, first nsample; remaining slots repeat the
 *           first hit (0 if none); dist2 = -1 in padded slots.
 *   sort=1: the nsample closest hits, closest first (ties by index), padded with the closest.
 *   workspace (optional, tp3d_ball_query_workspace_bytes(B, B*N, N) bytes): with it, clouds of >= 2048 points are
 *   searched through a uniform grid (27 cells per query) with bit-identical output; NULL = brute force.
 */
int tp3d_ball_query_dense_f32(const float *x, const float *y, int B, int N, int np, float radius, int nsample,
                              int sort, int64_t *idx, float *dist2, void *workspace, size_t workspace_bytes,
                              void *stream);
size_t tp3d_ball_query_workspace_bytes(int num_clouds, int64_t rows, int max_cloud_points);

/*
 * ball_query(..., mode="partial_dense", batch_x, batch_y)
 *                                               [reference call: core/spatial_ops/neighbour_finder.py:31-37]
 *   x (M,3) with ascending batch_x (M) int64, y (Nq,3) with batch_y (Nq) int64.
 *   idx (Nq,nsample) int64 = global rows of x, padded with -1; dist2 padded with -1.
 *   Optional grid acceleration: seg_x (num_clouds+1) int64 device array of cloud row offsets into x,
 *   max_cloud_points = the largest cloud, workspace = tp3d_ball_query_workspace_bytes(num_clouds, M,
 *   max_cloud_points) bytes; pass NULL / 0 for the brute-force scan of each query's cloud segment.
 *   reuse_grid != 0: the workspace still holds the grid the previous call on it built for the SAME x, seg_x,
 *   max_cloud_points and radius (e.g. the last block of a KPConv level and the strided block of the next level
 *   search the same support with the same radius, modules/KPConv/blocks.py:52): the build is skipped.
 */
int tp3d_ball_query_partial_dense_f32(const float *x, const float *y, const int64_t *batch_x,
                                      const int64_t *batch_y, int64_t M, int64_t Nq, float radius, int nsample,
                                      int sort, int64_t *idx, float *dist2, const int64_t *seg_x, int num_clouds,
                                      int max_cloud_points, void *workspace, size_t workspace_bytes, int reuse_grid,
                                      void *stream);

/*
 * three_nn(unknown, known) -> (dist, idx)       [reference call: core/base_conv/dense.py:136]
 *   unknown (B,n,3), known (B,m,3), m >= 3 -> dist (B,n,3) f32 Euclidean (sqrt), idx (B,n,3) int64;
 *   ascending distance, ties -> lowest index.
 */
int tp3d_three_nn_f32(const float *unknown, const float *known, int B, int n, int m, float *dist, int64_t *idx,
                      void *stream);

/*
 * three_interpolate(features, idx, weight)      [reference call: core/base_conv/dense.py:140]
 *   features (B,C,m), idx (B,n,3) int64, weight (B,n,3) -> out (B,C,n) = (w0*f0 + w1*f1) + w2*f2.
 *   bwd: grad_out (B,C,n) -> grad_features (B,C,m) (overwritten, not accumulated); atomic-free and bitwise
 *        reproducible; needs tp3d_scatter_workspace_bytes(B, 3*n, m, 1) bytes of device workspace.
 */
int tp3d_three_interpolate_fwd_f32(const float *features, const int64_t *idx, const float *weight, int B, int C,
                                   int m, int n, float *out, void *stream);
int tp3d_three_interpolate_bwd_f32(const float *grad_out, const int64_t *idx, const float *weight, int B, int C,
                                   int m, int n, float *grad_features, void *workspace, size_t workspace_bytes,
                                   void *stream);

/*
 * Device workspace (bytes) of the two scatter-add backward entry points: the inverse index of an
 * index table with L slots per cloud pointing into nbins destinations (+ the permuted weights).
 */
size_t tp3d_scatter_workspace_bytes(int B, int L, int nbins, int with_weights);

/*
 * grouping_operation(features, idx)             [reference call: modules/pointnet2/dense.py:38,45]
 *   features (B,C,N), idx (B,np,ns) int64 -> out (B,C,np,ns); bwd scatters grad_out into (B,C,N)
 *   (overwritten, not accumulated); atomic-free and bitwise reproducible; needs
 *   tp3d_scatter_workspace_bytes(B, np*ns, N, 0) bytes of device workspace.
 */
int tp3d_group_fwd_f32(const float *features, const int64_t *idx, int B, int C, int N, int np, int ns, float *out,
                       void *stream);
int tp3d_group_bwd_f32(const float *grad_out, const int64_t *idx, int B, int C, int N, int np, int ns,
                       float *grad_features, void *workspace, size_t workspace_bytes, void *stream);

/* =====================================================================================================
 * Grouped-MLP aggregation in channel-last ("rows") layout: the work PointNetMSGDown / DenseFPModule /
 * GlobalDenseBaseModule do around their 1x1-conv GEMMs (reference modules/pointnet2/dense.py:36-75,
 * core/base_conv/dense.py:102-184, core/common_modules/dense_modules.py:5-29).  Activations are (rows, C)
 * row-major fp32; the GEMMs themselves are plain library GEMMs issued by the host wrapper.
 * ===================================================================================================== */

/* out[(b,j,s), :] = [ (pos[b,idx[b,j,s]] - new_pos[b,j]) (/ radius if normalize), x_cl[b,idx[b,j,s], 0:C], 0.. ]
 * pos (B,N,3), new_pos (B,np,3), x_cl (B,N,C) or NULL when C == 0, idx (B,np,ns) -> out (B*np*ns, ld),
 * ld >= 3+C; columns past 3+C are written as zeros (row padding for 16-byte aligned GEMM operands). */
int tp3d_group_concat_fwd_f32(const float *pos, const float *new_pos, const float *x_cl, const int64_t *idx, int B,
                              int N, int np, int ns, int C, int ld, float radius, int normalize, float *out,
                              void *stream);

/* Scatter-add of row gradients back onto their source points, atomic-free (inverse index + gather-sum):
 *   grad_x_cl[b,k,0:C] = sum over slots l of cloud b with idx[b,l] == k (ascending l) of
 *                        weight[b,l] * grad_rows[(b, l/div), col0 : col0+C]          (weight NULL -> 1)
 * grad_rows (B, L/div, ld); idx (B, L) with values in [0, nbins); grad_x_cl (B, nbins, C).
 * workspace: tp3d_scatter_workspace_bytes(B, L, nbins, weight != NULL). */
int tp3d_rows_scatter_bwd_f32(const float *grad_rows, const int64_t *idx, const float *weight, int B, int L, int div,
                              int nbins, int ld, int col0, int C, float *grad_x_cl, void *workspace,
                              size_t workspace_bytes, void *stream);
/* The same in two halves: the inverted table depends on idx / weight only (geometry, not features), so it can be built
 * ahead of the backward pass -- tp3d_rows_scatter_invert fills `workspace` (same size query), tp3d_rows_scatter_apply_f32
 * consumes a table built for the same (idx, weight, B, L, div, nbins); with_weights = the table was built with weights.
 * Sums run in ascending slot order; a destination with more than 128 slots is summed in 16 contiguous pieces that are
 * then added in order (fixed association: reproducible, not the sequential sum bit for bit). */
int tp3d_rows_scatter_invert(const int64_t *idx, const float *weight, int B, int L, int div, int nbins, void *workspace,
                             size_t workspace_bytes, void *stream);
int tp3d_rows_scatter_apply_f32(const float *grad_rows, int B, int L, int div, int nbins, int ld, int col0, int C,
                                int with_weights, float *grad_x_cl, void *table, size_t table_bytes, void *stream);

/* BatchNorm statistics of Y (M, C): mean, invstd, scale = gamma*invstd and shift = beta; the normalised value is
 * always formed as (y - mean)*scale + shift (a folded shift beta - mean*scale would cancel against y*scale in fp32
 * when |mean| >> std).  Batch statistics are accumulated as shifted sums per row chunk and merged with Chan's formula.
 * training != 0: batch mean / biased variance (running stats updated in place with `momentum`, unbiased var;
 * *num_batches_tracked, if given, incremented on the device -- BatchNorm's counter without a launch of its own);
 * training == 0: running statistics.  workspace: tp3d_bn_workspace_floats(M, C) floats. */
size_t tp3d_bn_workspace_floats(int64_t M, int C);
int tp3d_bn_stats_f32(const float *Y, int64_t M, int C, float eps, float momentum, const float *gamma,
                      const float *beta, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                      int training, float *mean, float *invstd, float *scale, float *shift, float *workspace,
                      void *stream);

/* out = LeakyReLU_slope((Y - mean)*scale + shift) over (M, C);  the pooled form also takes the max over each group of
 * ns consecutive rows (first maximum wins) and records its row in argmax (G, C). */
int tp3d_bn_act_f32(const float *Y, const float *mean, const float *scale, const float *shift, float slope, int64_t M,
                    int C, float *out, void *stream);
int tp3d_bn_act_maxpool_f32(const float *Y, const float *mean, const float *scale, const float *shift, float slope,
                            int64_t G, int ns, int C, float *out, int *argmax, void *stream);

/* Backward of out = act(BN(Y)): dbeta, dgamma (C) and dY (M, C).  dA is (M, C), or -- with argmax != NULL --
 * the gradient (M/ns, C) of the pooled output.  workspace: tp3d_bn_workspace_floats(M, C) floats. */
int tp3d_bn_act_bwd_f32(const float *dA, const int *argmax, const float *Y, const float *scale, const float *shift,
                        const float *mean, const float *invstd, float slope, int64_t M, int ns, int C, int training,
                        float *dbeta, float *dgamma, float *dY, float *workspace, void *stream);
/* Only the reductions of that backward pass: dbeta, dgamma (C) and the two per-channel terms the fused GEMMs
 * (tp3d_gemm_rows_bnbwd_sp_f32, tp3d_gemm_tn_bn_narrow_f32) subtract while they form dY:  c1 = dbeta / M,  c2 = invstd * dgamma / M
 * (both zero with training == 0).  Same arguments and workspace as tp3d_bn_act_bwd_f32, no dY. */
int tp3d_bn_bwd_reduce_f32(const float *dA, const int *argmax, const float *Y, const float *scale, const float *shift,
                           const float *mean, const float *invstd, float slope, int64_t M, int ns, int C, int training,
                           float *dbeta, float *dgamma, float *c1, float *c2, float *workspace, int reverse,
                           void *stream);

/* out[(b,i), :] = [ (w0*f0 + w1*f1) + w2*f2 , skip_cl[b,i,0:C2], 0.. ],  f_t = feat_cl[b, idx[b,i,t], 0:C1]
 * feat_cl (B,m,C1), idx/weight (B,n,3), skip_cl (B,n,C2) or NULL -> out (B*n, ld), ld >= C1+C2 (zero padded). */
int tp3d_interp_concat_fwd_f32(const float *feat_cl, const int64_t *idx, const float *weight, const float *skip_cl,
                               int B, int m, int n, int C1, int C2, int ld, float *out, void *stream);

/* `reverse` (tp3d_gemm_tn_x3_act_red_f32, tp3d_gemm_rows_narrow_f32, tp3d_gemm_tn_bn_narrow_f32, tp3d_gemm_rows_bnact_sp_f32, tp3d_gemm_rows_bnact_x3_f32, tp3d_gemm_rows_bnbwd_sp_f32, tp3d_gemm_tn_x3_f32,
 * tp3d_gemm_tn_x3_act_f32, tp3d_bn_bwd_reduce_f32): 1 = walk the row blocks of the (M, .) operands last to first.  The result
 * is the same set of products / sums (the contractions over rows sum their blocks in the walked order: reproducible per
 * direction, the two directions differ by rounding).  A chain of kernels over 268 MB activation matrices alternates the
 * direction so that each kernel starts on the rows its predecessor touched last -- what the 256 MB memory-side cache
 * still holds (measured: 7.92 -> 7.76 ms per training step of the headline network). */
/* Tall-skinny GEMM of a shared-MLP layer, fp32 MFMA:  C[M,N] = A[M,K] * Bt[N,K]^T  (row-major, K % 4 == 0).
 * Forward: A = rows, Bt = W exactly as nn.Conv2d stores it (Cout x Cin).  With stat_partial != NULL the epilogue also
 * writes shifted partial column sums of C: stat_partial[chunk][4][N] = sum d, sum d^2 (d = value - shift), shift, rows;
 * chunk < tp3d_gemm_rows_stat_chunks(M, N); the buffer holds tp3d_gemm_rows_stat_floats(M, N) floats;
 * tp3d_bn_finalize_f32 turns them into the BatchNorm statistics -- no separate statistics pass over C.
 * Without statistics and with `workspace` (tp3d_gemm_rows_workspace_floats(M, N, K) floats, 0 = not needed) a long
 * contraction with few output tiles is split over K-ranges into partial slabs summed in fixed order. */
size_t tp3d_gemm_rows_stat_floats(int64_t M, int N);
int tp3d_gemm_rows_stat_chunks(int64_t M, int N);
size_t tp3d_gemm_rows_workspace_floats(int64_t M, int N, int K);
int tp3d_gemm_rows_f32(const float *A, const float *Bt, int64_t M, int N, int K, float *C, float *stat_partial,
                       float *workspace, void *stream);
/* The same contraction with the eval-mode BatchNorm + LeakyReLU of its OUTPUT applied to the accumulators:
 *   C = LeakyReLU((A Bt^T - mean[n]) * scale[n] + beta[n])      (mean / scale / beta: N floats, as tp3d_bn_stats_f32 leaves them)
 * -- one launch per Linear -> BatchNorm (running statistics) -> activation layer of an inference pass
 * (core/common_modules/base_modules.py: FastBatchNorm1d + activation after nn.Linear).  workspace as for
 * tp3d_gemm_rows_f32 without statistics (K-split slabs; the epilogue then runs in the slab sum, which needs N % 4 == 0). */
int tp3d_gemm_rows_epi_f32(const float *A, const float *Bt, int64_t M, int N, int K, const float *mean, const float *scale,
                           const float *beta, float slope, float *C, float *workspace, void *stream);
/* The same fused contraction on the split-role kernel (csrc/gemm_rows_sp.hip: four MFMA waves fed by four loader waves
 * that apply the prologue on their way into LDS, so it costs the MFMA waves nothing).  act_out != NULL additionally
 * receives the activated rows (M,K) the backward pass of the next layer contracts with (training); stat_partial as for
 * tp3d_gemm_rows_f32 but with tp3d_gemm_rows_sp_chunks(M, N, K, act_out != NULL) chunks of 4 * N floats.  Shapes: K % 4 == 0,
 * 4 <= K <= 512, N <= 64 or (N % 128 not in 1..64 and N <= 1024 in 1, 2, 4 or 8 column tiles), at least 512 output
 * tiles --
 * tp3d_gemm_rows_sp_chunks returns 0 for a shape that is not served (TP3D_E_BADARG from the launch).
 * Replaces the Conv2d -> BatchNorm2d -> LeakyReLU hand-over between two layers of MLP2D
 * (core/common_modules/dense_modules.py:25-29). */
int tp3d_gemm_rows_sp_chunks(int64_t M, int N, int K, int with_act_out);
int tp3d_gemm_rows_bnact_sp_f32(const float *Y, const float *mean, const float *scale, const float *beta, float slope,
                                const float *Bt, int64_t M, int N, int K, float *C, float *stat_partial, float *act_out,
                                int reverse, void *stream);
/* tp3d_gemm_rows_bnact_sp_f32 with the fp32 contraction carried by the bf16 matrix pipe (csrc/gemm_rows_x3.hip): every operand
 * value split exactly into three bf16 terms by the loader waves (x3_split.h), the product as the six term pairs of weight
 * >= 2^-15, each a v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Same arguments, outputs and statistics layout;
 * tp3d_gemm_rows_x3_chunks is its chunk count / shape rule (0: not served -- more than 64 output columns, K <= 512). */
int tp3d_gemm_rows_x3_chunks(int64_t M, int N, int K, int with_act_out);
int tp3d_gemm_rows_bnact_x3_f32(const float *Y, const float *mean, const float *scale, const float *beta, float slope,
                                const float *Bt, int64_t M, int N, int K, float *C, float *stat_partial, float *act_out,
                                int reverse, void *stream);

/* Input-gradient GEMM of a layer on the split-role kernel, with the layer's BatchNorm + activation BACKWARD formed by the
 * loader waves:  C[M,N] = dY[M,K] * Bt[N,K]^T,  dY = scale*((dA*act'(z) - c1) - (Y - mean)*c2),  z = (Y - mean)*scale + beta.
 * Y (M,K) pre-BatchNorm output of the layer, dA (M,K) gradient of its activated output -- or, with argmax != NULL, the
 * gradient (M/ns, K) of its max-pooled output and the winning rows (tp3d_bn_act_maxpool_f32; ns a power of two >= 64) --, c1 / c2 (K) from
 * tp3d_bn_bwd_reduce_f32, Bt (N,K) = W^T of the layer (N = its input width).  dY_out (M,K), if not NULL, receives dY for the weight-gradient contraction; ldc >= N is the row stride of C in
 * floats (the gradient of a column range of wider rows: the grouped rows' feature columns) and the pad_lo <= 32 columns
 * in front of / pad_hi <= 32 behind that range are written as zeros.  Shapes: as tp3d_gemm_rows_bnact_sp_f32 with K <= 256
 * (tp3d_gemm_rows_bnbwd_sp_serves).  Autograd of Conv2d -> BatchNorm2d -> LeakyReLU
 * (core/common_modules/dense_modules.py:25-29). */
int tp3d_gemm_rows_bnbwd_sp_serves(int64_t M, int N, int K);
int tp3d_gemm_rows_bnbwd_sp_f32(const float *Y, const float *dA, const float *mean, const float *scale, const float *beta,
                                const float *c1, const float *c2, float slope, const float *Bt, int64_t M, int N, int K,
                                float *C, int ldc, int pad_lo, int pad_hi, float *dY_out, const int *argmax, int ns,
                                int reverse, void *stream);
/* `partial` is consumed: with more than 64 chunks they are first folded, in place, into 32 slices. */
int tp3d_bn_finalize_f32(float *partial, int chunks, int64_t M, int C, float eps, float momentum, const float *gamma,
                         const float *beta, float *running_mean, float *running_var, int64_t *num_batches_tracked,
                         float *mean, float *invstd, float *scale, float *shift, void *stream);

/* Weight gradient of the FIRST layer of a shared MLP on grouped rows (K <= 16 input channels, K % 4 == 0) with dY formed
 * on the fly:  out[n,k] = sum_r dY[r,n] * A[r,k],  dY = BatchNorm + LeakyReLU backward of (Y, dA) as in
 * tp3d_gemm_rows_bnbwd_sp_f32 (c1_n / c2_n from tp3d_bn_bwd_reduce_f32).  Y, dA (M,N), A (M,K) -> out (N,K).
 * One streaming pass over Y and dA (2*M*N*K flops against 8*M*N bytes: HBM-bound); fixed summation order.
 * workspace: tp3d_gemm_tn_bn_narrow_workspace_floats(M, N, K) floats.
 * Autograd of Conv2d -> BatchNorm2d -> LeakyReLU (core/common_modules/dense_modules.py:25-29). */
int tp3d_gemm_tn_bn_narrow_serves(int64_t M, int N, int K);
/* The forward contraction of such a layer (same shape rule):  Y (M,N) = A (M,K) W (N,K)^T, k ascending per output, one
 * streaming pass (4 M (N + K) bytes: write-dominated; the MFMA tile kernel pads K to its 32-wide step and runs at 3 TB/s).
 * stat_partial != NULL: tp3d_gemm_rows_narrow_chunks(M) chunks of [4][N] floats (sum d, sum d^2, shift, rows) in the layout
 * tp3d_bn_finalize_f32 folds.  Conv2d 1x1 bias=False of core/common_modules/dense_modules.py:25-29. */
int tp3d_gemm_rows_narrow_chunks(int64_t M);
int tp3d_gemm_rows_narrow_f32(const float *A, const float *W, int64_t M, int N, int K, float *Y, float *stat_partial,
                              int reverse, void *stream);
size_t tp3d_gemm_tn_bn_narrow_workspace_floats(int64_t M, int N, int K);
int tp3d_gemm_tn_bn_narrow_f32(const float *Y, const float *dA, const float *mean_n, const float *scale_n,
                               const float *beta_n, const float *c1_n, const float *c2_n, float slope_n, const float *A,
                               int64_t M, int N, int K, float *out, float *workspace, int reverse, void *stream);

/* Weight gradient of a 1x1 conv / shared-MLP layer:  out[n,k] = sum_r dY[r,n] * A[r,k]
 * dY (M,N), A (M,K) row-major -> out (N,K); rows split over the grid, fp32 MFMA, fixed-order reduction of the
 * splits (reproducible).  workspace: tp3d_gemm_tn_workspace_floats(M, N, K) floats. */
size_t tp3d_gemm_tn_workspace_floats(int64_t M, int N, int K);
int tp3d_gemm_tn_f32(const float *dY, const float *A, int64_t M, int N, int K, float *out, float *workspace,
                     void *stream);

/* The same contraction with the fp32 products carried by the bf16 matrix pipe (csrc/gemm_tn_x3.hip): every operand value is
 * split exactly into three bf16 terms (3 x 8 significand bits) by the loader waves of a split-role kernel, and the product
 * is the sum of the term pairs -- terms = 9: all nine, every product exact as in the fp32 MFMA; terms = 6: the six of
 * weight >= 2^-16 -- each a v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The contraction then runs at the rate HBM
 * delivers the rows instead of at the fp32 MFMA rate.  Serves M >= 131072, N >= 64, K >= 64 (not both 64), N % 4 == 0, K % 4 == 0, M * max(N, K) < 2^30, and an output that
 * fills at least 80 % of at most two tiles (128- or 64-wide, or 128 + a 32-column strip)
 * (tp3d_gemm_tn_x3_serves); other shapes: tp3d_gemm_tn_f32.  workspace: tp3d_gemm_tn_x3_workspace_floats floats.
 * Reference: the weight gradient of Conv2d 1x1 in MLP2D (core/common_modules/dense_modules.py:5-12), autograd's
 * grad_output^T @ input. */
int tp3d_gemm_tn_x3_serves(int64_t M, int N, int K);
size_t tp3d_gemm_tn_x3_workspace_floats(int64_t M, int N, int K);
int tp3d_gemm_tn_x3_f32(const float *dY, const float *A, int64_t M, int N, int K, int terms, float *out, float *workspace,
                        int reverse, void *stream);
/* ... with the A operand formed on the fly: dW = dY^T * LeakyReLU((Yp - mean_k) * scale_k + beta_k), Yp (M, K) the previous
 * layer's pre-BatchNorm output, mean_k / scale_k / beta_k its statistics rows (K floats each).  The loader waves evaluate
 * the forward kernels' expression in their order, so the result is bit for bit that of the plain entry point on the
 * activated rows -- which the forward pass then need not write. */
int tp3d_gemm_tn_x3_act_f32(const float *dY, const float *Yp, const float *mean_k, const float *scale_k, const float *beta_k,
                            float slope_k, int64_t M, int N, int K, int terms, float *out, float *workspace, int reverse, void *stream);
/* ... and, from one more operand stream, the BatchNorm-backward REDUCTIONS of the layer Yp belongs to: dA_k (M,K) is the
 * gradient of that layer's activated output, invstd_k (K) its statistics row; red_out (4,K) = dbeta, dgamma,
 * c1 = dbeta / M, c2 = invstd * dgamma / M (zero with training == 0) -- what tp3d_bn_bwd_reduce_f32(dA_k, Yp) returns,
 * without that pass (its 8 M K bytes become 4 M K here: Yp is already streaming through this kernel's loader waves).
 * tp3d_gemm_tn_x3_red_chunks(M,N,K): 0 = shape not served (needs one tile column, K <= 128, and a 128 x 128, 128 x 64 or 64 x 64 tile), else the chunks of
 * [2][K] floats red_workspace must hold.  terms == 6. */
int tp3d_gemm_tn_x3_red_chunks(int64_t M, int N, int K);
int tp3d_gemm_tn_x3_act_red_f32(const float *dY, const float *Yp, const float *mean_k, const float *scale_k,
                                const float *beta_k, const float *invstd_k, float slope_k, const float *dA_k, int training,
                                int64_t M, int N, int K, int terms, float *out, float *workspace, float *red_out,
                                float *red_workspace, int reverse, void *stream);

/* KPConv rigid convolution, stage 1 (reference modules/KPConv/convolution_ops.py:19-98):
 *   weighted[q, k, :] = sum_n h(|(support[nbr[q,n]] - query[q]) - k_points[k]|) * features[nbr[q,n], :]
 * query (Nq,3), support (M,3), neighbors (Nq,Mn) int64 with -1 (or >= M) = shadow neighbour, features (M,Cin),
 * k_points (KP,3), KP <= 16 -> weighted (Nq, KP, Cin).  influence: 0 constant, 1 linear (max(1 - d/extent, 0)),
 * 2 gaussian (sigma = 0.3*extent); closest != 0 keeps only the nearest kernel point per neighbour.
 * Stage 2 (:101-105) is one GEMM  (Nq, KP*Cin) x (KP*Cin, Cout)  issued by the host wrapper. */
int tp3d_kpconv_weighted_f32(const float *query, const float *support, const int64_t *neighbors,
                             const float *features, const float *k_points, int64_t Nq, int64_t M, int Mn, int Cin,
                             int KP, float extent, int influence, int closest, float *weighted, void *stream);

/* Backward of stage 1 wrt the input features (autograd through convolution_ops.py:92-98):
 *   d_features[m, :] = sum over (q,n) with neighbors[q,n] == m of sum_k h(...) * d_weighted[q, k, :]
 * d_weighted (Nq, KP, Cin) = d_out @ W2^T (host GEMM) -> d_features (M, Cin), overwritten; atomic-free and
 * reproducible: per-slot gradient rows (Nq*Mn, Cin) are formed per query and summed per support point in slot order
 * through the inverted neighbour table.
 *   inverse: tp3d_kpconv_bwd_workspace_bytes(M, Nq*Mn) bytes holding the inverted table; it is (re)built by the call
 *            unless inverse_ready != 0, i.e. the caller kept the buffer of an earlier call on the SAME neighbours / M
 *            (tp3d_nbr_maxpool_bwd_f32 shares it: a strided block inverts its table once for both; with precomputed
 *            neighbour tables it is built once for the whole training run);
 *   workspace: tp3d_kpconv_grad_workspace_bytes(M, Nq*Mn, Cin) bytes for the per-slot rows. */
size_t tp3d_kpconv_bwd_workspace_bytes(int64_t M, int64_t slots);
size_t tp3d_kpconv_grad_workspace_bytes(int64_t M, int64_t slots, int Cin);
int tp3d_kpconv_bwd_features_f32(const float *query, const float *support, const int64_t *neighbors,
                                 const float *k_points, const float *d_weighted, int64_t Nq, int64_t M, int Mn,
                                 int Cin, int KP, float extent, int influence, int closest, float *d_features,
                                 void *inverse, size_t inverse_bytes, int inverse_ready, void *workspace,
                                 size_t workspace_bytes, void *stream);

/* Geometric relation rows of Relation-Shape convolution (reference modules/RSConv/dense.py:86-101):
 *   out[(b,j,s), 0:10] = [ |d|, new_pos[b,j] (3), p (3), d (3) ],  p = pos[b, idx[b,j,s]],  d = p - new_pos[b,j];
 *   columns 10 .. ld-1 are zero.  pos (B,N,3), new_pos (B,np,3), idx (B,np,ns) -> out (B*np*ns, ld), ld >= 10. */
int tp3d_relation_rows_f32(const float *pos, const float *new_pos, const int64_t *idx, int B, int N, int np, int ns,
                           int ld, float *out, void *stream);

/* inverse-distance weights of DenseFPModule (core/base_conv/dense.py:137-139): dist (rows,3) -> weight (rows,3) */
int tp3d_idw_weights_f32(const float *dist, int64_t rows, float *weight, void *stream);

/* GridSampling3D (reference core/data_transform/grid_transform.py:84-141; sampler of the strided KPConv blocks,
 * modules/KPConv/blocks.py:60-61,79).  pos (N,3) f32, batch (N) int64 or NULL, voxel edge `size`:
 *   coords = round_half_even(pos / size);  points sharing (batch, coords) form a cluster;  clusters are numbered in
 *   ascending (batch, z, y, x) order (torch_cluster grid_cluster key + torch.unique(sorted), :117-122).
 * Three steps, because the reference's own pipeline has two device->host waits (extent of the key, number of
 * clusters) and this library never synchronises:
 *   1. tp3d_voxel_bounds_f32   -> bounds[8] int32 on the DEVICE: min xyz, max xyz of coords, max batch, bad-input flag;
 *                                 the host copies them back and passes them to step 2 as a HOST array;
 *   2. tp3d_voxel_cluster_f32  -> cluster (N) id of each point; order (N) point indices sorted by (cluster, index);
 *                                 cluster_start (N+1; [0..K] valid) slot range of each cluster in `order`;
 *                                 last (N; [0..K) valid) highest point index of each cluster = the reference's
 *                                 unique_pos_indices (consecutive_cluster's scatter_, last write wins);
 *                                 meta (1 + clouds) int64 on the device, clouds = bounds[6] + 1 (1 without batch):
 *                                 meta[0] = K, meta[1 + b] = clusters of clouds 0..b together (0 for a cloud without
 *                                 points): the row offsets of the sampled batch vector.  workspace:
 *                                 tp3d_voxel_workspace_bytes(N);
 *   3. tp3d_cluster_mean_f32   -> out (K,C) = scatter_mean(x (N,C)) summed in ascending point order (:78), and
 *      tp3d_cluster_majority_i64 -> out (K) = majority label, ties -> lowest (:72-76); num_classes = max-min+1.
 */
int tp3d_voxel_bounds_f32(const float *pos, const int64_t *batch, int64_t N, float size, int32_t *bounds, void *stream);
size_t tp3d_voxel_workspace_bytes(int64_t N);
int tp3d_voxel_cluster_f32(const float *pos, const int64_t *batch, int64_t N, float size, const int32_t *bounds_host,
                           int64_t *cluster, int64_t *order, int64_t *cluster_start, int64_t *last, int64_t *meta,
                           void *workspace, size_t workspace_bytes, void *stream);
int tp3d_cluster_mean_f32(const float *x, const int64_t *order, const int64_t *cluster_start, int64_t K, int C,
                          float *out, void *stream);
int tp3d_cluster_majority_i64(const int64_t *labels, const int64_t *order, const int64_t *cluster_start, int64_t K,
                              int64_t min_label, int64_t num_classes, int64_t *out, void *stream);

/* Exact k nearest neighbours (reference call sites: core/spatial_ops/interpolate.py:27,69 -- KNNInterpolate /
 * FPModule_PD, k = 1 in the KPConv decoders; core/spatial_ops/neighbour_finder.py:42-47 -- KNNNeighbourFinder, k = 16
 * in RandLA-Net; both go through torch_cluster `knn`).
 *   partial_dense: x (M,3) support with sorted batch ids, seg_x (num_clouds+1) row offsets of the clouds in x,
 *                  y (Nq,3) queries with batch_y (Nq) -> idx (Nq,k) global rows of x, dist2 (Nq,k) squared distances;
 *   dense:         x (B,N,3), y (B,np,3) -> idx (B,np,k) cloud-local, dist2 (B,np,k).
 *   Closest first, ties by lower index; slots a cloud of fewer than k points cannot fill hold -1 / -1.0.
 *   cell: preferred grid cell edge (e.g. the sampling grid size); <= 0 lets the library choose.
 *   workspace: tp3d_knn_workspace_bytes(num_clouds, rows of x, largest cloud) bytes. */
size_t tp3d_knn_workspace_bytes(int num_clouds, int64_t rows, int max_cloud_points);
int tp3d_knn_partial_dense_f32(const float *x, const float *y, const int64_t *batch_y, const int64_t *seg_x,
                               int num_clouds, int max_cloud_points, int64_t M, int64_t Nq, int k, float cell,
                               int64_t *idx, float *dist2, void *workspace, size_t workspace_bytes, void *stream);
int tp3d_knn_dense_f32(const float *x, const float *y, int B, int N, int np, int k, float cell, int64_t *idx,
                       float *dist2, void *workspace, size_t workspace_bytes, void *stream);

/* knn_interpolate + skip concatenation of FPModule_PD (core/base_conv/partial_dense.py:136-140):
 *   w_j = 1 / max(dist2[i,j], 1e-16);  out[i, 0:C] = (sum_j x[idx[i,j], :] * w_j) / (sum_j w_j)   (slot order, -1 skipped)
 *   out[i, C:C+C2] = skip[i, :];  columns up to ld are zero.   x (M,C), idx/dist2 (Nq,k), skip (Nq,C2) or NULL.
 *   wnorm (Nq,k) or NULL receives w_j / sum_j w_j (0 in -1 slots): the weights of the backward scatter
 *   (tp3d_rows_scatter_bwd_f32 with B = 1). */
int tp3d_knn_interpolate_fwd_f32(const float *x, const int64_t *idx, const float *dist2, const float *skip, int64_t Nq,
                                 int k, int C, int C2, int ld, float *out, float *wnorm, void *stream);

/* RandLA-Net local feature aggregation over a fixed-k neighbour table (modules/RandLANet/modules.py:9-54; the reference
 * runs it as a torch_geometric MessagePassing over the edge list of torch_cluster's knn).  Edge e = q*k + n.
 *
 * relative position encoding (modules.py:36-41): out (Nq*k, 12) rows
 *   [q_pos[q] (3), s_pos[j] (3), q_pos[q] - s_pos[j] (3), |q_pos[q] - s_pos[j]| (1), 0, 0],  j = nbr[e];
 *   a missing neighbour (j < 0 or j >= M) gives a zero row. */
int tp3d_randla_relpos_f32(const float *q_pos, const float *s_pos, const int64_t *nbr, int64_t Nq, int k, int64_t M,
                           float *out, void *stream);

/* attentive pooling (modules.py:46-52 with aggr="add"):  out[q, c] = sum_n softmax_c(g[e, :])[c] * f[e, c]
 *   g (Nq*k, ldg) attention scores, f (Nq*k, ldf) edge features (C <= ldg, ldf; C, ldf <= 256), out (Nq, C).
 *   nbr (Nq*k) or NULL: edges with nbr[e] < 0 are left out of the sum.
 * backward: dg (Nq*k, ldg) columns [0, C) and df (Nq*k, ldf) all columns (padding zeroed) are overwritten:
 *   df = s * dout[q];  dg = s * (f * dout[q] - sum_c s * f * dout[q]),  s = softmax_c(g[e, :]). */
int tp3d_attn_pool_fwd_f32(const float *g, const float *f, const int64_t *nbr, int64_t Nq, int k, int C, int ldg, int ldf,
                           float *out, void *stream);
int tp3d_attn_pool_bwd_f32(const float *g, const float *f, const float *dout, const int64_t *nbr, int64_t Nq, int k, int C,
                           int ldg, int ldf, float *dg, float *df, void *stream);

/* Skinny row GEMM of the edge-wise MLPs (modules/RandLANet/modules.py:20-22; nn.Linear inside MLP,
 * core/common_modules/base_modules.py:29-43):  Y (M, N) = A (M, K; row stride lda >= K) * W (N, K)^T,  N, K <= 32.
 * One row per lane, k ascending per output.  Y is dense (row stride N). */
int tp3d_gemm_skinny_f32(const float *A, const float *W, int64_t M, int N, int K, int lda, float *Y, void *stream);
/* The same with BatchNorm (given statistics rows mean / scale / beta of N: eval mode) and LeakyReLU applied to every
 * output before it is stored: one pass over the rows instead of three (Linear, affine, activation). */
int tp3d_gemm_skinny_bnact_f32(const float *A, const float *W, int64_t M, int N, int K, int lda, const float *mean,
                               const float *scale, const float *beta, float slope, float *out, void *stream);

/* Strided shortcut of the KPConv ResnetBBlock (modules/KPConv/blocks.py:206-210):
 *   out[q, c] = max over n of x[neighbors[q,n], c], a shadow neighbour (-1 or >= M) contributing 0.0
 * x (M,C), neighbors (Nq,Mn) -> out (Nq,C); argmax (Nq,C) int32 or NULL = winning slot n (first maximum).
 * Backward: d_x (M,C) overwritten, atomic-free (inverse neighbour table); inverse / inverse_ready as for
 * tp3d_kpconv_bwd_features_f32. */
int tp3d_nbr_maxpool_fwd_f32(const float *x, const int64_t *neighbors, int64_t Nq, int64_t M, int Mn, int C, float *out,
                             int32_t *argmax, void *stream);
int tp3d_nbr_maxpool_bwd_f32(const float *grad_out, const int32_t *argmax, const int64_t *neighbors, int64_t Nq,
                             int64_t M, int Mn, int C, float *d_x, void *inverse, size_t inverse_bytes,
                             int inverse_ready, void *stream);

/* =====================================================================================================
 * Launch plans (host arithmetic only, no device work): what an entry point WILL do for given sizes -- how it
 * splits the rows, how many partial rows it writes, how it carves its workspace.  tests/test_plans_cpu.py sweeps
 * them against the workspace-size queries above, so that "the extent a kernel writes <= the size the caller was
 * told to allocate" is checked for every shape without a GPU.  All return 0 or TP3D_E_BADARG.
 * ===================================================================================================== */
/* plan[8]: splits, rows per split, tile rows (N side), tile columns (K side), tiles, rows staged per step,
 * workspace floats written, first row of the last split */
int tp3d_gemm_tn_plan(int64_t M, int N, int K, int64_t *plan);
/* plan[0..7] = splits, rows per split, tile rows (n), tile columns (k, strip included), tiles, rows staged per step,
 * workspace floats, dynamic LDS bytes of tp3d_gemm_tn_x3_f32 */
int tp3d_gemm_tn_x3_plan(int64_t M, int N, int K, int64_t *plan);
/* plan[9]: column tiles, row blocks, work items, workgroups, statistics chunks written, 1 = one chunk set per
 * workgroup, wave rows per tile, K-ranges of a split launch (K = 0: not asked), tile columns */
int tp3d_gemm_rows_plan(int64_t M, int N, int K, int64_t *plan);
/* plan[3]: rows per chunk, chunks, workspace floats written by tp3d_bn_stats_f32 (pooled_ns = 0) or
 * tp3d_bn_act_bwd_f32 (pooled_ns = its ns; 1 for the dense form) */
int tp3d_bn_plan(int64_t M, int C, int pooled_ns, int64_t *plan);
/* plan[9]: byte offsets of start, order, scratch, wsorted (-1 = none), merge_tmp; workspace bytes; 1 = table
 * inverted flat over the batch; ints of scratch that path needs; byte offset of the hub list (1 + B*nbins ints) */
int tp3d_scatter_plan(int B, int L, int nbins, int with_weights, int64_t *plan);

#ifdef __cplusplus
}
#endif
#endif /* TP3D_HIP_H */
