"""CPU ORACLE for GridSampling3D -- test infrastructure, NOT product code (only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import anything under oracle/).

numpy restatement of the pipeline the reference composes at torch_points3d/core/data_transform/grid_transform.py:113-128
from third-party packages that are absent from /root/reference and from this image:

  * torch_cluster 1.5.9 `grid_cluster` (csrc/cpu/grid_cpu.cpp: pos - start, true_divide(size), toType(long),
    multiplied by the running product of the per-dimension voxel counts and summed), reached through
    torch_geometric 1.7.2 `voxel_grid` (nn/pool/voxel_grid.py: batch appended to pos as a 4th coordinate of size 1);
  * torch_geometric 1.7.2 `consecutive_cluster` (nn/pool/consecutive.py: torch.unique(sorted, return_inverse), then
    perm = empty(K).scatter_(0, inv, arange(N)) -- sequential on the CPU, so the LAST index of each cluster stays);
  * torch_scatter 2.0.8 `scatter_mean` (sum in row order, divide by the clamped count) and `scatter_add`.

PARITY UNPINNED against those packages (they cannot be imported or fetched here); what pins this file is the
reference's own test properties for the transform (test/test_grid_sampling.py:29-67): majority label of a single voxel,
idempotence of a second sampling at the same size, `coords` dtype/shape.
"""
import numpy as np


def voxel_coords(pos, size):
    """torch.round(pos / size): fp32 true division, round half to even"""
    return np.rint(np.asarray(pos, np.float32) / np.float32(size)).astype(np.float32)


def grid_cluster_key(coords, batch=None):
    """voxel_grid(coords, batch, 1) -> int64 key per point (x fastest, batch slowest)"""
    c = np.asarray(coords, np.float32)
    if batch is not None:
        c = np.concatenate([c, np.asarray(batch).astype(np.float32)[:, None]], axis=1)
    start, end = c.min(axis=0), c.max(axis=0)
    num_voxels = (end - start).astype(np.int64) + 1  # size 1 in every dimension
    strides = np.concatenate([[1], np.cumprod(num_voxels)])[: c.shape[1]]
    return ((c - start[None, :]).astype(np.int64) * strides[None, :]).sum(axis=1)


def consecutive_cluster(key):
    uniq, inv = np.unique(key, return_inverse=True)
    perm = np.empty(uniq.shape[0], np.int64)
    perm[inv] = np.arange(inv.shape[0])  # numpy assigns in order: the last index of each cluster wins
    return inv.astype(np.int64), perm


def scatter_mean(x, cluster, K):
    x = np.asarray(x, np.float32)
    out = np.zeros((K,) + x.shape[1:], np.float32)
    np.add.at(out, cluster, x)  # unbuffered, row order: fp32 sums accumulate exactly like a sequential scatter_add
    cnt = np.maximum(np.bincount(cluster, minlength=K), 1).astype(np.float32)
    return out / cnt.reshape((-1,) + (1,) * (x.ndim - 1))


def majority_label(labels, cluster, K):
    labels = np.asarray(labels, np.int64)
    lo = labels.min()
    onehot = np.zeros((K, int(labels.max() - lo) + 1), np.int64)
    np.add.at(onehot, (cluster, labels - lo), 1)
    return onehot.argmax(axis=1) + lo  # first maximum = lowest label


def grid_sampling_mean(pos, size, batch=None, x=None, y=None):
    """GridSampling3D(size, mode='mean') on arrays -> dict(pos, batch, x, y, cluster, unique_pos_indices, coords)"""
    coords = voxel_coords(pos, size)
    cluster, perm = consecutive_cluster(grid_cluster_key(coords, batch))
    K = perm.shape[0]
    out = {"cluster": cluster, "unique_pos_indices": perm, "pos": scatter_mean(pos, cluster, K),
           "coords": coords[perm].astype(np.int32)}
    if batch is not None:
        out["batch"] = np.asarray(batch)[perm]
    if x is not None:
        out["x"] = scatter_mean(x, cluster, K)
    if y is not None:
        out["y"] = majority_label(y, cluster, K)
    return out
