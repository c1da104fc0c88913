"""Host-side form of the multi-scale precompute: the chain of grid samplings, radius searches and interpolation tables
of a partial-dense network, computed on CPU tensors -- what the reference's `MultiScaleTransform` does inside DataLoader
worker processes (core/data_transform/transforms.py:579-654, datasets/base_dataset.py:251-263), so that the training
process receives ready tables and its forward pass contains no sampling, no search and no host read.

Built from the SAME strategy objects the device form takes (`model.get_spatial_ops()`): only their parameters are read
(voxel edge, radius, max_num_neighbors, k); the work is done by
  * voxel clustering on CPU tensors (torch.unique over the (batch, z, y, x) key -- reference grid_transform.py:84-141
    with torch_cluster's key order -- and sequential index_add_ means, the order the device kernels sum in),
  * `torch_points_kernels.points_cpu` (libtp3d_cpu.so: uniform grid, no GPU runtime, fork-safe) for the radius
    searches and the k-NN interpolation tables.
The result has the layout of `multiscale.MultiScaleTransform` (device form) and is bit-identical to it: the tables are
integers, positions are means summed in the same order (tests/test_gpu_kpconv_unet.py).  `.to(device)` moves it.
"""
import torch

from torch_points_kernels import points_cpu

from .kpconv_blocks import PDData


def grid_sampling_cpu(pos, batch, size):
    """GridSampling3D(mode="mean") on CPU tensors: (pos (K,3), batch (K,)) of the voxel representatives, voxels in
    ascending (batch, z, y, x) order, each position the mean of its members summed in ascending point order."""
    n = pos.shape[0]
    if n == 0:
        return pos.clone(), batch.clone()
    size_t = torch.full((), float(size), dtype=torch.float32)
    coords = torch.round(pos.float() / size_t).to(torch.int64)  # round half to even, true fp32 division
    lo = coords.min(0)[0]
    ext = coords.max(0)[0] - lo + 1
    c = coords - lo
    key = ((batch.to(torch.int64) * ext[2] + c[:, 2]) * ext[1] + c[:, 1]) * ext[0] + c[:, 0]
    uniq, inverse = torch.unique(key, sorted=True, return_inverse=True)
    k = uniq.numel()
    sums = torch.zeros((k, 3), dtype=torch.float32).index_add_(0, inverse, pos.float())  # sequential: ascending index
    counts = torch.zeros(k, dtype=torch.float32).index_add_(0, inverse, torch.ones(n))
    new_batch = torch.zeros(k, dtype=batch.dtype).scatter_(0, inverse, batch)  # members of a voxel share the cloud id
    return sums / counts.unsqueeze(1), new_batch


def _segments(batch):
    """row offsets of the clouds of a sorted batch vector"""
    nb = int(batch.max()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=nb)
    seg = torch.zeros(nb + 1, dtype=torch.int64)
    seg[1:] = torch.cumsum(counts, 0)
    return seg


def radius_search_cpu(support, query, batch_s, batch_q, radius, max_num):
    """partial-dense radius search (reference core/spatial_ops/neighbour_finder.py:25-39): (Nq, max_num) int64 global
    rows of `support`, ascending index, first max_num, -1 padded -- per cloud through points_cpu.ball_query"""
    out = torch.full((query.shape[0], max_num), -1, dtype=torch.int64)
    ss, sq = _segments(batch_s), _segments(batch_q)
    for b in range(min(ss.numel(), sq.numel()) - 1):
        s0, s1, q0, q1 = int(ss[b]), int(ss[b + 1]), int(sq[b]), int(sq[b + 1])
        if s1 == s0 or q1 == q0:
            continue
        ind, _ = points_cpu.ball_query(support[s0:s1], query[q0:q1], radius=radius, max_num=max_num, mode=0)
        out[q0:q1] = torch.where(ind >= 0, ind + s0, ind)
    return out


def knn_cpu(support, query, batch_s, batch_q, k):
    """exact k-NN of every query among the support points of its cloud: (idx global rows (Nq,k), squared dist (Nq,k))"""
    idx = torch.full((query.shape[0], k), -1, dtype=torch.int64)
    d2 = torch.full((query.shape[0], k), -1.0, dtype=torch.float32)
    ss, sq = _segments(batch_s), _segments(batch_q)
    for b in range(min(ss.numel(), sq.numel()) - 1):
        s0, s1, q0, q1 = int(ss[b]), int(ss[b + 1]), int(sq[b]), int(sq[b + 1])
        if s1 == s0 or q1 == q0:
            continue
        i, d = points_cpu.dense_knn(support[s0:s1].unsqueeze(0), query[q0:q1].unsqueeze(0), k)
        idx[q0:q1] = torch.where(i[0] >= 0, i[0] + s0, i[0])
        d2[q0:q1] = d[0]
    return idx, d2


class MultiScaleTransformCPU(object):
    def __init__(self, strategies):
        """strategies: {"sampler": [...], "neighbour_finder": [...], "upsample_op": [...]} as the model lists them
        (`get_spatial_ops()`); only `_grid_size`, `_radius`, `_max_num_neighbors` and `k` are read."""
        self.levels = []
        for sampler, finder in zip(strategies["sampler"], strategies["neighbour_finder"]):
            self.levels.append((None if not sampler else float(sampler._grid_size), float(finder._radius),
                                int(finder._max_num_neighbors)))
        self.up_k = [int(u.k) for u in strategies["upsample_op"]]

    def __call__(self, data):
        """data: CPU pos (N,3) [, batch (N,) sorted] -> PDData(multiscale=[...], upsample=[...]) + data's attributes"""
        pos = data.pos.detach().float().contiguous()
        if pos.device.type != "cpu":
            raise RuntimeError("MultiScaleTransformCPU works on CPU tensors (use multiscale.MultiScaleTransform on the device)")
        batch = getattr(data, "batch", None)
        if batch is None:
            batch = torch.zeros(pos.shape[0], dtype=torch.long)
        precomputed = [PDData(pos=pos, batch=batch)]
        upsample, up_index = [], 0
        for grid, radius, max_num in self.levels:
            support = precomputed[-1]
            if grid is not None:
                qpos, qbatch = grid_sampling_cpu(support.pos, support.batch, grid)
                query = PDData(pos=qpos, batch=qbatch)
                if self.up_k:
                    if up_index >= len(self.up_k):
                        raise ValueError("You are missing some upsample blocks in your network")
                    k = self.up_k[up_index]
                    up_index += 1
                    # interpolation from the sampled (query) cloud back to the support cloud
                    idx, d2 = knn_cpu(query.pos, support.pos, query.batch, support.batch, k)
                    n_sup = support.pos.shape[0]
                    y_idx = torch.arange(n_sup).repeat_interleave(k)
                    x_idx = idx.reshape(-1)
                    keep = x_idx >= 0
                    weights = (1.0 / torch.clamp(d2.reshape(-1, 1), min=1e-16))[keep]
                    y_idx, x_idx = y_idx[keep], x_idx[keep]
                    norm = torch.zeros((n_sup, 1)).index_add_(0, y_idx, weights)
                    upsample.append(PDData(num_nodes=n_sup, x_idx=x_idx, y_idx=y_idx, weights=weights, normalisation=norm,
                                           knn_idx=idx, knn_d2=d2))
            else:
                query = PDData(pos=support.pos, batch=support.batch)
            query.idx_neighboors = radius_search_cpu(support.pos, query.pos, support.batch, query.batch, radius, max_num)
            precomputed.append(query)
        out = data.shallow_copy() if hasattr(data, "shallow_copy") else PDData(**vars(data))
        out.multiscale = precomputed[1:]
        upsample.reverse()  # innermost decoder stage first
        out.upsample = upsample
        return out

    def __repr__(self):
        return "{}(levels={}, up_k={})".format(self.__class__.__name__, self.levels, self.up_k)


def to_device(ms_data, device):
    """move a precomputed sample (and the tables inside it) to the training device"""
    def move(bag):
        out = PDData()
        for k, v in vars(bag).items():
            if torch.is_tensor(v):
                v = v.to(device)
            elif isinstance(v, list):
                v = [move(e) if isinstance(e, PDData) else e for e in v]
            setattr(out, k, v)
        return out
    return move(ms_data)
