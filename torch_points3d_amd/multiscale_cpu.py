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
from .multiscale import LevelChain, attach


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


class HostGridSampler(object):
    """sampler strategy on CPU tensors: `sampler(data) -> data` with pos / batch of the voxel representatives"""

    def __init__(self, size):
        self._grid_size = float(size)

    def __call__(self, data):
        pos, batch = grid_sampling_cpu(data.pos, data.batch, self._grid_size)
        return PDData(pos=pos, batch=batch)


class HostRadiusFinder(object):
    """neighbour_finder strategy on CPU tensors (the call shape of core/spatial_ops/neighbour_finder.py:25-39)"""

    def __init__(self, radius, max_num_neighbors):
        self._radius, self._max_num_neighbors = float(radius), int(max_num_neighbors)

    def __call__(self, x, y, batch_x=None, batch_y=None):
        if batch_x is None:
            batch_x = torch.zeros(x.shape[0], dtype=torch.long)
        if batch_y is None:
            batch_y = torch.zeros(y.shape[0], dtype=torch.long)
        return radius_search_cpu(x, y, batch_x, batch_y, self._radius, self._max_num_neighbors)


class HostKnnTable(object):
    """upsample_op strategy on CPU tensors: `.precompute(query, support)` as core/spatial_ops/interpolate.py:11-32 --
    the inverse-squared-distance table that carries features from the sampled (query) cloud back to `support`"""

    def __init__(self, k):
        self.k = int(k)

    def precompute(self, query, support):
        k = self.k
        idx, d2 = knn_cpu(query.pos, support.pos, query.batch, support.batch, k)
        n_sup = support.pos.shape[0]
        y_idx = torch.arange(n_sup).repeat_interleave(k)
        x_idx = idx.reshape(-1)
        keep = x_idx >= 0
        weights = (1.0 / torch.clamp(d2.reshape(-1, 1), min=1e-16))[keep]
        y_idx, x_idx = y_idx[keep], x_idx[keep]
        norm = torch.zeros((n_sup, 1)).index_add_(0, y_idx, weights)
        return PDData(num_nodes=n_sup, x_idx=x_idx, y_idx=y_idx, weights=weights, normalisation=norm, knn_idx=idx, knn_d2=d2)


def host_strategies(strategies):
    """The model's strategy lists (`get_spatial_ops()`) as host strategies with the same call signatures: only
    `_grid_size`, `_radius`, `_max_num_neighbors` and `k` are read.  They are what a caller of the reference's own
    `MultiScaleTransform` hands it to run the precompute over this package on CPU tensors (tests/golden/check_dropin.py)."""
    return {"sampler": [None if not s else HostGridSampler(s._grid_size) for s in strategies["sampler"]],
            "neighbour_finder": [HostRadiusFinder(f._radius, f._max_num_neighbors) for f in strategies["neighbour_finder"]],
            "upsample_op": [HostKnnTable(u.k) for u in strategies["upsample_op"]]}


class MultiScaleTransformCPU(object):
    def __init__(self, strategies):
        """strategies: {"sampler": [...], "neighbour_finder": [...], "upsample_op": [...]} as the model lists them
        (`get_spatial_ops()`) or already host strategies; only their parameters are read."""
        host = host_strategies(strategies)
        self.levels = [(None if s is None else s._grid_size, f._radius, f._max_num_neighbors)
                       for s, f in zip(host["sampler"], host["neighbour_finder"])]
        self.up_k = [u.k for u in host["upsample_op"]]
        self.chain = LevelChain(
            [(s, lambda parent, child, f=f: f(parent.pos, child.pos, batch_x=parent.batch, batch_y=child.batch))
             for s, f in zip(host["sampler"], host["neighbour_finder"])],
            [u.precompute for u in host["upsample_op"]])

    @classmethod
    def from_parameters(cls, levels, up_k):
        """levels: [(voxel edge or None, radius, max_num_neighbors)] per block; up_k: k of every decoder stage"""
        return cls({"sampler": [None if g is None else HostGridSampler(g) for g, _, _ in levels],
                    "neighbour_finder": [HostRadiusFinder(r, m) for _, r, m in levels],
                    "upsample_op": [HostKnnTable(k) for k in up_k]})

    def __call__(self, data):
        """data: CPU pos (N,3) [, batch (N,) sorted] -> data + multiscale=[...], upsample=[...]"""
        pos = data.pos.detach().float().contiguous()
        if pos.device.type != "cpu":
            raise RuntimeError("MultiScaleTransformCPU works on CPU tensors (use multiscale.MultiScaleTransform on the device)")
        batch = getattr(data, "batch", None)
        if batch is None:
            batch = torch.zeros(pos.shape[0], dtype=torch.long)
        clouds, tables = self.chain.run(PDData(pos=pos, batch=batch))
        return attach(data, clouds, tables)

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
