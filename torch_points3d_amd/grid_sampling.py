"""GridSampling3D on the device: the sampler of every strided KPConv block.

Mirrors torch_points3d/core/data_transform/grid_transform.py:33-141 (`group_data`, `GridSampling3D`): same constructor
arguments, same `mode` semantics ("mean": positions/features averaged per voxel, integer label keys by majority vote,
`batch` taken from the voxel's representative point; "last": one point per voxel after a random shuffle), same
`coords` / `grid_size` attributes on the output.  The reference builds this from torch_cluster.grid_cluster,
torch_geometric.voxel_grid / consecutive_cluster and torch_scatter (none importable here); this module calls the
voxel-clustering entry points of libtp3d_hip.so instead (include/tp3d_hip.h, csrc/voxel.hip).

Works on any attribute bag (`PDData`, the reference's torch_geometric `Data`): every tensor attribute whose first
dimension equals the number of points is grouped, the rest is carried over.
Parity: the transform's own logic (which attribute is grouped how, coords, bool / label handling) is pinned by
tests/golden/grid_sampling.npz, produced by the reference's grid_transform.py itself; the third-party arithmetic under
it (torch_cluster / torch_scatter, absent) is unpinned and replaced by oracle/voxel_ref.py, a numpy restatement of the
published algorithms, checked against the reference's own test properties (test/test_grid_sampling.py:29-67).
"""
import re

import numpy as np
import torch

from . import _lib

_INTEGER_LABEL_KEYS = ["y", "instance_labels"]  # grid_transform.py:20
_ORIGIN_ID_KEY = "origin_id"  # SaveOriginalPosId.KEY (core/data_transform/transforms.py)


def _check(pos, batch):
    if not pos.is_cuda:
        raise RuntimeError("voxel clustering runs on the GPU only (no CPU fallback in this build)")
    if pos.dim() != 2 or pos.shape[1] != 3:
        raise ValueError("pos must be (N, 3)")
    if batch is not None and (batch.dim() != 1 or batch.shape[0] != pos.shape[0]):
        raise ValueError("batch must be (N,)")


def voxel_cluster(pos, batch, size, return_counts=False, coord_bounds=None):
    """-> (cluster (N,), unique_pos_indices (K,), order (N,), cluster_start (K+1,)); all int64 on pos.device.
    With return_counts=True also the number of clusters of every cloud (host int64 tensor) and the coordinate extent
    the key was built over (host int32 array: min x, y, z, max x, y, z, highest batch id, flag).
    coord_bounds: optional host sequence (min x, y, z, max x, y, z of round(pos / size), highest batch id) known to
    contain every point (e.g. derived from the bounding box of the cloud this one was sampled from); it replaces the
    device reduction and its host read.  The key only needs an enclosing extent, so a loose bound changes nothing.

    cluster[i]: consecutive id of the voxel of point i (ids ascend with (batch, z, y, x) of the voxel);
    unique_pos_indices[c]: highest point index inside voxel c; order / cluster_start: members of voxel c are
    order[cluster_start[c]:cluster_start[c+1]], ascending.
    """
    _check(pos, batch)
    dev = pos.device
    N = pos.shape[0]
    empty = torch.empty(0, dtype=torch.int64, device=dev)
    if N == 0:
        out = (empty, empty, empty, torch.zeros(1, dtype=torch.int64, device=dev))
        return out + (torch.zeros(0, dtype=torch.int64), np.zeros(8, np.int32)) if return_counts else out
    pos = pos.detach().contiguous().float()
    if batch is not None:
        batch = batch.contiguous().long()
    with _lib.on_device(dev):
        s = _lib.stream_ptr(dev)
        bounds = torch.empty(8, dtype=torch.int32, device=dev)
        _lib.call("tp3d_voxel_bounds_f32", _lib.ptr(pos), _lib.ptr(batch), N, float(size), _lib.ptr(bounds), s)
        if coord_bounds is None:
            bounds_host = np.ascontiguousarray(bounds.cpu().numpy())  # wait 1: extent of the voxel key
        else:
            # the caller's enclosing extent sizes the key; the true extent is still reduced on the device and is
            # checked against it after the one host read at the end (no wait here)
            bounds_host = np.ascontiguousarray(np.asarray(list(coord_bounds) + [0], dtype=np.int32))
        nbytes = _lib.load().tp3d_voxel_workspace_bytes(N)
        ws = _lib.workspace("voxel", nbytes, dev)
        cluster = torch.empty(N, dtype=torch.int64, device=dev)
        order = torch.empty(N, dtype=torch.int64, device=dev)
        start = torch.empty(N + 1, dtype=torch.int64, device=dev)
        last = torch.empty(N, dtype=torch.int64, device=dev)
        nclouds = int(bounds_host[6]) + 1 if batch is not None else 1
        meta = torch.empty(1 + max(nclouds, 1), dtype=torch.int64, device=dev)
        _lib.call("tp3d_voxel_cluster_f32", _lib.ptr(pos), _lib.ptr(batch), N, float(size),
                  bounds_host.ctypes.data, _lib.ptr(cluster), _lib.ptr(order), _lib.ptr(start), _lib.ptr(last),
                  _lib.ptr(meta), _lib.ptr(ws), nbytes, s)
        if coord_bounds is None:
            meta_host = meta.cpu()  # wait 2: number of occupied voxels (+ how many of them each cloud holds)
        else:
            both = torch.cat([meta, bounds.to(torch.int64)]).cpu()  # the only wait on this path
            meta_host, true_b = both[:meta.numel()], both[meta.numel():].numpy()
            inside = (true_b[7] == 0 and all(true_b[a] >= bounds_host[a] for a in range(3))
                      and all(true_b[3 + a] <= bounds_host[3 + a] for a in range(3))
                      and (batch is None or true_b[6] <= bounds_host[6]))
            if not inside:  # a stale or wrong hint: redo with the measured extent
                return voxel_cluster(pos, batch, size, return_counts=return_counts)
        K = int(meta_host[0])
    out = (cluster, last[:K], order, start[:K + 1])
    if not return_counts:
        return out
    ends = torch.cummax(meta_host[1:], 0)[0]  # running cluster count after each cloud (empty clouds recorded 0)
    # + the coordinate extent the key was built over (GridSampling3D hands it on to the next, coarser level)
    return out + (torch.diff(ends, prepend=ends.new_zeros(1)), bounds_host)


def cluster_mean(x, order, cluster_start):
    """scatter_mean of x (N, ...) over the clusters, members summed in ascending point order -> (K, ...) fp32."""
    K = cluster_start.shape[0] - 1
    flat = x.detach().reshape(x.shape[0], -1).contiguous().float()
    C = flat.shape[1]
    out = torch.empty((K, C), dtype=torch.float32, device=x.device)
    if K > 0 and C > 0:
        with _lib.on_device(x.device):
            _lib.call("tp3d_cluster_mean_f32", _lib.ptr(flat), _lib.ptr(order), _lib.ptr(cluster_start), K, C,
                      _lib.ptr(out), _lib.stream_ptr(x.device))
    return out.reshape((K,) + tuple(x.shape[1:]))


def cluster_majority(labels, order, cluster_start):
    """Most frequent label per cluster, ties -> the lowest label (one-hot scatter_add + argmax in the reference)."""
    K = cluster_start.shape[0] - 1
    lab = labels.contiguous().long()
    out = torch.empty(K, dtype=torch.int64, device=labels.device)
    if K > 0:
        lo, hi = int(lab.min().item()), int(lab.max().item())
        with _lib.on_device(labels.device):
            _lib.call("tp3d_cluster_majority_i64", _lib.ptr(lab), _lib.ptr(order), _lib.ptr(cluster_start), K, lo,
                      hi - lo + 1, _lib.ptr(out), _lib.stream_ptr(labels.device))
    return out.to(labels.dtype)


def _items(data):
    keys = data.keys if hasattr(data, "keys") and not callable(data.keys) else list(vars(data).keys())
    for key in list(keys):
        item = getattr(data, key, None)
        if item is not None:
            yield key, item


def _num_nodes(data):
    n = getattr(data, "num_nodes", None)
    return int(n) if n is not None else int(data.pos.shape[0])


def group_data(data, cluster=None, unique_pos_indices=None, mode="last", skip_keys=(), order=None, cluster_start=None):
    """In-place grouping of every per-point tensor of `data` (reference grid_transform.py:33-81)."""
    assert mode in ["mean", "last"]
    if mode == "mean" and cluster is None:
        raise ValueError("In mean mode the cluster argument needs to be specified")
    if mode == "last" and unique_pos_indices is None:
        raise ValueError("In last mode the unique_pos_indices argument needs to be specified")
    num_nodes = _num_nodes(data)
    if mode == "mean" and (order is None or cluster_start is None):
        # a cluster vector from elsewhere: rebuild the member lists (stable, so members stay in point order)
        order = torch.sort(cluster, stable=True)[1]
        counts = torch.bincount(cluster, minlength=int(cluster.max().item()) + 1 if cluster.numel() else 0)
        cluster_start = torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)])
    for key, item in _items(data):
        if bool(re.search("edge", key)):
            raise ValueError("Edges not supported. Wrong data type.")
        if key in skip_keys:
            continue
        if torch.is_tensor(item) and item.dim() > 0 and item.size(0) == num_nodes:
            if mode == "last" or key == "batch" or key == _ORIGIN_ID_KEY:
                setattr(data, key, item[unique_pos_indices])
            else:
                is_item_bool = item.dtype == torch.bool
                if is_item_bool:
                    item = item.int()
                if key in _INTEGER_LABEL_KEYS:
                    out = cluster_majority(item, order, cluster_start)
                elif item.is_floating_point():
                    out = cluster_mean(item, order, cluster_start).to(item.dtype)
                else:  # scatter_mean of an integer tensor: floor division of the sums
                    sums = torch.zeros((cluster_start.shape[0] - 1,) + tuple(item.shape[1:]), dtype=item.dtype,
                                       device=item.device).index_add_(0, cluster, item)
                    cnt = (cluster_start[1:] - cluster_start[:-1]).to(item.dtype)
                    out = torch.div(sums, cnt.reshape((-1,) + (1,) * (item.dim() - 1)), rounding_mode="floor")
                setattr(data, key, out.bool() if is_item_bool else out)
    if hasattr(data, "num_nodes") and getattr(data, "num_nodes") is not None and "num_nodes" in vars(data):
        data.num_nodes = int(unique_pos_indices.shape[0])
    return data


def shuffle_data(data):
    num_points = data.pos.shape[0]
    shuffle_idx = torch.randperm(num_points).to(data.pos.device)
    for key, item in _items(data):
        if torch.is_tensor(item) and item.dim() > 0 and num_points == item.shape[0]:
            setattr(data, key, item[shuffle_idx])
    return data


class GridSampling3D(object):
    """Clusters points into voxels of edge `size` (reference grid_transform.py:84-141)."""

    def __init__(self, size, quantize_coords=False, mode="mean", verbose=False):
        self._grid_size = size
        self._quantize_coords = quantize_coords
        self._mode = mode

    def _process(self, data):
        if self._mode == "last":
            data = shuffle_data(data)
        batch = getattr(data, "batch", None)
        # a cloud produced by a previous GridSampling3D carries the bounding box of its ancestors' voxel coordinates
        # (`pos_bounds`, host floats): voxel means stay inside it, so the key extent needs no device reduction here
        hint = None
        pb = getattr(data, "pos_bounds", None)
        if pb is not None and data.pos.shape[0] > 0:
            size = float(self._grid_size)
            nb = 0
            if batch is not None:
                from . import torchpoints as _tp
                nb = _tp._segments(_tp._i64(batch))[1] - 1  # cached for a batch vector the searches have seen
            hint = [int(np.floor(pb[0][a] / size)) - 1 for a in range(3)] + \
                   [int(np.ceil(pb[1][a] / size)) + 1 for a in range(3)] + [max(nb, 0)]
        cluster, unique_pos_indices, order, cluster_start, counts, b = voxel_cluster(
            data.pos, batch, self._grid_size, return_counts=True, coord_bounds=hint)
        if data.pos.shape[0] > 0:
            if pb is None:
                size = float(self._grid_size)
                pb = ([(int(b[a]) - 0.5) * size for a in range(3)], [(int(b[3 + a]) + 0.5) * size for a in range(3)])
            data.pos_bounds = pb
        if self._quantize_coords:
            # tensor / tensor is a true fp32 division on the device (tensor / python-scalar multiplies by 1/size)
            size_t = torch.full((), float(self._grid_size), dtype=torch.float32, device=data.pos.device)
            coords = torch.round(data.pos[unique_pos_indices].float() / size_t)
        data = group_data(data, cluster, unique_pos_indices, mode=self._mode, order=order, cluster_start=cluster_start)
        if self._quantize_coords:
            data.coords = coords.int()
        data.grid_size = torch.tensor([self._grid_size])
        new_batch = getattr(data, "batch", None)
        if batch is not None and torch.is_tensor(new_batch) and new_batch.dtype == torch.int64 and counts.numel():
            # the sampled batch vector's segment table is known from the clustering: spare the next radius / kNN
            # search on this level its own pass over the vector and its host read
            from . import torchpoints as _tp
            seg = torch.zeros(counts.numel() + 1, dtype=torch.int64)
            seg[1:] = torch.cumsum(counts, 0)
            _tp.prime_segments(new_batch, seg.to(new_batch.device), int(counts.numel()), int(counts.max()))
        return data

    def __call__(self, data):
        if isinstance(data, list):
            return [self._process(d) for d in data]
        return self._process(data)

    def __repr__(self):
        return "{}(grid_size={}, quantize_coords={}, mode={})".format(self.__class__.__name__, self._grid_size,
                                                                      self._quantize_coords, self._mode)
