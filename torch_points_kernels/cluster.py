"""``torch_points_kernels.region_grow``: the clustering step of PointGroup (reference
torch_points3d/models/panoptic/pointgroup.py:3,101-115): for every semantic label that is not ignored, the points of
that label are grouped, cloud by cloud, into the sets connected through their radius neighbourhoods.

    region_grow(pos (N,3), labels (N,), batch (N,), ignore_labels=[], radius=0.03, nsample=300, min_cluster_size=10)
        -> list of LongTensors of point indices (into pos), one per cluster of at least min_cluster_size points

The neighbourhoods are this library's partial-dense ball query (the first `nsample` points within `radius`, ascending
index -- HIP kernel for device tensors, the host grid of points_cpu for CPU tensors); the growing itself is a plain
graph walk over that table on the host (libtp3d_cpu.so: tp3d_cpu_grow_clusters), which is what torch-points-kernels 0.7.0
does with a numba loop (its source is not in the reference tree: semantics restated from its published description --
clusters are grown from the lowest unvisited index; members of a cluster come in discovery order, which the reference
never relies on: it reads clusters as index sets, panoptic_losses.py:37, panoptic_tracker.py:194-207).
"""
import torch

from . import points_cpu as _cpu


def _neighbour_table(pos, batch, radius, nsample):
    """(n, nsample) int64 table of the points within `radius` of every point of its own cloud, -1 terminated rows"""
    if pos.is_cuda:
        from torch_points3d_amd.torchpoints import ball_query
        return ball_query(radius, nsample, pos, pos, mode="partial_dense", batch_x=batch, batch_y=batch)[0].cpu()
    table = torch.full((pos.shape[0], nsample), -1, dtype=torch.int64)
    for b in torch.unique(batch).tolist():
        sel = torch.nonzero(batch == b, as_tuple=False).view(-1)
        ind, _ = _cpu.ball_query(pos[sel], pos[sel], radius, nsample, mode=0, sorted=False)
        width = ind.shape[1]
        table[sel, :width] = torch.where(ind >= 0, sel[ind.clamp(min=0)], ind)
    return table


def grow_proximity(pos, batch, nsample=16, radius=0.02, min_cluster_size=32):
    """clusters of the points `pos` (n,3) with cloud ids `batch` (n,), as lists of row indices"""
    assert pos.shape[0] == batch.shape[0]
    n = pos.shape[0]
    if n == 0:
        return []
    table = _neighbour_table(pos.detach().float().contiguous(), batch.contiguous(), float(radius), int(nsample)).contiguous()
    members = torch.empty(n, dtype=torch.int64)
    starts = torch.empty(n + 1, dtype=torch.int64)
    kept = _cpu._lib().tp3d_cpu_grow_clusters(table.data_ptr(), n, table.shape[1], int(min_cluster_size), members.data_ptr(),
                                               starts.data_ptr())
    if kept < 0:
        raise RuntimeError("region_grow: tp3d_cpu_grow_clusters failed (code %d)" % kept)
    return [members[int(starts[c]):int(starts[c + 1])] for c in range(int(kept))]


def region_grow(pos, labels, batch, ignore_labels=[], radius=0.03, nsample=300, min_cluster_size=10):
    assert labels.dim() == 1 and pos.dim() == 2 and pos.shape[0] == labels.shape[0] == batch.shape[0]
    ignore = set(int(v) for v in (ignore_labels.tolist() if torch.is_tensor(ignore_labels) else ignore_labels))
    clusters = []
    for label in torch.unique(labels).tolist():
        if int(label) in ignore:
            continue
        sel = torch.nonzero(labels == label, as_tuple=False).view(-1)  # ascending: cloud ids stay sorted
        # cloud ids renumbered 0..k-1 (the partial-dense search wants consecutive ids)
        local_batch = torch.unique(batch[sel], return_inverse=True)[1]
        for cluster in grow_proximity(pos[sel], local_batch, nsample=nsample, radius=radius, min_cluster_size=min_cluster_size):
            clusters.append(sel[cluster.to(sel.device)])
    return clusters
