"""CPU ORACLE for the host-side API `torch_points_kernels.points_cpu` and `region_grow` -- test infrastructure, NOT
product code (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/).

Brute-force numpy / torch restatement of what the reference's call sites require of torch-points-kernels 0.7.0's CPU
searches (source absent from the reference tree -- parity unpinned at the kernel boundary; conventions from
core/data_transform/transforms.py:805,853-857,887-890,919-920,1044 and datasets/registration/utils.py:150-166):
    ball_query(support, query, radius, max_num, mode, sorted) -> (ind, dist)
        mode 0: (Nq, W) support indices / squared distances, -1 / -1.0 in unused slots, W = max_num or the largest count
        mode 1: (P, 2) [support index, query index] pairs and (P, 1) squared distances
        hits in ascending support index (sorted: closest first, ties by index), at most max_num per query when > 0
and of region_grow (models/panoptic/pointgroup.py:101-115, test/test_pointgroup.py:28-39): connected components of the
radius graph among the points of one label and one cloud, grown from the lowest unvisited index."""
import numpy as np
import torch


def _hits(support, query, radius, max_num, sort):
    s, q = support.numpy().astype(np.float32), query.numpy().astype(np.float32)
    out = []
    r2 = np.float32(radius) * np.float32(radius)
    for j in range(q.shape[0]):
        d = s - q[j]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        idx = np.nonzero(d2 < r2)[0]
        if sort:
            idx = idx[np.lexsort((idx, d2[idx]))]
        if max_num and max_num > 0:
            idx = idx[:max_num]
        out.append((idx, d2[idx]))
    return out


def ball_query(support, query, radius, max_num, mode=0, sorted=False):
    hits = _hits(support, query, radius, max_num, sorted)
    nq = len(hits)
    if mode == 0:
        width = max_num if (max_num and max_num > 0) else max([len(h[0]) for h in hits] + [0])
        ind = -np.ones((nq, width), dtype=np.int64)
        dist = -np.ones((nq, width), dtype=np.float32)
        for j, (i, d) in enumerate(hits):
            ind[j, :len(i)], dist[j, :len(i)] = i, d
        return torch.from_numpy(ind), torch.from_numpy(dist)
    pairs = [np.stack([i, np.full_like(i, j)], 1) for j, (i, _) in enumerate(hits)]
    dist = [d for _, d in hits]
    return (torch.from_numpy(np.concatenate(pairs, 0).astype(np.int64)) if pairs else torch.zeros((0, 2), dtype=torch.long),
            torch.from_numpy(np.concatenate(dist, 0).astype(np.float32)).unsqueeze(1) if dist else torch.zeros((0, 1)))


def region_grow(pos, labels, batch, ignore_labels=(), radius=0.03, nsample=300, min_cluster_size=10):
    """list of LongTensors (index sets).  The neighbourhood of a point = the first `nsample` points of its cloud and
    label within `radius`, ascending index (the partial-dense radius search the library uses)."""
    ignore = set(int(v) for v in (ignore_labels.tolist() if torch.is_tensor(ignore_labels) else ignore_labels))
    clusters = []
    for label in torch.unique(labels).tolist():
        if int(label) in ignore:
            continue
        sel = torch.nonzero(labels == label).view(-1)
        p, b = pos[sel].float(), batch[sel]
        n = p.shape[0]
        nbrs = []
        for i in range(n):
            same = torch.nonzero(b == b[i]).view(-1)
            d = p[same] - p[i]
            d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            nbrs.append(same[d2 < np.float32(radius) * np.float32(radius)][:nsample].tolist())
        seen = [False] * n
        for start in range(n):
            if seen[start]:
                continue
            seen[start] = True
            members, queue = [start], [start]
            while queue:
                cur = queue.pop(0)
                for j in nbrs[cur]:
                    if not seen[j]:
                        seen[j] = True
                        members.append(j)
                        queue.append(j)
            if len(members) >= min_cluster_size:
                clusters.append(sel[torch.tensor(members)])
    return clusters
