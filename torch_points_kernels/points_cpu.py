"""``torch_points_kernels.points_cpu``: the host-side searches torch-points3d's data transforms and registration
dataset builders import (reference core/data_transform/transforms.py:16, datasets/registration/utils.py:8,
datasets/registration/basetest.py:33, datasets/registration/base_siamese_dataset.py:7).

    ball_query(support, query, radius, max_num, mode=0, sorted=False) -> (ind, dist)
        mode 0  ind (Nq, W) int64 support indices per query, dist (Nq, W) squared distances; W = max_num, or the
                largest hit count when max_num <= 0; unused slots hold -1 / -1.0
                (transforms.py:805, 1044:  `(dist > 0).sum(1)` counts the real non-self neighbours)
        mode 1  ind (P, 2) int64 pairs [support index, query index] of every hit (at most max_num per query when
                max_num > 0), dist (P, 1) squared distances  (transforms.py:853-857, 919-920: `ind[:, 0]` indexes the
                SUPPORT cloud; datasets/registration/utils.py:150-166 flips the columns to get (query, support) pairs)
        sorted  hits closest first (ties by index) instead of ascending support index
    dense_knn(support (B,N,3), query (B,nq,3), k) -> (ind (B,nq,k), dist (B,nq,k))   closest first

CPU tensors in, CPU tensors out.  The arithmetic lives in libtp3d_cpu.so (torch_points3d_amd/csrc_cpu/points_cpu.c,
C-ABI include/tp3d_cpu.h): a uniform grid over the support cloud; no GPU runtime is touched, no thread pool survives
a call, so it is safe inside forked DataLoader workers.  The reference binds torch-points-kernels 0.7.0 (nanoflann),
whose source is not in the reference tree: the conventions above are what its call sites require; where they leave a
choice (order of unsorted hits) ascending index is used.
"""
import ctypes
import os

import torch

from torch_points3d_amd import build as _build

_p, _l, _i, _f = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
_h = None
_threads = 1


def _lib():
    global _h
    if _h is None:
        path = _build.CPU_LIB_PATH
        if not os.path.exists(path):
            _build.build_cpu_library()
        h = ctypes.CDLL(path)
        h.tp3d_cpu_grid_build.restype = _p
        h.tp3d_cpu_grid_build.argtypes = [_p, _l, _f]
        h.tp3d_cpu_grid_free.restype = None
        h.tp3d_cpu_grid_free.argtypes = [_p]
        h.tp3d_cpu_ball_count.argtypes = [_p, _p, _l, _f, _p, _i]
        h.tp3d_cpu_ball_fill.argtypes = [_p, _p, _l, _f, _i, _i, _p, _p, _p, _i]
        h.tp3d_cpu_knn.argtypes = [_p, _p, _l, _i, _p, _p, _i]
        h.tp3d_cpu_grow_clusters.restype = _l
        h.tp3d_cpu_grow_clusters.argtypes = [_p, _l, _i, _l, _p, _p]
        if h.tp3d_cpu_abi_version() != 2:
            raise RuntimeError("libtp3d_cpu.so ABI mismatch")
        _h = h
    return _h


def set_num_threads(n):
    """worker threads per call (default 1: DataLoader workers already parallelise over samples)"""
    global _threads
    _threads = max(1, int(n))


def _xyz(t, name):
    if not torch.is_tensor(t) or t.device.type != "cpu":
        raise RuntimeError("points_cpu.%s expects CPU tensors" % name)
    if t.dim() != 2 or t.shape[1] != 3:
        raise ValueError("%s must be (N, 3), got %s" % (name, tuple(t.shape)))
    return t.detach().to(torch.float32).contiguous()


class _Grid(object):
    def __init__(self, pts, cell):
        self.pts = pts  # keeps the borrowed coordinates alive
        if pts.numel() and not bool(torch.isfinite(pts).all()):
            raise ValueError("points_cpu: the support cloud holds NaN or Inf coordinates")
        self.h = _lib().tp3d_cpu_grid_build(pts.data_ptr(), pts.shape[0], float(cell))
        if not self.h:
            raise MemoryError("points_cpu: grid build failed")

    def __del__(self):
        if getattr(self, "h", None):
            _lib().tp3d_cpu_grid_free(self.h)
            self.h = None


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("points_cpu.%s failed (code %d)" % (what, rc))


def ball_query(support, query, radius, max_num, mode=0, sorted=False):
    support, query = _xyz(support, "support"), _xyz(query, "query")
    if mode not in (0, 1):
        raise ValueError("mode must be 0 (matrix) or 1 (pairs)")
    radius = float(radius)
    nq = query.shape[0]
    grid = _Grid(support, max(radius, 1e-12))
    h = _lib()
    counts = torch.zeros(nq, dtype=torch.int64)
    _check(h.tp3d_cpu_ball_count(grid.h, query.data_ptr(), nq, radius, counts.data_ptr(), _threads), "ball_query")
    limit = int(max_num) if max_num is not None and int(max_num) > 0 else 0
    kept = counts.clamp(max=limit) if limit else counts
    if mode == 0:
        width = limit if limit else (int(kept.max()) if nq else 0)
        offsets = torch.arange(nq + 1, dtype=torch.int64) * width
        ind = torch.empty((nq, width), dtype=torch.int64)
        dist = torch.empty((nq, width), dtype=torch.float32)
        _check(h.tp3d_cpu_ball_fill(grid.h, query.data_ptr(), nq, radius, limit, int(bool(sorted)), offsets.data_ptr(),
                                    ind.data_ptr(), dist.data_ptr(), _threads), "ball_query")
        return ind, dist
    offsets = torch.zeros(nq + 1, dtype=torch.int64)
    torch.cumsum(kept, 0, out=offsets[1:])
    total = int(offsets[-1])
    hit = torch.empty(total, dtype=torch.int64)
    dist = torch.empty(total, dtype=torch.float32)
    _check(h.tp3d_cpu_ball_fill(grid.h, query.data_ptr(), nq, radius, limit, int(bool(sorted)), offsets.data_ptr(),
                                hit.data_ptr(), dist.data_ptr(), _threads), "ball_query")
    owner = torch.repeat_interleave(torch.arange(nq, dtype=torch.int64), kept)
    return torch.stack([hit, owner], 1), dist.unsqueeze(1)


def dense_knn(support, query, k):
    if support.dim() != 3 or query.dim() != 3 or support.shape[0] != query.shape[0]:
        raise ValueError("dense_knn expects support (B,N,3) and query (B,nq,3)")
    B, n, nq = support.shape[0], support.shape[1], query.shape[1]
    ind = torch.empty((B, nq, int(k)), dtype=torch.int64)
    dist = torch.empty((B, nq, int(k)), dtype=torch.float32)
    h = _lib()
    for b in range(B):
        pts, q = _xyz(support[b], "support"), _xyz(query[b], "query")
        # cell edge from the mean point density: ~2 points per cell keeps the first shells short
        ext = (pts.max(0)[0] - pts.min(0)[0]).clamp(min=1e-6) if n else torch.ones(3)
        cell = float((ext.prod() / max(n, 1) * 2.0) ** (1.0 / 3.0))
        grid = _Grid(pts, max(cell, 1e-6))
        _check(h.tp3d_cpu_knn(grid.h, q.data_ptr(), nq, int(k), ind[b].data_ptr(), dist[b].data_ptr(), _threads), "dense_knn")
    return ind, dist
