"""Partial-dense decoder pieces on the HIP kNN: KNNInterpolate, FPModule_PD, the Linear/BatchNorm/LeakyReLU `MLP`,
and KNNNeighbourFinder.

Mirrors (same constructor arguments, attribute names -- hence state_dict keys -- and forward contracts):
  * `KNNInterpolate`       torch_points3d/core/spatial_ops/interpolate.py:7-69
  * `FPModule_PD`          torch_points3d/core/base_conv/partial_dense.py:103-146
  * `MLP`                  torch_points3d/core/common_modules/base_modules.py:29-43 (Linear, FastBatchNorm1d, LeakyReLU(0.2))
  * `KNNNeighbourFinder`   torch_points3d/core/spatial_ops/neighbour_finder.py:42-47
The reference gets the neighbour search from torch_cluster (`knn`) and the weighted sum from torch_scatter; here both
are entry points of libtp3d_hip.so (csrc/knn.hip), and the backward scatter is the atomic-free inverse-index gather
(csrc/csr.hip) shared with the dense path.
"""
import torch
import torch.nn as nn

from . import _lib
from . import fused as _fused
from . import torchpoints as _tp
from .kpconv_blocks import FastBatchNorm1d


class _KnnInterpolate(torch.autograd.Function):
    """out (Nq, ld) = [ idw-interpolated x | skip | 0 ];  differentiable wrt x and skip."""

    @staticmethod
    def forward(ctx, x, skip, idx, dist2, ld):
        dev = x.device
        xf = x.detach().float().contiguous()
        Nq, k = idx.shape
        M, C = xf.shape
        C2 = 0 if skip is None else skip.shape[1]
        sk = None if skip is None else skip.detach().float().contiguous()
        out = torch.empty((Nq, ld), dtype=torch.float32, device=dev)
        wnorm = torch.empty((Nq, k), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.call("tp3d_knn_interpolate_fwd_f32", _lib.ptr(xf), _lib.ptr(idx), _lib.ptr(dist2), _lib.ptr(sk), Nq, k,
                      C, C2, ld, _lib.ptr(out), _lib.ptr(wnorm), _lib.stream_ptr(dev))
        ctx.save_for_backward(idx, wnorm)
        ctx.cfg = (M, C, C2, ld)
        return out

    @staticmethod
    def backward(ctx, g):
        idx, wnorm = ctx.saved_tensors
        M, C, C2, ld = ctx.cfg
        g = g.float().contiguous()
        dev = g.device
        Nq, k = idx.shape
        dx = dskip = None
        if ctx.needs_input_grad[0]:
            # slots without a neighbour (-1) go to an extra bin that is dropped afterwards
            safe = torch.where(idx < 0, torch.full_like(idx, M), idx).reshape(1, Nq * k).contiguous()
            dxp = torch.empty((M + 1, C), dtype=torch.float32, device=dev)
            ws, nbytes = _lib.scatter_workspace(1, Nq * k, M + 1, True, dev)
            with _lib.on_device(dev):
                _lib.call("tp3d_rows_scatter_bwd_f32", _lib.ptr(g), _lib.ptr(safe), _lib.ptr(wnorm), 1, Nq * k, k, M + 1,
                          ld, 0, C, _lib.ptr(dxp), _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
            dx = dxp[:M]
        if C2 and ctx.needs_input_grad[1]:
            dskip = g[:, C:C + C2]
        return dx, dskip, None, None, None


def knn_interpolate(x, pos_x, pos_y, batch_x=None, batch_y=None, k=3, skip=None, cell=0.0):
    """torch_geometric's knn_interpolate: features x at pos_x -> inverse-squared-distance blend at pos_y (Nq, C).
    With `skip` (Nq, C2) the result is [interpolated | skip] in one pass (FPModule_PD's concatenation)."""
    with torch.no_grad():
        idx, d2 = _tp.knn(k, pos_x, pos_y, batch_x, batch_y, cell=cell)
    C = x.shape[1] + (0 if skip is None else skip.shape[1])
    return _KnnInterpolate.apply(x, skip, idx, d2, C)


class KNNInterpolate(object):
    def __init__(self, k):
        self.k = k

    def precompute(self, query, support):
        """Neighbours and weights for the reference's precomputed path (interpolate.py:11-32): PDData-like bag with
        x_idx, y_idx, weights, normalisation, num_nodes."""
        from .kpconv_blocks import PDData
        pos_x, pos_y = query.pos, support.pos
        idx, d2 = _tp.knn(self.k, pos_x, pos_y, getattr(query, "batch", None), getattr(support, "batch", None))
        Nq = pos_y.shape[0]
        y_idx = torch.arange(Nq, device=pos_y.device).repeat_interleave(self.k)
        x_idx = idx.reshape(-1)
        keep = x_idx >= 0
        weights = (1.0 / torch.clamp(d2.reshape(-1, 1), min=1e-16))[keep]
        y_idx, x_idx = y_idx[keep], x_idx[keep]
        normalisation = torch.zeros((Nq, 1), device=pos_y.device).index_add_(0, y_idx, weights)
        # knn_idx / knn_d2: the same table in the fixed-k layout the fused interpolation kernel reads
        return PDData(num_nodes=Nq, x_idx=x_idx, y_idx=y_idx, weights=weights, normalisation=normalisation,
                      knn_idx=idx, knn_d2=d2)

    def __call__(self, query, support, precomputed=None, skip=None):
        """query: low-resolution data (pos, x[, batch]); support: the positions to interpolate to."""
        if precomputed:
            num_points = support.pos.size(0)
            if num_points != precomputed.num_nodes:
                raise ValueError("Precomputed indices do not match with the data given to the transform")
            x = query.x
            if getattr(precomputed, "knn_idx", None) is not None and x.is_cuda:
                C = x.shape[1] + (0 if skip is None else skip.shape[1])
                return _KnnInterpolate.apply(x, skip, precomputed.knn_idx, precomputed.knn_d2, C)
            y = torch.zeros((num_points, x.shape[1]), dtype=x.dtype, device=x.device).index_add_(
                0, precomputed.y_idx, x[precomputed.x_idx] * precomputed.weights)
            y = y / precomputed.normalisation
            return y if skip is None else torch.cat([y, skip], dim=1)
        grid = getattr(query, "grid_size", None)  # set by GridSampling3D (a CPU tensor): the natural search-cell edge
        cell = float(grid[0]) if torch.is_tensor(grid) and grid.numel() and not grid.is_cuda else 0.0
        return knn_interpolate(query.x, query.pos, support.pos, getattr(query, "batch", None),
                               getattr(support, "batch", None), k=self.k, skip=skip, cell=cell)


class MLP(nn.Sequential):
    """[Linear(bias) -> FastBatchNorm1d -> LeakyReLU(0.2)] per consecutive channel pair (base_modules.py:29-43)."""

    def __init__(self, channels, activation=None, bn_momentum=0.1, bias=True):
        activation = activation if activation is not None else nn.LeakyReLU(0.2)
        super().__init__(*[
            nn.Sequential(nn.Linear(channels[i - 1], channels[i], bias=bias),
                          FastBatchNorm1d(channels[i], momentum=bn_momentum), activation)
            for i in range(1, len(channels))])


class FPModule_PD(nn.Module):
    """Upsampling module: interpolate the coarse features onto the skip resolution, concatenate, MLP."""

    def __init__(self, up_k, up_conv_nn, *args, **kwargs):
        super().__init__()
        self.upsample_op = KNNInterpolate(up_k)
        bn_momentum = kwargs.get("bn_momentum", 0.1)
        self.nn = MLP(up_conv_nn, bn_momentum=bn_momentum, bias=False)
        self.fused = kwargs.get("fused", True)

    def forward(self, data, precomputed=None, **kwargs):
        data, data_skip = data
        # (the reference deep-copies data_skip; only .x is replaced below, so sharing the other tensors is equivalent)
        batch_out = data_skip.shallow_copy() if hasattr(data_skip, "shallow_copy") else data_skip.clone()
        x_skip = data_skip.x
        # reference: len(data.x) == data.batch.max() + 1, i.e. one (globally pooled) feature row per cloud; the cloud
        # count comes from the cached segment table of the skip level's batch vector (no extra host read)
        skip_batch = getattr(data_skip, "batch", None)
        nclouds = 1 if skip_batch is None else _tp._segments(_tp._i64(skip_batch))[1]
        has_innermost = len(data.x) == nclouds
        pre_data = None
        if precomputed and not has_innermost:
            if not hasattr(data, "up_idx"):
                batch_out.up_idx = 0
            else:
                batch_out.up_idx = data.up_idx
            pre_data = precomputed[batch_out.up_idx]
            batch_out.up_idx = batch_out.up_idx + 1
        if has_innermost:
            x = torch.gather(data.x, 0, data_skip.batch.unsqueeze(-1).repeat((1, data.x.shape[-1])))
            if x_skip is not None:
                x = torch.cat([x, x_skip], dim=1)
        else:
            x = self.upsample_op(data, data_skip, precomputed=pre_data, skip=x_skip)
        batch_out.x = (_fused.rows_mlp(self.nn, x) if self.fused else self.nn(x)) if hasattr(self, "nn") else x
        return batch_out


class KNNNeighbourFinder(object):
    """k nearest neighbours as an edge list (2, Nq*k): row 0 = query index, row 1 = support index, queries in order,
    neighbours closest first (torch_cluster `knn` layout; reference neighbour_finder.py:42-47)."""

    def __init__(self, k):
        self.k = k

    def find_neighbours(self, x, y, batch_x, batch_y):
        idx, _ = _tp.knn(self.k, x, y, batch_x, batch_y)
        row = torch.arange(y.shape[0], device=y.device).repeat_interleave(self.k)
        col = idx.reshape(-1)
        keep = col >= 0
        return torch.stack([row[keep], col[keep]], dim=0)

    def __call__(self, x, y, batch_x, batch_y):
        return self.find_neighbours(x, y, batch_x, batch_y)
