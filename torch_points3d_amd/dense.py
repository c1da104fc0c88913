"""Host-side mirror of the reference's DENSE-format PointNet++ building blocks.

Same class names, constructor arguments, attribute names (hence state_dict keys) and tensor contracts as
  torch_points3d/core/common_modules/dense_modules.py:5-29      (Conv2D / Conv1D / MLP2D)
  torch_points3d/core/spatial_ops/sampling.py:84-100            (DenseFPSSampler)
  torch_points3d/core/spatial_ops/neighbour_finder.py:89-178    (DenseRadiusNeighbourFinder)
  torch_points3d/core/base_conv/dense.py:30-184                 (BaseDenseConvolutionDown/Up, DenseFPModule,
                                                                 GlobalDenseBaseModule)
  torch_points3d/modules/pointnet2/dense.py:11-75               (PointNetMSGDown)
so a reference checkpoint loads unchanged and the parity tests read like the reference's own.

Every spatial op goes through a `kernels` namespace exposing the torch_points_kernels functions; the default
is the HIP implementation (torch_points3d_amd.torchpoints).  Tests and the CPU-baseline leg of bench.py
pass the CPU oracle instead -- the product code itself never imports it.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused as _fused
from . import torchpoints as _hip_kernels


class Data(object):
    """Minimal attribute bag standing in for torch_geometric.data.Data (pos, x, ...)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]

    def to(self, device):
        for k, v in self.__dict__.items():
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self

    def __repr__(self):
        parts = ["%s=%s" % (k, list(v.shape) if torch.is_tensor(v) else v) for k, v in self.__dict__.items()]
        return "Data(%s)" % ", ".join(parts)


def _leaky():
    return nn.LeakyReLU(negative_slope=0.01)


class Seq(nn.Sequential):
    """nn.Sequential whose children are named "0", "1", ... in append order (reference base_modules.py:156-164)."""

    def append(self, module):
        self.add_module(str(len(self)), module)
        return self


class Conv2D(Seq):
    def __init__(self, in_channels, out_channels, bias=True, bn=True, activation=None):
        super().__init__()
        activation = _leaky() if activation is None else activation
        self.append(nn.Conv2d(in_channels, out_channels, kernel_size=(1, 1), stride=(1, 1), bias=bias))
        if bn:
            self.append(nn.BatchNorm2d(out_channels))
        if activation:
            self.append(activation)


class Conv1D(Seq):
    def __init__(self, in_channels, out_channels, bias=True, bn=True, activation=None):
        super().__init__()
        activation = _leaky() if activation is None else activation
        self.append(nn.Conv1d(in_channels, out_channels, kernel_size=1, bias=bias))
        if bn:
            self.append(nn.BatchNorm1d(out_channels))
        if activation:
            self.append(activation)


class MLP2D(Seq):
    """Stack of 1x1 Conv2d -> BatchNorm2d -> LeakyReLU(0.01) over (B, C, npoint, nsample)."""

    def __init__(self, channels, bias=False, bn=True, activation=None):
        super().__init__()
        for cin, cout in zip(channels[:-1], channels[1:]):
            self.append(Conv2D(cin, cout, bn=bn, bias=bias, activation=activation))


class DenseFPSSampler(object):
    """Furthest point sampling of a (B,N,3) cloud; exactly num_to_sample or floor(N*ratio) points."""

    def __init__(self, ratio=None, num_to_sample=None, kernels=None):
        if (ratio is None) == (num_to_sample is None):
            raise ValueError("Specify exactly one of ratio and num_to_sample")
        self._ratio = ratio
        self._num_to_sample = num_to_sample
        self._tp = kernels or _hip_kernels

    def _get_num_to_sample(self, n):
        return self._num_to_sample if self._num_to_sample is not None else math.floor(n * self._ratio)

    def sample(self, pos, **kwargs):
        if pos.dim() != 3:
            raise ValueError(" This class is for dense data and expects the pos tensor to be of dimension 2")
        return self._tp.furthest_point_sample(pos, self._get_num_to_sample(pos.shape[1]))

    __call__ = sample


def _is_list(v):
    return isinstance(v, (list, tuple))


class DenseRadiusNeighbourFinder(object):
    """Multi-scale dense radius search; scalar/list broadcasting as neighbour_finder.py:102-135."""

    def __init__(self, radius, max_num_neighbors=64, kernels=None):
        if _is_list(radius) and _is_list(max_num_neighbors):
            if len(radius) != len(max_num_neighbors):
                raise ValueError("Both lists max_num_neighbors and radius should be of the same length")
            self._radius, self._max_num_neighbors = list(radius), list(max_num_neighbors)
        elif _is_list(radius):
            self._radius, self._max_num_neighbors = list(radius), [max_num_neighbors] * len(radius)
        elif _is_list(max_num_neighbors):
            self._radius, self._max_num_neighbors = [radius] * len(max_num_neighbors), list(max_num_neighbors)
        else:
            self._radius, self._max_num_neighbors = [radius], [max_num_neighbors]
        self._tp = kernels or _hip_kernels

    @property
    def num_scales(self):
        return len(self._radius)

    def find_neighbours(self, x, y, scale_idx=0):
        if scale_idx >= self.num_scales:
            raise ValueError("Scale %i is out of bounds %i" % (scale_idx, self.num_scales))
        return self._tp.ball_query(self._radius[scale_idx], self._max_num_neighbors[scale_idx], x, y)[0]

    def __call__(self, x, y, scale_idx=0, **kwargs):
        return self.find_neighbours(x, y, scale_idx)


def _is_channel_last(t):
    return t.dim() == 3 and t.stride(1) == 1 and t.transpose(1, 2).is_contiguous()


def _use_fused(tp, *tensors):
    """The fused channel-last kernels serve the HIP product path only (never the oracle-driven CPU graph)."""
    return tp is _hip_kernels and all(t is None or t.is_cuda for t in tensors)


class BaseDenseConvolutionDown(nn.Module):
    """sample -> gather centroids -> per scale (radius search + conv) -> concat   (reference dense.py:60-87)."""

    def __init__(self, sampler, neighbour_finder, *args, **kwargs):
        super().__init__()
        self.sampler = sampler
        self.neighbour_finder = neighbour_finder
        self._index = kwargs.get("index", None)
        self._save_sampling_id = kwargs.get("save_sampling_id", None)

    def conv(self, x, pos, new_pos, radius_idx, scale_idx):
        raise NotImplementedError

    def precompute(self, pos, backward_tables=False):
        """The geometry of this level for the cloud `pos` (B,N,3): everything forward() derives from positions alone --
        sampled indices, sampled positions, one neighbour table per scale.  Pass the result as forward(precomputed=).
        backward_tables: also the inverted form of every neighbour table, which the backward pass of the fused grouping
        gathers through (fused.scatter_table) -- geometry as well, so it can leave the training pass with the rest."""
        with torch.no_grad():
            idx = self.sampler(pos).long()
            new_pos = pos.gather(1, idx.unsqueeze(-1).repeat(1, 1, pos.shape[-1]))
            tables = [self.neighbour_finder(pos, new_pos, scale_idx=s) for s in range(self.neighbour_finder.num_scales)]
            inverse = [None] * len(tables)
            if backward_tables and pos.is_cuda and getattr(self, "fused", False):
                inverse = [_fused.scatter_table(t, None, pos.shape[1], 1) for t in tables]
        return Data(idx=idx, new_pos=new_pos, radius_idx=tables, radius_inv=inverse)

    def forward(self, data, sample_idx=None, precomputed=None, **kwargs):
        x, pos = data.x, data.pos
        if precomputed is not None:
            idx = precomputed.idx.unsqueeze(-1)
            new_pos = precomputed.new_pos
        else:
            idx = sample_idx if sample_idx is not None else self.sampler(pos)
            idx = idx.unsqueeze(-1).repeat(1, 1, pos.shape[-1]).long()
            new_pos = pos.gather(1, idx)
        ms_x = []
        for scale_idx in range(self.neighbour_finder.num_scales):
            radius_idx = (precomputed.radius_idx[scale_idx] if precomputed is not None
                          else self.neighbour_finder(pos, new_pos, scale_idx=scale_idx))
            table = None if precomputed is None else getattr(precomputed, "radius_inv", [None] * (scale_idx + 1))[scale_idx]
            ms_x.append(self.conv(x, pos, new_pos, radius_idx, scale_idx, **({} if table is None else {"table": table})))
        if all(_is_channel_last(t) for t in ms_x):
            # fused path: scales are (B, C_i, np) VIEWS of channel-last storage; concatenate the storage itself
            new_x = ms_x[0] if len(ms_x) == 1 else torch.cat([t.transpose(1, 2) for t in ms_x], 2).transpose(1, 2)
        else:
            new_x = torch.cat(ms_x, 1)
        new_data = Data(pos=new_pos, x=new_x)
        if self._save_sampling_id:
            setattr(new_data, "sampling_id_{}".format(self._index), idx[:, :, 0])
        return new_data


class PointNetMSGDown(BaseDenseConvolutionDown):
    """PointNet++ set abstraction (single or multi scale): group -> centre -> [xyz, feats] -> MLP2D -> max."""

    def __init__(self, npoint=None, radii=None, nsample=None, down_conv_nn=None, bn=True, activation=None,
                 use_xyz=True, normalize_xyz=False, kernels=None, fused=True, **kwargs):
        assert len(radii) == len(nsample) == len(down_conv_nn)
        tp = kernels or _hip_kernels
        super().__init__(DenseFPSSampler(num_to_sample=npoint, kernels=tp),
                         DenseRadiusNeighbourFinder(radii, nsample, kernels=tp), **kwargs)
        self._tp = tp
        self.fused = fused
        self.use_xyz = use_xyz
        self.npoint = npoint
        self.radii = radii
        self.normalize_xyz = normalize_xyz
        self.mlps = nn.ModuleList(MLP2D(c, bn=bn, activation=activation, bias=False) for c in down_conv_nn)

    def _prepare_features(self, x, pos, new_pos, idx, scale_idx):
        grouped_pos = self._tp.grouping_operation(pos.transpose(1, 2).contiguous(), idx)  # (B,3,np,ns)
        grouped_pos = grouped_pos - new_pos.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            grouped_pos = grouped_pos / self.radii[scale_idx]
        if x is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_pos
        grouped_features = self._tp.grouping_operation(x, idx)
        if self.use_xyz:
            return torch.cat([grouped_pos, grouped_features], dim=1)  # (B,3+C,np,ns), xyz first
        return grouped_features

    def _conv_fused(self, parts, x, pos, new_pos, radius_idx, scale_idx, table=None):
        """Same arithmetic on (rows, C) activations: HIP gather/BN/act/pool kernels around library GEMMs."""
        B, npnt, ns = radius_idx.shape
        x_cl = None
        if x is not None and self.use_xyz:
            x_cl = _fused._cl(x)
        elif x is not None:
            return None  # use_xyz=False never occurs in the bundled configs: keep it on the reference graph
        rows = _fused.group_concat(pos, new_pos, x_cl, radius_idx, self.radii[scale_idx], self.normalize_xyz, table)
        pooled = _fused.run_mlp(rows, parts, pool_ns=ns)  # (B*np, Cout)
        return pooled.view(B, npnt, -1).transpose(1, 2)  # (B, Cout, np) view of channel-last storage

    def conv(self, x, pos, new_pos, radius_idx, scale_idx, table=None):
        assert scale_idx < len(self.mlps)
        if self.fused and _use_fused(self._tp, x, pos):
            parts = _fused.mlp_parts(self.mlps[scale_idx])
            if parts is not None:
                out = self._conv_fused(parts, x, pos, new_pos, radius_idx, scale_idx, table)
                if out is not None:
                    return out
        feats = self._prepare_features(x, pos, new_pos, radius_idx, scale_idx)
        feats = self.mlps[scale_idx](feats)
        feats = F.max_pool2d(feats, kernel_size=[1, feats.size(3)])
        return feats.squeeze(-1)


class BaseDenseConvolutionUp(nn.Module):
    def __init__(self, neighbour_finder=None, *args, **kwargs):
        super().__init__()
        self.neighbour_finder = neighbour_finder
        self._index = kwargs.get("index", None)
        self._skip = kwargs.get("skip", True)

    def conv(self, pos, pos_skip, x):
        raise NotImplementedError

    def forward(self, data, precomputed=None, **kwargs):
        data, data_skip = data
        fused_out = self._forward_fused(data, data_skip, precomputed) if hasattr(self, "_forward_fused") else None
        if fused_out is not None:
            return Data(x=fused_out, pos=data_skip.pos)
        new_features = self.conv(data.pos, data_skip.pos, data.x, precomputed)
        if data_skip.x is not None:
            new_features = torch.cat([new_features, data_skip.x], dim=1)
        new_features = new_features.unsqueeze(-1)
        if hasattr(self, "nn"):
            new_features = self.nn(new_features)
        return Data(x=new_features.squeeze(-1), pos=data_skip.pos)


class DenseFPModule(BaseDenseConvolutionUp):
    """PointNet++ feature propagation: inverse-distance 3-NN interpolation + skip concat + MLP2D."""

    def __init__(self, up_conv_nn, bn=True, bias=False, activation=None, kernels=None, fused=True, **kwargs):
        super().__init__(None, **kwargs)
        self._tp = kernels or _hip_kernels
        self.fused = fused
        self.nn = MLP2D(up_conv_nn, bn=bn, activation=activation, bias=False)

    def precompute(self, pos, pos_skip, backward_tables=False):
        """3-NN interpolation table of this stage: (idx (B,n,3), inverse-distance weights (B,n,3)); None below the
        global module (pos is None there).  backward_tables: also its inverted form (fused.scatter_table)."""
        if pos is None:
            return None
        with torch.no_grad():
            dist, idx = self._tp.three_nn(pos_skip, pos)
            table = None
            if self._tp is _hip_kernels and dist.is_cuda:
                weight = _fused.idw_weights(dist)
                if backward_tables and self.fused:
                    table = _fused.scatter_table(idx, weight, pos.shape[1], 3)
            else:
                dist_recip = 1.0 / (dist + 1e-8)
                weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
        return Data(idx=idx, weight=weight, table=table)

    def _forward_fused(self, data, data_skip, precomputed=None):
        """interpolate + skip concat + MLP on (rows, C): HIP kernels around library GEMMs; None -> reference graph."""
        if not (self.fused and _use_fused(self._tp, data.x, data_skip.pos, data_skip.x)):
            return None
        parts = _fused.mlp_parts(self.nn)
        if parts is None:
            return None
        pos, pos_skip, x, x_skip = data.pos, data_skip.pos, data.x, data_skip.x
        B, n = pos_skip.shape[0], pos_skip.shape[1]
        skip_cl = None if x_skip is None else _fused._cl(x_skip)
        if pos is None:  # below the global module: one feature column broadcast to every skip point
            rows = x.transpose(1, 2).expand(B, n, x.shape[1])
            rows = _fused.cat_rows([rows] + ([skip_cl] if skip_cl is not None else []))
        elif precomputed is not None:
            rows = _fused.interp_concat(_fused._cl(x), precomputed.idx, precomputed.weight, skip_cl,
                                        getattr(precomputed, "table", None))
        else:
            dist, idx = self._tp.three_nn(pos_skip, pos)
            rows = _fused.interp_concat(_fused._cl(x), idx, _fused.idw_weights(dist), skip_cl)
        out = _fused.run_mlp(rows, parts)
        return out.view(B, n, -1).transpose(1, 2)

    def conv(self, pos, pos_skip, x, precomputed=None):
        assert pos_skip.shape[2] == 3
        if pos is None:  # below the global module: broadcast the single feature column
            return x.expand(*(x.size()[0:2] + (pos_skip.size(1),)))
        if precomputed is not None:
            return self._tp.three_interpolate(x, precomputed.idx, precomputed.weight)
        dist, idx = self._tp.three_nn(pos_skip, pos)
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
        return self._tp.three_interpolate(x, idx, weight)


class GlobalDenseBaseModule(nn.Module):
    """MLP2D over [x, pos] of all points then max/mean over points; returns pos=None (dense.py:169-184)."""

    def __init__(self, nn, aggr="max", bn=True, activation=None, fused=True, **kwargs):
        super().__init__()
        self.fused = fused
        self.nn = MLP2D(nn, bn=bn, activation=activation, bias=False)
        if aggr.lower() not in ["mean", "max"]:
            raise Exception("The aggregation provided is unrecognized {}".format(aggr))
        self._aggr = aggr.lower()

    def forward(self, data, **kwargs):
        if self.fused and self._aggr == "max" and data.x.is_cuda:
            parts = _fused.mlp_parts(self.nn)
            if parts is not None:
                B, n = data.pos.shape[0], data.pos.shape[1]
                rows = _fused.cat_rows([_fused._cl(data.x), data.pos])  # channel order [x, pos]
                pooled = _fused.run_mlp(rows, parts, pool_ns=n)  # max over all points of a cloud
                return Data(x=pooled.view(B, -1, 1), pos=None)
        x = self.nn(torch.cat([data.x, data.pos.transpose(1, 2).contiguous()], dim=1).unsqueeze(-1)).squeeze(-1)
        x = x.max(-1)[0] if self._aggr == "max" else x.mean(-1)
        return Data(x=x.unsqueeze(-1), pos=None)
