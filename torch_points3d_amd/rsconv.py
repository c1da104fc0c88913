"""Relation-Shape convolution (dense format) computed on channel-last rows.

Specification: torch_points3d/modules/RSConv/dense.py:18-190, 398-476 (RSConvMapper, SharedRSConv,
RSConvSharedMSGDown, RSConvMSGDown).  What those classes compute per scale, for every (centroid j, neighbour s) pair:

    h   = [ |p - c|, c, p, p - c ]                         10-vector geometric relation
    msg = MLP(h)                                           two 1x1 conv + BatchNorm + LeakyReLU(0.01) layers
    f   = [p - c, x[idx]]  (first layer: MLP-raised to the message width)
    out = max_s act(BatchNorm(f * msg));  then  act(BatchNorm(W out + b))   (channel raising)

Here one (B*np*ns, C) row matrix per tensor replaces the reference's (B, C, np, ns) layout: the relation rows come
from one HIP kernel (tp3d_relation_rows_f32), the feature rows from the set-abstraction gather kernel
(tp3d_group_concat_fwd_f32, which already emits [p - c, x[idx]]), every conv+BN+activation from the fused row kernels
of fused.py, and the pooled normalisation from tp3d_bn_act_maxpool_f32.  The absolute-xyz channels the reference
carries through its grouped tensor only to slice them off again are never materialised.

Only the PARAMETER CONTAINERS follow the reference (attribute names `_mapper`, `nn["features_nn" | "mlp_msg" |
"norm"]`, `mlps[i]._mapper`, `mlp_out`), so that its state_dict loads strictly.  With a `kernels` namespace other than
the HIP one (the CPU oracle in tests) the same row algorithm runs on plain torch ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused as _fused
from . import torchpoints as _hip_kernels
from .dense import BaseDenseConvolutionDown, DenseFPSSampler, DenseRadiusNeighbourFinder, MLP2D


# ------------------------------------------------------------------------------------------- row back-ends
def _gather_rows(t, idx):
    """t (B, N, C), idx (B, np, ns) -> (B*np*ns, C) rows t[b, idx[b,j,s]]"""
    B, _, C = t.shape
    flat = idx.reshape(B, -1, 1).expand(B, idx.shape[1] * idx.shape[2], C)
    return t.gather(1, flat).reshape(-1, C)


class _TorchRows(object):
    """plain torch ops on rows (any device): used when the spatial kernels are not the HIP ones"""

    @staticmethod
    def relation(pos, new_pos, idx):
        ns = idx.shape[2]
        p = _gather_rows(pos, idx)
        c = new_pos.reshape(-1, 1, 3).expand(-1, ns, 3).reshape(-1, 3)
        d = p - c
        return torch.cat([torch.sqrt((d * d).sum(1, keepdim=True)), c, p, d], 1)

    @staticmethod
    def features(pos, new_pos, x_cl, idx):
        ns = idx.shape[2]
        d = _gather_rows(pos, idx) - new_pos.reshape(-1, 1, 3).expand(-1, ns, 3).reshape(-1, 3)
        return d if x_cl is None else torch.cat([d, _gather_rows(x_cl, idx)], 1)

    @staticmethod
    def layer(rows, conv, bn, act):
        w = conv.weight.reshape(conv.out_channels, -1)
        y = F.linear(rows, w, conv.bias)
        return _TorchRows.norm(y, bn, act)

    @staticmethod
    def norm(y, bn, act, pool_ns=0):
        if bn.training:
            bn.num_batches_tracked.add_(1)
        y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
        y = act(y) if act is not None else y
        return y.reshape(-1, pool_ns, y.shape[1]).max(1)[0] if pool_ns else y


class _HipRows(object):
    """the fused HIP row kernels"""

    @staticmethod
    def relation(pos, new_pos, idx):
        return _fused.relation_rows(pos, new_pos, idx)

    @staticmethod
    def features(pos, new_pos, x_cl, idx):
        return _fused.group_concat(pos, new_pos, x_cl, idx, 1.0, False)

    @staticmethod
    def layer(rows, conv, bn, act):
        return _fused.linear_bn_act(rows, conv, bn, _fused._slope_of(act))

    @staticmethod
    def norm(y, bn, act, pool_ns=0):
        return _fused.bn_act(y, bn, _fused._slope_of(act), pool_ns)


def _layers(mlp):
    """[(conv, bn, activation)] of an MLP2D"""
    out = []
    for block in mlp.children():
        mods = list(block.children())
        out.append((mods[0], mods[1], mods[2] if len(mods) > 2 else None))
    return out


# ------------------------------------------------------------------------------------------- parameter containers
class RSConvMapper(nn.Module):
    """Holds the relation MLP (`mlp_msg`), the optional feature-raising MLP of a first layer (`features_nn`) and the
    normalisation applied to their product (`norm`); `down_conv_nn` is [f_in, f_mid, f_out] or, for a first layer,
    [[f_in, f_mid, f_out], [c_in, f_out]]."""

    def __init__(self, down_conv_nn, use_xyz, bn=True, activation=None, *args, **kwargs):
        super().__init__()
        self._use_xyz = use_xyz
        self._first_layer = len(down_conv_nn) == 2
        relation_widths = list(down_conv_nn[0] if self._first_layer else down_conv_nn)
        self._f_out = relation_widths[-1]
        self.nn = nn.ModuleDict()
        if self._first_layer:
            self.nn["features_nn"] = MLP2D(down_conv_nn[1], bn=bn, bias=False)
        self.nn["mlp_msg"] = MLP2D(relation_widths, bn=bn, bias=False)
        self.nn["norm"] = nn.Sequential(nn.BatchNorm2d(self._f_out),
                                        activation if activation is not None else nn.LeakyReLU(negative_slope=0.01))

    @property
    def f_out(self):
        return self._f_out

    def modulate(self, rows, relation, ns, ops):
        """(M, C) feature rows x (M, 10+) relation rows -> (M/ns, f_out): max over each centroid's ns pairs of
        act(BatchNorm(features * MLP(relation)))."""
        for conv, bn, act in _layers(self.nn["mlp_msg"]):
            relation = ops.layer(relation, conv, bn, act)
        if self._first_layer:
            for conv, bn, act in _layers(self.nn["features_nn"]):
                rows = ops.layer(rows, conv, bn, act)
        elif rows.shape[1] != self._f_out:  # the gather pads rows to a multiple of four columns
            rows = rows[:, :self._f_out]
        bn, act = self.nn["norm"][0], self.nn["norm"][1]
        return ops.norm(rows * relation, bn, act, pool_ns=ns)


class SharedRSConv(nn.Module):
    """One scale of a Relation-Shape layer: a radius and (a reference to) the mapper that serves it."""

    def __init__(self, mapper, radius):
        super().__init__()
        self._mapper = mapper
        self._radius = radius

    def __repr__(self):
        return "{}(radius={})".format(self.__class__.__name__, self._radius)


class _RelationShapeDown(BaseDenseConvolutionDown):
    """sample -> per scale (radius search, relation-modulated features, max) -> channel raising -> concat of scales"""

    def __init__(self, npoint, radii, nsample, channel_raising_nn, use_xyz, activation, kernels, **kwargs):
        assert len(radii) == len(nsample)
        tp = kernels or _hip_kernels
        super().__init__(DenseFPSSampler(num_to_sample=npoint, kernels=tp),
                         DenseRadiusNeighbourFinder(radii, nsample, kernels=tp), **kwargs)
        self._tp = tp
        self.use_xyz = use_xyz
        self.npoint = npoint
        self.mlps = nn.ModuleList()
        self.mlp_out = nn.Sequential(
            nn.Conv1d(channel_raising_nn[0], channel_raising_nn[-1], kernel_size=1, stride=1, bias=True),
            nn.BatchNorm1d(channel_raising_nn[-1]), activation)

    def conv(self, x, pos, new_pos, radius_idx, scale_idx):
        """x (B, C, N) or None, pos (B, N, 3), new_pos (B, np, 3), radius_idx (B, np, ns) -> (B, C_raised, np)"""
        assert scale_idx < len(self.mlps)
        if not self.use_xyz:
            raise NotImplementedError("use_xyz=False never occurs in the reference's RSConv configurations")
        B, npnt, ns = radius_idx.shape
        ops = _HipRows if (self._tp is _hip_kernels and pos.is_cuda) else _TorchRows
        x_cl = None if x is None else _fused._cl(x)
        rows = ops.features(pos, new_pos, x_cl, radius_idx)
        relation = ops.relation(pos, new_pos, radius_idx)
        pooled = self.mlps[scale_idx]._mapper.modulate(rows, relation, ns, ops)  # (B*np, f_out)
        raised = ops.layer(pooled, self.mlp_out[0], self.mlp_out[1], self.mlp_out[2])
        return raised.view(B, npnt, -1).transpose(1, 2)


class RSConvSharedMSGDown(_RelationShapeDown):
    """Multi-scale Relation-Shape set abstraction with ONE mapper shared by every scale."""

    def __init__(self, npoint=None, radii=None, nsample=None, down_conv_nn=None, channel_raising_nn=None, bn=True,
                 use_xyz=True, activation=None, kernels=None, **kwargs):
        activation = activation if activation is not None else nn.ReLU()
        super().__init__(npoint, radii, nsample, channel_raising_nn, use_xyz, activation, kernels, **kwargs)
        self._mapper = RSConvMapper(down_conv_nn, activation=activation, use_xyz=self.use_xyz)
        self.mlps.extend(SharedRSConv(self._mapper, r) for r in radii)


class RSConvMSGDown(_RelationShapeDown):
    """Multi-scale Relation-Shape set abstraction with one mapper per scale (`_mapper` = the last one built)."""

    def __init__(self, npoint=None, radii=None, nsample=None, down_conv_nn=None, channel_raising_nn=None, bn=True,
                 bias=True, use_xyz=True, activation=None, kernels=None, **kwargs):
        activation = activation if activation is not None else nn.ReLU()
        super().__init__(npoint, radii, nsample, channel_raising_nn, use_xyz, activation, kernels, **kwargs)
        self.mlps.extend(SharedRSConv(RSConvMapper(down_conv_nn, activation=activation, use_xyz=self.use_xyz), r)
                         for r in radii)
        self._mapper = self.mlps[-1]._mapper
