"""Relation-Shape convolution (dense format) on the HIP sampling / radius-search / grouping kernels.

Mirrors (same constructor arguments, attribute names -- hence state_dict keys -- and tensor contracts):
  * `RSConvMapper`          torch_points3d/modules/RSConv/dense.py:18-66
  * `SharedRSConv`          torch_points3d/modules/RSConv/dense.py:69-103
  * `RSConvSharedMSGDown`   torch_points3d/modules/RSConv/dense.py:106-190
  * `RSConvMSGDown`         torch_points3d/modules/RSConv/dense.py:398-476
The reference reaches torch_points_kernels three times per scale here (furthest_point_sample through the sampler,
ball_query through the neighbour finder, grouping_operation twice in `_prepare_features`); the relation message and
the channel-raising MLPs are plain torch layers in both.  Channel order of the grouped tensor is
[absolute xyz (3), centred xyz (3), features (C)] -- not PointNet++'s [centred xyz, features].
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import torchpoints as _hip_kernels
from .dense import BaseDenseConvolutionDown, DenseFPSSampler, DenseRadiusNeighbourFinder, MLP2D


class RSConvMapper(nn.Module):
    """Maps the 10-channel geometric relation to feature space and modulates the (raised) features with it."""

    def __init__(self, down_conv_nn, use_xyz, bn=True, activation=None, *args, **kwargs):
        super().__init__()
        activation = activation if activation is not None else nn.LeakyReLU(negative_slope=0.01)
        self._down_conv_nn = down_conv_nn
        self._use_xyz = use_xyz
        self.nn = nn.ModuleDict()
        if len(self._down_conv_nn) == 2:  # first layer: the raw features are raised to f_out channels as well
            self._first_layer = True
            f_in, f_intermediate, f_out = self._down_conv_nn[0]
            self.nn["features_nn"] = MLP2D(self._down_conv_nn[1], bn=bn, bias=False)
        else:
            self._first_layer = False
            f_in, f_intermediate, f_out = self._down_conv_nn
        self.nn["mlp_msg"] = MLP2D([f_in, f_intermediate, f_out], bn=bn, bias=False)
        self.nn["norm"] = nn.Sequential(nn.BatchNorm2d(f_out), activation)
        self._f_out = f_out

    @property
    def f_out(self):
        return self._f_out

    def forward(self, features, msg):
        """features (B, C, np, ns); msg (B, 10, np, ns) = [distance, centroid xyz, neighbour xyz, delta xyz]."""
        msg = self.nn["mlp_msg"](msg)
        if self._first_layer:
            features = self.nn["features_nn"](features)
        return self.nn["norm"](torch.mul(features, msg))


class SharedRSConv(nn.Module):
    """(B, 3 + 3 + C, np, ns) grouped tensor + centroids (B, 3, np, 1) -> modulated features (B, C_out, np, ns)."""

    def __init__(self, mapper, radius):
        super().__init__()
        self._mapper = mapper
        self._radius = radius

    def forward(self, aggr_features, centroids):
        abs_coord = aggr_features[:, :3]
        delta_x = aggr_features[:, 3:6]
        features = aggr_features[:, 3:]  # centred xyz stays part of the features, as in the reference
        nsample = abs_coord.shape[-1]
        coord_xi = centroids.repeat(1, 1, 1, nsample)
        distance = torch.norm(delta_x, p=2, dim=1).unsqueeze(1)
        h_xi_xj = torch.cat((distance, coord_xi, abs_coord, delta_x), dim=1)
        return self._mapper(features, h_xi_xj)

    def __repr__(self):
        return "{}(radius={})".format(self.__class__.__name__, self._radius)


class _RSConvDownBase(BaseDenseConvolutionDown):
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

    def _prepare_features(self, x, pos, new_pos, idx):
        grouped_pos_absolute = self._tp.grouping_operation(pos.transpose(1, 2).contiguous(), idx)  # (B,3,np,ns)
        centroids = new_pos.transpose(1, 2).unsqueeze(-1)
        grouped_pos_normalized = grouped_pos_absolute - centroids
        if x is not None:
            grouped_features = self._tp.grouping_operation(x, idx)
            if self.use_xyz:
                new_features = torch.cat([grouped_pos_absolute, grouped_pos_normalized, grouped_features], dim=1)
            else:
                new_features = grouped_features
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = torch.cat([grouped_pos_absolute, grouped_pos_normalized], dim=1)
        return new_features, centroids

    def conv(self, x, pos, new_pos, radius_idx, scale_idx):
        assert scale_idx < len(self.mlps)
        aggr_features, centroids = self._prepare_features(x, pos, new_pos, radius_idx)
        new_features = self.mlps[scale_idx](aggr_features, centroids)  # (B, C_out, np, ns)
        new_features = F.max_pool2d(new_features, kernel_size=[1, new_features.size(3)])
        return self.mlp_out(new_features.squeeze(-1))  # (B, C_raised, np)


class RSConvSharedMSGDown(_RSConvDownBase):
    """One mapper shared by every scale."""

    def __init__(self, npoint=None, radii=None, nsample=None, down_conv_nn=None, channel_raising_nn=None, bn=True,
                 use_xyz=True, activation=None, kernels=None, **kwargs):
        activation = activation if activation is not None else nn.ReLU()
        super().__init__(npoint, radii, nsample, channel_raising_nn, use_xyz, activation, kernels, **kwargs)
        self._mapper = RSConvMapper(down_conv_nn, activation=activation, use_xyz=self.use_xyz)
        for i in range(len(radii)):
            self.mlps.append(SharedRSConv(self._mapper, radii[i]))


class RSConvMSGDown(_RSConvDownBase):
    """One mapper per scale (`_mapper` keeps the last one, as the reference does)."""

    def __init__(self, npoint=None, radii=None, nsample=None, down_conv_nn=None, channel_raising_nn=None, bn=True,
                 bias=True, use_xyz=True, activation=None, kernels=None, **kwargs):
        activation = activation if activation is not None else nn.ReLU()
        super().__init__(npoint, radii, nsample, channel_raising_nn, use_xyz, activation, kernels, **kwargs)
        mapper = None
        for i in range(len(radii)):
            mapper = RSConvMapper(down_conv_nn, activation=activation, use_xyz=self.use_xyz)
            self.mlps.append(SharedRSConv(mapper, radii[i]))
        self._mapper = mapper
