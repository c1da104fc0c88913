"""RandLA-Net down-convolution on the HIP kNN (BASELINE config 5).

Mirrors torch_points3d/modules/RandLANet/modules.py:9-67 (`RandlaKernel`, `RandlaConv`),
core/base_conv/message_passing.py:35-58 (`BaseConvolutionDown.forward`) and core/spatial_ops/sampling.py:103-112
(`RandomSampler`): same constructor arguments and attribute names (point_pos_nn / attention_nn / global_nn under
`_conv`), same message (relative-position encoding [pos_i, pos_j, pos_i - pos_j, |.|] -> MLP, concatenation with the
neighbour feature, softmax attention over channels, sum over the k neighbours, global MLP).

The reference runs this as a torch_geometric MessagePassing over an edge list built by torch_cluster's `knn`; here the
neighbour table (Nq, k) comes from libtp3d_hip.so's exact grid kNN and, because every query owns exactly k consecutive
edges, the "add" aggregation is a sum over a (Nq, k, C) view -- no scatter.  Parity: unpinned (torch_cluster absent;
the reference's own model test skips randlanet, test/test_models.py:116-125).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import fused as _fused
from . import torchpoints as _tp
from .kpconv_blocks import PDData
from .partial_dense import MLP, _KnnInterpolate


def relative_position_rows(pos_q, pos_s, nbr):
    """(Nq*k, 12) rows [pos_i, pos_j, pos_i - pos_j, |pos_i - pos_j|, 0, 0] of the edges of a fixed-k table
    (modules.py:36-41); positions carry no gradient."""
    Nq, k = nbr.shape
    dev = pos_q.device
    pq, ps = _tp._f32(pos_q), _tp._f32(pos_s)
    nb = _tp._i64(nbr)
    out = torch.empty((Nq * k, 12), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        _lib.call("tp3d_randla_relpos_f32", _lib.ptr(pq), _lib.ptr(ps), _lib.ptr(nb), Nq, k, ps.shape[0], _lib.ptr(out),
                  _lib.stream_ptr(dev))
    return out


class _AttentivePool(torch.autograd.Function):
    """out (Nq, C) = sum over the k edges of a query of softmax_c(g[e]) * f[e, :C]   (modules.py:46-52, aggr="add")."""

    @staticmethod
    def forward(ctx, g, f, nbr, C):
        g, f = g.contiguous(), f.contiguous()
        Nq, k = nbr.shape
        dev = g.device
        if g.shape[0] != Nq * k or f.shape[0] != Nq * k or g.shape[1] < C or f.shape[1] < C:
            raise ValueError("attentive_pool: g %s / f %s do not match a (%d, %d) neighbour table with %d channels"
                             % (tuple(g.shape), tuple(f.shape), Nq, k, C))
        out = torch.empty((Nq, C), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.call("tp3d_attn_pool_fwd_f32", _lib.ptr(g), _lib.ptr(f), _lib.ptr(nbr), Nq, k, C, g.shape[1],
                      f.shape[1], _lib.ptr(out), _lib.stream_ptr(dev))
        ctx.save_for_backward(g, f, nbr)
        ctx.C = C
        return out

    @staticmethod
    def backward(ctx, dout):
        g, f, nbr = ctx.saved_tensors
        Nq, k = nbr.shape
        dev = g.device
        dout = dout.float().contiguous()
        dg = torch.empty_like(g) if g.shape[1] == ctx.C else torch.zeros_like(g)
        df = torch.empty_like(f)
        with _lib.on_device(dev):
            _lib.call("tp3d_attn_pool_bwd_f32", _lib.ptr(g), _lib.ptr(f), _lib.ptr(dout), _lib.ptr(nbr), Nq, k, ctx.C,
                      g.shape[1], f.shape[1], _lib.ptr(dg), _lib.ptr(df), _lib.stream_ptr(dev))
        return dg, df, None, None


def attentive_pool(g, f, nbr, C):
    return _AttentivePool.apply(g, f, nbr, C)


class RandomSampler(object):
    """floor(N * ratio) (or num_to_sample) indices drawn uniformly WITH replacement (sampling.py:109-111)."""

    def __init__(self, ratio=None, num_to_sample=None):
        if num_to_sample is not None:
            if ratio is not None:
                raise ValueError("Can only specify ratio or num_to_sample or subsampling_param, not several !")
            self._num_to_sample = num_to_sample
        elif ratio is not None:
            self._ratio = ratio
        else:
            raise Exception('At least ["ratio, num_to_sample, subsampling_param"] should be defined')

    def _get_num_to_sample(self, n):
        return self._num_to_sample if hasattr(self, "_num_to_sample") else math.floor(n * self._ratio)

    def sample(self, pos, batch=None, **kwargs):
        if len(pos.shape) != 2:
            raise ValueError(" This class is for sparse data and expects the pos tensor to be of dimension 2")
        return torch.randint(0, pos.shape[0], (self._get_num_to_sample(pos.shape[0]),), device=pos.device)

    def __call__(self, pos, x=None, batch=None):
        return self.sample(pos, batch=batch, x=x)


class RandlaKernel(nn.Module):
    """Local spatial encoding + attentive pooling over a fixed-k neighbour table."""

    def __init__(self, point_pos_nn=None, attention_nn=None, global_nn=None, *args, **kwargs):
        super().__init__()
        self.point_pos_nn = MLP(point_pos_nn)
        self.attention_nn = MLP(attention_nn)
        self.global_nn = MLP(global_nn)
        self.fused = kwargs.get("fused", True)

    def forward(self, x, pos, nbr):
        """x (M,C) or None, pos = (query positions (Nq,3), support positions (M,3)), nbr (Nq,k) rows of the support"""
        pos_q, pos_s = pos
        Nq, k = nbr.shape
        if pos_s.is_cuda and self.fused:
            return self._forward_fused(x, pos_q, pos_s, nbr)
        j = nbr.reshape(-1)
        pos_i = pos_q.repeat_interleave(k, dim=0)
        pos_j = pos_s[j]
        x_j = pos_j if x is None else x[j]
        vij = pos_i - pos_j
        dij = torch.norm(vij, dim=1).unsqueeze(1)
        rij = _fused.rows_mlp(self.point_pos_nn, torch.cat([pos_i, pos_j, vij, dij], dim=1))
        fij_hat = torch.cat([x_j, rij], dim=1)
        s_ij = F.softmax(_fused.rows_mlp(self.attention_nn, fij_hat), -1)
        msg = s_ij * fij_hat
        return _fused.rows_mlp(self.global_nn, msg.reshape(Nq, k, -1).sum(dim=1))

    def _forward_fused(self, x, pos_q, pos_s, nbr):
        """Same arithmetic; the edge-wise pieces between the MLPs are HIP row kernels (csrc/randla.hip) and the
        neighbour-feature gather + concatenation is the k = 1 case of the interpolation kernel (weight exactly 1),
        whose backward is the atomic-free inverse-index gather."""
        Nq, k = nbr.shape
        nbr = _tp._i64(nbr)
        rij = _fused.rows_mlp(self.point_pos_nn, relative_position_rows(pos_q, pos_s, nbr))
        xs = _tp._f32(pos_s) if x is None else x
        C = xs.shape[1] + rij.shape[1]
        edges = nbr.reshape(-1, 1)
        ones = torch.ones((edges.shape[0], 1), dtype=torch.float32, device=nbr.device)
        fij_hat = _KnnInterpolate.apply(xs, rij, edges, ones, (C + 3) // 4 * 4)  # (Nq*k, pad4(C)) = [x_j | rij | 0]
        g_fij = _fused.rows_mlp(self.attention_nn, fij_hat)
        return _fused.rows_mlp(self.global_nn, attentive_pool(g_fij, fij_hat, nbr, C))


class RandlaConv(nn.Module):
    def __init__(self, ratio=None, k=None, *args, **kwargs):
        super().__init__()
        self.sampler = RandomSampler(ratio)
        self.k = k
        self.edge_list = bool(kwargs.get("edge_list", False))
        if kwargs.get("index") == 0 and kwargs.get("nb_feature") is not None:
            kwargs["point_pos_nn"][-1] = kwargs.get("nb_feature")
            kwargs["attention_nn"][0] = kwargs["attention_nn"][-1] = kwargs.get("nb_feature") * 2
            kwargs["down_conv_nn"][0] = kwargs.get("nb_feature") * 2
        self._conv = RandlaKernel(point_pos_nn=kwargs["point_pos_nn"], attention_nn=kwargs["attention_nn"],
                                  global_nn=kwargs["down_conv_nn"], fused=kwargs.get("fused", True))

    def forward(self, data, **kwargs):
        x, pos, batch = data.x, data.pos, data.batch
        idx = self.sampler(pos, batch=batch)
        q_pos, q_batch = pos[idx], batch[idx]
        nbr, _ = _tp.knn(self.k, pos, q_pos, batch, q_batch)  # exact kNN of every sampled point in its own cloud
        out = data.shallow_copy() if hasattr(data, "shallow_copy") else PDData(**vars(data))
        out.idx = idx
        out.neighbors = nbr
        if self.edge_list:
            # the reference's edge list (message_passing.py:49-52: edge_index = stack([col, row])): query-major, each
            # query's neighbours closest first -- row 0 = support index, row 1 = query index
            row = torch.arange(nbr.shape[0], device=nbr.device).repeat_interleave(self.k)
            col = nbr.reshape(-1)
            keep = col >= 0
            out.edge_index = torch.stack([col[keep], row[keep]], dim=0)
        out.x = self._conv(x, (q_pos, pos), nbr)
        out.pos = q_pos
        out.batch = q_batch
        return out


class DilatedResidualBlock(nn.Module):
    """Two RandlaConv in sequence inside a residual frame (modules/RandLANet/modules.py:70-102 on
    core/base_conv/message_passing.py:212-255 `BaseResnetBlock`): the shortcut rows are gathered with the LAST
    convolution's sample indices, resized by `shortcut_feature_resize_nn` and added to the up-sampled convolution output.
    As in the reference, `features_downsample_nn` is evaluated on the input (its BatchNorm statistics move in training
    mode) but its result is not what the convolutions consume -- they read `data.x` -- and `activation` is unused."""

    def __init__(self, indim, outdim, ratio1, ratio2, point_pos_nn1, point_pos_nn2, attention_nn1, attention_nn2,
                 global_nn1, global_nn2, *args, **kwargs):
        super().__init__()
        if kwargs.get("index") == 0 and kwargs.get("nb_feature") is not None:
            indim = kwargs.get("nb_feature")
        self.indim, self.outdim, self.convdim = indim, outdim, outdim
        self.features_downsample_nn = MLP([indim, outdim // 4])
        self.features_upsample_nn = MLP([outdim, outdim])
        self.shortcut_feature_resize_nn = MLP([indim, outdim])
        self.activation = nn.ReLU()
        kw = dict(kwargs)
        self.conv1 = RandlaConv(ratio1, 16, point_pos_nn=point_pos_nn1, attention_nn=attention_nn1,
                                down_conv_nn=global_nn1, **kw)
        kw["nb_feature"] = None
        self.conv2 = RandlaConv(ratio2, 16, point_pos_nn=point_pos_nn2, attention_nn=attention_nn2,
                                down_conv_nn=global_nn2, **kw)

    def convs(self, data):
        return self.conv2(self.conv1(data))

    def forward(self, data, **kwargs):
        shortcut = data.x
        _fused.rows_mlp(self.features_downsample_nn, data.x)  # evaluated, not consumed (see the class note)
        out = self.convs(data)
        x = _fused.rows_mlp(self.features_upsample_nn, out.x)
        if out.idx is not None:
            shortcut = shortcut[out.idx]
        out.x = _fused.rows_mlp(self.shortcut_feature_resize_nn, shortcut) + x
        return out


class RandLANetRes(nn.Module):
    """conf/models/segmentation/randlanet.yaml `Randlanet_Res` down module: lists of two entries per argument"""

    def __init__(self, indim, outdim, ratio, point_pos_nn, attention_nn, down_conv_nn, *args, **kwargs):
        super().__init__()
        self._conv = DilatedResidualBlock(indim, outdim, ratio[0], ratio[1], point_pos_nn[0], point_pos_nn[1],
                                          attention_nn[0], attention_nn[1], down_conv_nn[0], down_conv_nn[1], *args, **kwargs)

    def forward(self, data):
        return self._conv(data)
