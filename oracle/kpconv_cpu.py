"""CPU ORACLE pieces for the partial-dense KPConv path -- test infrastructure, NOT product code (only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/).

Plain PyTorch fp32 restatements of the two feature ops (reference modules/KPConv/convolution_ops.py:19-107 `KPConv_ops`;
torch_geometric 1.7.2 `knn_interpolate`, called at core/spatial_ops/interpolate.py:69) and a helper that turns a
torch_points3d_amd KPConv model into its CPU mirror: the same modules with every device call replaced by an oracle piece
(radius search / kNN -> oracle/tpk_ref_cpu.c, GridSampling3D -> oracle/voxel_ref.py, KPConv_ops / knn_interpolate ->
the functions below).  `torch_kpconv` is pinned by tests/golden/kpconv_ops.npz (the reference's own KPConv_ops).
"""
import contextlib
import copy

import torch

from . import tpk_ref, voxel_ref


def torch_kpconv(query, support, nbr, feats, kpts, W, extent, influence, aggregation):
    """KPConv_ops in plain torch ops (fp32): shadow index -1 -> far point with zero feature."""
    M = support.shape[0]
    sup = torch.cat([support, torch.full_like(support[:1], 1e6)], 0)
    fx = torch.cat([feats, torch.zeros_like(feats[:1])], 0)
    idx = torch.where(nbr < 0, torch.full_like(nbr, M), nbr)
    rel = sup[idx] - query[:, None, :]                           # (Nq, Mn, 3)
    d2 = ((rel[:, :, None, :] - kpts[None, None]) ** 2).sum(-1)  # (Nq, Mn, KP)
    if influence == "constant":
        w = torch.ones_like(d2)
    elif influence == "linear":
        w = (1 - d2.sqrt() / extent).clamp(min=0)
    else:
        w = torch.exp(-d2 / (2 * (extent * 0.3) ** 2 + 1e-9))
    if aggregation == "closest":
        w = w * torch.nn.functional.one_hot(d2.argmin(-1), kpts.shape[0])
    wf = torch.einsum("qnk,qnc->qkc", w, fx[idx])
    return torch.einsum("qkc,kco->qo", wf, W)


def torch_knn_interpolate(x, idx, d2):
    """torch_geometric's knn_interpolate on given neighbours, literal (scatter_add in edge order)."""
    Nq, k = idx.shape
    y_idx = torch.arange(Nq).repeat_interleave(k)
    x_idx = idx.reshape(-1)
    keep = x_idx >= 0
    w = 1.0 / torch.clamp(d2.reshape(-1, 1).to(x.dtype), min=1e-16)
    y_idx, x_idx, w = y_idx[keep], x_idx[keep], w[keep]
    num = torch.zeros(Nq, x.shape[1], dtype=x.dtype).index_add_(0, y_idx, x[x_idx] * w)
    den = torch.zeros(Nq, 1, dtype=x.dtype).index_add_(0, y_idx, w)
    return num / den


class CpuSampler(object):
    """GridSampling3D(mode='mean') over pos / batch / x through the numpy restatement."""

    def __init__(self, size):
        self.size = size
        self._grid_size = size

    def __call__(self, data):
        x = getattr(data, "x", None)
        dt = data.pos.dtype  # a float64 evaluation samples the SAME fp32 cloud (positions are fp32 values on both sides)
        out = voxel_ref.grid_sampling_mean(data.pos.float().numpy(), self.size, batch=data.batch.numpy(),
                                           x=None if x is None else x.detach().float().numpy())
        data.pos = torch.from_numpy(out["pos"]).to(dt)
        data.batch = torch.from_numpy(out["batch"])
        if x is not None:
            data.x = torch.from_numpy(out["x"]).to(dt)
        data.grid_size = torch.tensor([self.size])
        return data


class CpuInterp(object):
    """KNNInterpolate(k) on the brute-force kNN."""

    def __init__(self, k):
        self.k = k

    def __call__(self, query, support, precomputed=None, skip=None):
        idx, d2 = tpk_ref.knn(self.k, query.pos.float(), support.pos.float(), query.batch, support.batch)
        if query.x.dtype == torch.float64:  # the float64 evaluation: same neighbours, distances in double
            nb = query.pos[idx.clamp(min=0)]
            d2 = ((nb - support.pos[:, None, :]) ** 2).sum(-1)
        y = torch_knn_interpolate(query.x, idx, d2)
        return y if skip is None else torch.cat([y, skip], dim=1)


def cpu_mirror(model, double=False):
    """-> (CPU copy of a torch_points3d_amd KPConv model with oracle samplers / up-samplers, context manager that
    routes its radius searches and convolutions to the oracle while active).  double: the float64 evaluation of the same
    pass -- parameters and features in double, sampling and searches on the fp32 coordinates (hence the same clouds and
    tables): the yardstick "how far may a correct fp32 implementation be from the exact result"."""
    from torch_points3d_amd import kpconv as kpconv_mod
    from torch_points3d_amd import torchpoints as tp_mod
    from torch_points3d_amd.kpconv_blocks import SimpleBlock
    from torch_points3d_amd.partial_dense import FPModule_PD
    cpu = copy.deepcopy(model).cpu()
    if double:
        cpu = cpu.double()
    for m in cpu.modules():
        if isinstance(m, SimpleBlock) and m.sampler is not None:
            m.sampler = CpuSampler(m.sampler._grid_size)
        if isinstance(m, FPModule_PD):
            m.upsample_op = CpuInterp(m.upsample_op.k)

    @contextlib.contextmanager
    def routed():
        saved = (tp_mod.ball_query, kpconv_mod.KPConv_ops)
        search = tpk_ref.ball_query
        if double:
            search = lambda r, n, x, y, **kw: tpk_ref.ball_query(r, n, x.float(), y.float(), **kw)  # noqa: E731
        tp_mod.ball_query, kpconv_mod.KPConv_ops = search, torch_kpconv
        try:
            yield
        finally:
            tp_mod.ball_query, kpconv_mod.KPConv_ops = saved

    return cpu, routed
