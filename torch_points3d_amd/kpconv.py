"""KPConv rigid kernel-point convolution on MI355X.

Mirrors `KPConv_ops` (torch_points3d/modules/KPConv/convolution_ops.py:19-107) and `KPConvLayer`
(modules/KPConv/kernels.py:20-104): same arguments, same shadow-neighbour convention (-1 -> zero feature), same
influence / aggregation modes, same parameter names (`K_points`, `weight`).  Stage 1 (kernel-point weighted
neighbourhood features) and its backward are HIP kernels (csrc/kpconv.hip); stage 2 is the single
(Nq, KP*Cin) x (KP*Cin, Cout) GEMM the reference's permute/matmul/sum amounts to; the kernel-weight gradient runs
on the split-K MFMA kernel (csrc/gemm_tn.hip).  Differentiable wrt `features` and `K_values` (what the reference
trains); positions and kernel points carry no gradient (kernels.py:57-59 sets requires_grad=False on K_points).
"""
import torch
import torch.nn as nn

from . import _lib
from .fused import gemm_tn

_INFLUENCE = {"constant": 0, "linear": 1, "gaussian": 2}


class _KPConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, K_values, query, support, nbr, kp, extent, influence, closest):
        dev = query.device
        x = features.detach().float().contiguous()
        W = K_values.detach().float().contiguous()
        Nq, Mn = nbr.shape
        M, Cin = x.shape
        KP = kp.shape[0]
        wf = torch.empty((Nq, KP * Cin), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.call("tp3d_kpconv_weighted_f32", _lib.ptr(query), _lib.ptr(support), _lib.ptr(nbr), _lib.ptr(x),
                      _lib.ptr(kp), Nq, M, Mn, Cin, KP, float(extent), influence, closest, _lib.ptr(wf),
                      _lib.stream_ptr(dev))
        out = torch.mm(wf, W.reshape(KP * Cin, -1))  # the dense contraction: a plain library GEMM
        ctx.save_for_backward(query, support, nbr, kp, W, wf)
        ctx.cfg = (float(extent), influence, closest, M, Cin, KP, tuple(K_values.shape))
        return out

    @staticmethod
    def backward(ctx, d_out):
        query, support, nbr, kp, W, wf = ctx.saved_tensors
        extent, influence, closest, M, Cin, KP, wshape = ctx.cfg
        dev = d_out.device
        d_out = d_out.float().contiguous()
        Nq, Mn = nbr.shape
        dW = gemm_tn(wf, d_out).reshape(wshape) if ctx.needs_input_grad[1] else None
        dx = None
        if ctx.needs_input_grad[0]:
            d_wf = torch.mm(d_out, W.reshape(KP * Cin, -1).t())  # (Nq, KP*Cin)
            dx = torch.empty((M, Cin), dtype=torch.float32, device=dev)
            nbytes = _lib.load().tp3d_kpconv_grad_workspace_bytes(M, Nq * Mn, Cin)
            ws = _lib.workspace("kpconv_bwd", nbytes, dev)
            with _lib.on_device(dev):
                inv, inv_bytes, ready, token = _lib.neighbour_inverse(nbr, M, dev)
                _lib.call("tp3d_kpconv_bwd_features_f32", _lib.ptr(query), _lib.ptr(support), _lib.ptr(nbr), _lib.ptr(kp),
                          _lib.ptr(d_wf), Nq, M, Mn, Cin, KP, extent, influence, closest, _lib.ptr(dx), _lib.ptr(inv),
                          inv_bytes, ready, _lib.ptr(ws), nbytes, _lib.stream_ptr(dev))
                _lib.inverse_built(token, dev)
        return dx, dW, None, None, None, None, None, None, None


def KPConv_ops(query_points, support_points, neighbors_indices, features, K_points, K_values, KP_extent,
               KP_influence, aggregation_mode):
    if KP_influence not in _INFLUENCE:
        raise ValueError("Unknown influence function type (config.KP_influence)")
    if aggregation_mode not in ("sum", "closest"):
        raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
    for t in (query_points, support_points, neighbors_indices, features, K_points, K_values):
        if t.device.type != "cuda":
            raise RuntimeError("torch_points3d_amd runs on MI355X only: got a %s tensor (no CPU fallback is provided)"
                               % t.device.type)
    q = query_points.detach().float().contiguous()
    s = support_points.detach().float().contiguous()
    nbr = neighbors_indices.long().contiguous()
    kp = K_points.detach().float().contiguous()
    return _KPConv.apply(features, K_values, q, s, nbr, kp, float(KP_extent), _INFLUENCE[KP_influence],
                         int(aggregation_mode == "closest"))


def default_kernel_points(num_points=15, iterations=400):
    """A kernel-point disposition in unit scale: one point at the centre, the others spread over the unit sphere by
    electrostatic repulsion from a Fibonacci lattice (deterministic).

    The reference loads a pre-optimised disposition from a data file (modules/KPConv/kernels/dispositions/*.ply via
    kernel_utils.load_kernels, kernels.py:51-56) and applies a random rotation; the file is not shipped with this
    build.  Any well-spread disposition is a valid initialisation, and a reference checkpoint's `K_points` replaces
    it on load_state_dict."""
    import numpy as np
    n = num_points - 1
    i = np.arange(n) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / n)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * i
    p = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)
    for _ in range(iterations):
        d = p[:, None, :] - p[None, :, :]
        r2 = (d * d).sum(-1) + np.eye(n)
        f = (d / r2[..., None] ** 1.5).sum(1)
        p = p + 0.05 * f
        p /= np.linalg.norm(p, axis=1, keepdims=True)
    return torch.from_numpy(np.concatenate([np.zeros((1, 3)), p], axis=0).astype(np.float32))


class KPConvLayer(nn.Module):
    """Kernel-point convolution layer with the reference's parameters (`K_points` frozen, `weight` (KP, Cin, Cout)
    xavier-normal) and forward signature (modules/KPConv/kernels.py:20-104).  The kernel-point disposition file of the
    reference is not shipped here: pass `K_points` (KP, 3), e.g. taken from a reference checkpoint or generated by
    the reference's `load_kernels`."""

    _INFLUENCE_TO_RADIUS = 1.5

    def __init__(self, num_inputs, num_outputs, point_influence, K_points, KP_influence="linear",
                 aggregation_mode="sum", add_one=False, **kwargs):
        # **kwargs: n_kernel_points / fixed / dimension of the reference's YAML are implied by K_points here
        super().__init__()
        self.kernel_radius = self._INFLUENCE_TO_RADIUS * point_influence
        self.point_influence = point_influence
        self.add_one = add_one
        self.num_inputs = num_inputs + int(add_one)
        self.num_outputs = num_outputs
        self.KP_influence = KP_influence
        self.aggregation_mode = aggregation_mode
        K_points = torch.as_tensor(K_points, dtype=torch.float32)
        self.n_kernel_points = K_points.shape[0]
        self.K_points = nn.Parameter(K_points.clone(), requires_grad=False)
        w = torch.empty([self.n_kernel_points, self.num_inputs, num_outputs], dtype=torch.float32)
        nn.init.xavier_normal_(w)
        self.weight = nn.Parameter(w)

    def forward(self, query_points, support_points, neighbors, x):
        if self.add_one:
            ones = torch.ones(support_points.shape[0], 1, dtype=torch.float32, device=support_points.device)
            x = ones if x is None else torch.cat([ones, x.float()], dim=-1)
        return KPConv_ops(query_points, support_points, neighbors, x, self.K_points, self.weight, self.point_influence,
                          self.KP_influence, self.aggregation_mode)
