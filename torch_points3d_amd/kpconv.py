"""KPConv rigid kernel-point convolution on MI355X.

Mirrors `KPConv_ops` (torch_points3d/modules/KPConv/convolution_ops.py:19-107) and `KPConvLayer.forward`
(modules/KPConv/kernels.py:74-94): same arguments, same shadow-neighbour convention (-1 -> zero feature), same
influence / aggregation modes.  Stage 1 (kernel-point weighted neighbourhood features) is a HIP kernel
(csrc/kpconv.hip); stage 2 is the single (Nq, KP*Cin) x (KP*Cin, Cout) GEMM the reference's permute/matmul/sum
amounts to.  Forward only this round (BASELINE config 4 is a forward benchmark): tensors that require grad raise.
"""
import torch

from . import _lib

_INFLUENCE = {"constant": 0, "linear": 1, "gaussian": 2}


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
    if torch.is_grad_enabled() and (features.requires_grad or K_values.requires_grad):
        raise NotImplementedError("KPConv_ops backward is not built yet (forward-only this round); "
                                  "wrap the call in torch.no_grad()")
    dev = query_points.device
    q = query_points.detach().float().contiguous()
    s = support_points.detach().float().contiguous()
    nbr = neighbors_indices.long().contiguous()
    x = features.detach().float().contiguous()
    kp = K_points.detach().float().contiguous()
    W = K_values.detach().float().contiguous()
    Nq, Mn = nbr.shape
    M, Cin = x.shape
    KP = kp.shape[0]
    wf = torch.empty((Nq, KP * Cin), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        _lib.call("tp3d_kpconv_weighted_f32", _lib.ptr(q), _lib.ptr(s), _lib.ptr(nbr), _lib.ptr(x), _lib.ptr(kp), Nq,
                  M, Mn, Cin, KP, float(KP_extent), _INFLUENCE[KP_influence], int(aggregation_mode == "closest"),
                  _lib.ptr(wf), _lib.stream_ptr(dev))
    return torch.mm(wf, W.reshape(KP * Cin, -1))  # the dense contraction: a plain library GEMM
