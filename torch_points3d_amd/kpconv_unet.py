"""KPConv U-Net backbone (BASELINE config 4): the model `KPConv(architecture="unet", ...)` of the reference's
applications API, assembled from the partial-dense blocks of this package.

Mirrors torch_points3d/applications/kpconv.py:22-181 (`KPConv`, `BaseKPConv`, `KPConvUnet`) over the bundled
architecture applications/conf/kpconv/unet_4.yaml and the flat U-Net runner
models/base_architectures/unet.py:312-530 (`UnwrappedUnetBasedModel`: down_modules / inner_modules / up_modules, skip
stack popped symmetrically).  Module and parameter names match, so a reference state_dict loads.

The architecture table is generated from (num_layers, in_feat, in_grid_size) by the same rule the YAML spells out
for four layers: level 0 = SimpleBlock(FEAT+1 -> f) + ResnetBBlock(f -> 2f); level i = strided ResnetBBlock
(2^i f -> 2^i f, grid doubles) + ResnetBBlock(2^i f -> 2^(i+1) f); decoder = FPModule_PD(up_k = 1) with
[48f, 8f], [16f, 4f], [8f, 2f], [4f, f] for four layers.
"""
import torch.nn as nn

from . import fused as _fused
from .kpconv_blocks import KPDualBlock
from .partial_dense import MLP, FPModule_PD


def unet_config(num_layers=4, input_nc=3, in_feat=64, in_grid_size=0.02, bn_momentum=0.2, max_neighbors=25):
    """The resolved option lists of applications/conf/kpconv/unet_<num_layers>.yaml (only unet_4 ships in the reference)."""
    f = in_feat
    down = []
    for i in range(num_layers + 1):
        w = f * 2 ** i
        g = in_grid_size * 2 ** i
        if i == 0:
            down.append(dict(down_conv_nn=[[input_nc + 1, f], [f, 2 * f]], grid_size=[g, g], prev_grid_size=[g, g],
                             block_names=["SimpleBlock", "ResnetBBlock"], has_bottleneck=[False, True],
                             max_num_neighbors=[max_neighbors, max_neighbors]))
        else:
            down.append(dict(down_conv_nn=[[w, w], [w, 2 * w]], grid_size=[g, g], prev_grid_size=[g / 2, g],
                             block_names=["ResnetBBlock", "ResnetBBlock"], has_bottleneck=[True, True],
                             max_num_neighbors=[max_neighbors, max_neighbors]))
    up = []
    for j in range(num_layers):
        lvl = num_layers - j  # coarse level feeding this decoder stage
        coarse = f * 2 ** (lvl + 1) if j == 0 else f * 2 ** lvl
        skip = f * 2 ** lvl
        up.append(dict(up_k=1, up_conv_nn=[coarse + skip, f * 2 ** (lvl - 1)], bn_momentum=bn_momentum))
    return dict(down_conv=down, up_conv=up)


class KPConvUnet(nn.Module):
    """down_modules (KPDualBlock per level) -> Identity inner -> up_modules (FPModule_PD), symmetric skips."""

    def __init__(self, config, output_nc=None, kernel_points=None, fused=True):
        super().__init__()
        self.fused = fused
        self.down_modules = nn.ModuleList()
        self.inner_modules = nn.ModuleList([nn.Identity()])
        self.up_modules = nn.ModuleList()
        for i, opt in enumerate(config["down_conv"]):
            self.down_modules.append(KPDualBlock(kernel_points=kernel_points, fused=fused, **opt))
        for opt in config["up_conv"]:
            self.up_modules.append(FPModule_PD(fused=fused, **opt))
        default_output_nc = config["up_conv"][-1]["up_conv_nn"][-1]
        self._output_nc = default_output_nc
        self._has_mlp_head = False
        if output_nc is not None:
            self._has_mlp_head = True
            self._output_nc = output_nc
            self.mlp = MLP([default_output_nc, output_nc], activation=nn.LeakyReLU(0.2), bias=False)

    @property
    def has_mlp_head(self):
        return self._has_mlp_head

    @property
    def output_nc(self):
        return self._output_nc

    def get_spatial_ops(self):
        """{"sampler", "neighbour_finder", "upsample_op"}: one entry per block / decoder stage, in forward order
        (reference models/base_architectures/unet.py:315-334; the input of `MultiScaleTransform`)."""
        ops = {"sampler": [], "neighbour_finder": [], "upsample_op": []}
        for down in self.down_modules:
            for block in down.blocks:
                ops["sampler"].append(block.sampler)
                ops["neighbour_finder"].append(block.neighbour_finder)
        for up in self.up_modules:
            ops["upsample_op"].append(up.upsample_op)
        return ops

    def forward(self, data, precomputed_down=None, precomputed_up=None):
        """data: pos (N,3), x (N, input_nc + 1), batch (N) sorted -> data with x (N, output_nc) at the input resolution.
        Data that went through `MultiScaleTransform` (attributes `multiscale` / `upsample`) runs on those tables
        (reference applications/kpconv.py:99-111)."""
        if precomputed_down is None and getattr(data, "multiscale", None) is not None:
            precomputed_down, precomputed_up = data.multiscale, getattr(data, "upsample", None)
            data = data.shallow_copy() if hasattr(data, "shallow_copy") else data
            data.multiscale = data.upsample = None
        stack_down = []
        for i in range(len(self.down_modules) - 1):
            data = self.down_modules[i](data, precomputed=precomputed_down)
            stack_down.append(data)
        data = self.down_modules[-1](data, precomputed=precomputed_down)
        for i in range(len(self.up_modules)):
            data = self.up_modules[i]((data, stack_down.pop()), precomputed=precomputed_up)
        if self.has_mlp_head:
            data.x = _fused.rows_mlp(self.mlp, data.x) if self.fused else self.mlp(data.x)
        return data


def KPConv(architecture="unet", input_nc=None, num_layers=4, config=None, in_feat=64, in_grid_size=0.02, output_nc=None,
           kernel_points=None, fused=True, **kwargs):
    """Factory with the reference's signature (applications/kpconv.py:22-49); only the U-Net is assembled here."""
    if not architecture:
        raise ValueError()
    if architecture.lower() != "unet":
        raise NotImplementedError("only architecture='unet' is built (the encoder is its down_modules half)")
    cfg = config if config is not None else unet_config(num_layers, input_nc, in_feat, in_grid_size,
                                                        kwargs.get("bn_momentum", 0.2), kwargs.get("max_neighbors", 25))
    return KPConvUnet(cfg, output_nc=output_nc, kernel_points=kernel_points, fused=fused)
