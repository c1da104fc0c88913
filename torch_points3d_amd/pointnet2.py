"""PointNet++ U-Net assembled from the dense blocks, reproducing the reference's caller for the hot path.

Mirrors torch_points3d/applications/pointnet2.py:22-191 (`PointNet2(architecture="unet", ...)` ->
`PointNet2Unet.forward`) and the argument unpacking of models/base_architectures/unet.py:400-487 for the
bundled configs torch_points3d/applications/conf/pointnet2/{unet_3_ss,unet_4_ss,unet_3_ms}.yaml:
module order, skip-stack order, attribute names (`down_modules`, `inner_modules`, `up_modules`, `mlp`) and
therefore state_dict keys are the reference's, so its checkpoints load unchanged.
"""
import torch
import torch.nn as nn

from . import fused as _fused
from .dense import Conv1D, Data, DenseFPModule, GlobalDenseBaseModule, PointNetMSGDown, Seq


def unet_config(name, feat):
    """The reference's bundled YAMLs with FEAT (and in_feat) already substituted."""
    if name == "unet_3_ss":  # conf/pointnet2/unet_3_ss.yaml:1-19
        return dict(
            npoint=[512, 128], radii=[[0.2], [0.4]], nsample=[[64], [64]],
            down_conv_nn=[[[feat + 3, 64, 64, 128]], [[128 + 3, 128, 128, 256]]],
            innermost=[256 + 3, 256, 512, 1024],
            up_conv_nn=[[1024 + 256, 256, 256], [256 + 128, 256, 128], [128 + feat, 128, 128, 128]],
            normalize_xyz=[False, False], save_sampling_id=[False, False])
    if name == "unet_4_ss":  # conf/pointnet2/unet_4_ss.yaml:1-25 (three down layers are built)
        f = 64
        return dict(
            npoint=[2048, 1024, 512], radii=[[0.2], [0.4], [0.8]], nsample=[[64], [32], [16]],
            down_conv_nn=[[[feat + 3, f, f, f * 2]], [[f * 2 + 3, f * 2, f * 2, f * 4]],
                          [[f * 4 + 3, f * 2, f * 2, f * 4]]],
            innermost=[f * 4 + 3, f * 8, f * 16],
            up_conv_nn=[[f * 16 + f * 4, f * 8, f * 8], [f * 8 + f * 4, f * 8, f * 8],
                        [f * 8 + f * 2, f * 4, f * 4], [f * 4 + feat, f * 2, f * 2]],
            normalize_xyz=[True, True, True], save_sampling_id=[True, False, False])
    if name == "unet_3_ms":  # conf/pointnet2/unet_3_ms.yaml
        return dict(
            npoint=[512, 128], radii=[[0.1, 0.2, 0.4], [0.4, 0.8]], nsample=[[32, 64, 128], [64, 128]],
            down_conv_nn=[[[feat + 3, 32, 32, 64], [feat + 3, 64, 64, 128], [feat + 3, 64, 96, 128]],
                          [[64 + 128 + 128 + 3, 128, 128, 256], [64 + 128 + 128 + 3, 128, 196, 256]]],
            innermost=[256 * 2 + 3, 256, 512, 1024],
            up_conv_nn=[[1024 + 256 * 2, 256, 256], [256 + 128 * 2 + 64, 256, 128], [128 + feat, 128, 128]],
            normalize_xyz=[False, False], save_sampling_id=[False, False])
    raise ValueError("unknown PointNet++ config %r" % name)


class PointNet2Unet(nn.Module):
    """Input -- D1 -- D2 -- I -- U1 -- U2 -- U3 -- (head), symmetric skips (applications/pointnet2.py:154-191)."""

    def __init__(self, input_nc, output_nc=None, config="unet_3_ss", kernels=None, activation=None, fused=True):
        """`activation` (default LeakyReLU(0.01), as the reference modules) is shared by every layer.
        `fused=False` keeps the reference's (B,C,np,ns) PyTorch graph around the HIP spatial kernels."""
        super().__init__()
        self.fused = fused
        cfg = unet_config(config, input_nc) if isinstance(config, str) else config
        self.config = cfg
        self._kernels_are_hip = kernels is None
        self.down_modules = nn.ModuleList()
        for i in range(len(cfg["down_conv_nn"])):
            self.down_modules.append(PointNetMSGDown(
                npoint=cfg["npoint"][i], radii=cfg["radii"][i], nsample=cfg["nsample"][i],
                down_conv_nn=cfg["down_conv_nn"][i], normalize_xyz=cfg["normalize_xyz"][i],
                save_sampling_id=cfg["save_sampling_id"][i], index=i, kernels=kernels, activation=activation,
                fused=fused))
        self.inner_modules = nn.ModuleList([GlobalDenseBaseModule(nn=cfg["innermost"], activation=activation,
                                                                  fused=fused and kernels is None)])
        self.up_modules = nn.ModuleList(
            DenseFPModule(up_conv_nn=c, index=i, kernels=kernels, activation=activation, fused=fused)
            for i, c in enumerate(cfg["up_conv_nn"]))
        self._output_nc = cfg["up_conv_nn"][-1][-1]
        self.has_mlp_head = output_nc is not None
        if self.has_mlp_head:
            # BN + LeakyReLU are applied on the logits too (applications/pointnet2.py:100-104)
            self.mlp = Seq().append(Conv1D(self._output_nc, output_nc, bn=True, bias=False, activation=activation))
            self._output_nc = output_nc

    @property
    def output_nc(self):
        return self._output_nc

    def precompute_geometry(self, pos, backward_tables=False):
        """Everything the forward pass derives from the positions alone -- per set-abstraction level the sampled indices
        / positions / neighbour tables, per feature-propagation stage the 3-NN interpolation table -- computed once for
        `pos` (B,N,3).  Pass it as forward(data, geometry=): the pass then contains no sampling and no search, so the
        geometry of the NEXT batch can be computed on a second stream while this batch trains (dp.PipelinedStep), the
        dense-format counterpart of the reference's MultiScaleTransform precompute
        (core/data_transform/transforms.py:579-654).  backward_tables: for a training pass -- also the inverted neighbour
        and interpolation tables its backward pass gathers through (not for the first level, whose input features are
        the data and take no gradient)."""
        levels, cur, positions = [], pos, [pos]
        for i, down in enumerate(self.down_modules):
            g = down.precompute(cur, backward_tables=backward_tables and i > 0)
            levels.append(g)
            cur = g.new_pos
            positions.append(cur)
        ups, below = [], None  # `below` = position set of the stage's input (None under the global module)
        for up in self.up_modules:
            skip_pos = positions.pop()
            ups.append(up.precompute(below, skip_pos, backward_tables=backward_tables))
            below = skip_pos
        return Data(down=levels, up=ups)

    def forward(self, data, geometry=None):
        """data.pos (B,N,3), data.x (B,N,C) or None -> Data(pos (B,N,3), x (B,output_nc,N))."""
        assert data.pos.dim() == 3
        x = None
        if data.x is not None:
            # the reference makes (B,C,N) contiguous (pointnet2.py:118-121); the fused modules consume the
            # channel-last storage directly, so there the transposed VIEW is handed on instead of a copy
            x = data.x.transpose(1, 2)
            if not (self.fused and self._kernels_are_hip and x.is_cuda):
                x = x.contiguous()
        cur = Data(pos=data.pos, x=x)
        stack_down = [cur]
        for i in range(len(self.down_modules)):
            cur = self.down_modules[i](cur, precomputed=None if geometry is None else geometry.down[i])
            stack_down.append(cur)
        cur = self.inner_modules[0](cur)
        sampling_ids = {}
        for d in stack_down:
            for k, v in d.__dict__.items():
                if k.startswith("sampling_id"):
                    sampling_ids[k] = v
        for i, up in enumerate(self.up_modules):
            cur = up((cur, stack_down.pop()), precomputed=None if geometry is None else geometry.up[i])
        for k, v in sampling_ids.items():
            setattr(cur, k, v)
        if self.has_mlp_head:
            cur.x = self._head(cur.x)
        return cur

    def _head(self, x):
        if self.fused and x.is_cuda and self._kernels_are_hip:
            parts = _fused.mlp_parts(self.mlp)
            if parts is not None:
                B, _, n = x.shape
                out = _fused.run_mlp(_fused._cl(x).reshape(B * n, -1), parts)
                return out.view(B, n, -1).transpose(1, 2)
        return self.mlp(x)


class UnetSkipConnectionBlock(nn.Module):
    """One level of the nested U-Net the reference's segmentation models are built from
    (models/base_architectures/unet.py:245-297): down -> submodule -> up with the level's input as skip; the innermost
    level runs the global module instead.  Attribute names (`down`, `submodule`, `up`, `inner`) and the construction
    order (which fixes the default initialisation under a seed) are the reference's, so its checkpoints load."""

    def __init__(self, make_up, make_down=None, make_inner=None, submodule=None):
        super().__init__()
        self.innermost = make_inner is not None
        if self.innermost:
            self.inner = make_inner()
            self.up = make_up()
        else:
            down, up = make_down(), make_up()
            self.down = down
            self.submodule = submodule
            self.up = up

    def forward(self, data):
        below = self.inner(data) if self.innermost else self.submodule(self.down(data))
        return self.up((below, data))


def segmentation_config(name, feat):
    """conf/models/segmentation/pointnet2.yaml resolved for FEAT."""
    if name == "pointnet2_charlesmsg":  # :95-130 -- the multi-scale network of the PointNet++ paper (part segmentation)
        return dict(unet_config("unet_3_ms", feat), mlp_cls=[128, 128], dropout=0.5)
    raise ValueError("unknown segmentation config %r" % name)


class PointNet2_D(nn.Module):
    """Dense PointNet++ segmentation model with the per-object category fed to the classifier
    (models/segmentation/pointnet2.py:19-110): nested U-Net `model`, then `FC_layer` = Conv1D+BN+LeakyReLU over
    [features, one-hot category], Dropout, Conv1D with bias to the class scores.

    forward(data[, category]) -> scores (B*N, num_classes), the layout the reference hands to cross_entropy (:101)."""

    def __init__(self, input_nc, num_classes, config="pointnet2_charlesmsg", num_categories=0, kernels=None, fused=True):
        super().__init__()
        cfg = segmentation_config(config, input_nc) if isinstance(config, str) else config
        self.config = cfg
        self.fused = fused and kernels is None
        self._num_classes = num_classes
        self._num_categories = num_categories
        n = len(cfg["down_conv_nn"])

        def down(i):
            return lambda: PointNetMSGDown(npoint=cfg["npoint"][i], radii=cfg["radii"][i], nsample=cfg["nsample"][i],
                                           down_conv_nn=cfg["down_conv_nn"][i], normalize_xyz=cfg["normalize_xyz"][i],
                                           index=i, kernels=kernels, fused=fused)

        def up(j):
            return lambda: DenseFPModule(up_conv_nn=cfg["up_conv_nn"][j], index=j, kernels=kernels, fused=fused)

        block = UnetSkipConnectionBlock(up(0), make_inner=lambda: GlobalDenseBaseModule(
            nn=cfg["innermost"], fused=fused and kernels is None))
        for index in range(n - 1, 0, -1):
            block = UnetSkipConnectionBlock(up(n - index), make_down=down(index), submodule=block)
        self.model = UnetSkipConnectionBlock(up(n), make_down=down(0), submodule=block)
        widths = list(cfg["mlp_cls"])
        widths[0] += num_categories
        self.FC_layer = Seq()
        for a, b in zip(widths[:-1], widths[1:]):
            self.FC_layer.append(Conv1D(a, b, bn=True, bias=False))
        if cfg["dropout"]:
            self.FC_layer.append(nn.Dropout(p=cfg["dropout"]))
        self.FC_layer.append(Seq().append(nn.Conv1d(widths[-1], num_classes, kernel_size=1, bias=True)))

    def stages(self):
        """(down modules outermost first, global module, up modules innermost first) of the nested model."""
        downs, ups, b = [], [], self.model
        while not b.innermost:
            downs.append(b.down)
            ups.insert(0, b.up)
            b = b.submodule
        ups.insert(0, b.up)
        return downs, b.inner, ups

    def _with_category(self, x, category):
        if not self._num_categories:
            return None
        if category is None:
            raise ValueError("this model was built with num_categories=%d: pass `category`" % self._num_categories)
        return torch.nn.functional.one_hot(category.long(), self._num_categories).float()  # (B, N, K)

    def classifier_hidden(self, x, category=None):
        """x (B, C, N) -> rows (B*N, width) after the Conv1D+BN+LeakyReLU layers of FC_layer (before its Dropout)."""
        B, _, n = x.shape
        onehot = self._with_category(x, category)
        hidden = [m for m in self.FC_layer.children() if isinstance(m, Conv1D)]
        if self.fused and x.is_cuda:
            parts = [_fused.layer_parts(m) for m in hidden]
            if all(p is not None for p in parts):
                rows = _fused.cat_rows([_fused._cl(x)] + ([onehot] if onehot is not None else []))
                return _fused.run_mlp(rows, parts)
        if onehot is not None:
            x = torch.cat((x, onehot.transpose(1, 2)), dim=1)
        for m in hidden:
            x = m(x)
        return x.transpose(1, 2).reshape(B * n, -1)

    def classify(self, x, category=None):
        """x (B, C, N) features -> scores (B*N, num_classes)."""
        rows = self.classifier_hidden(x, category)
        layers = list(self.FC_layer.children())
        for m in layers:
            if isinstance(m, nn.Dropout):
                rows = m(rows)
        last = layers[-1][0]
        return torch.addmm(last.bias, rows, last.weight.reshape(self._num_classes, -1).t())

    def precompute_geometry(self, pos, backward_tables=False):
        """as PointNet2Unet.precompute_geometry: sampling, neighbour tables and interpolation tables of `pos` (B,N,3)"""
        downs, _, ups = self.stages()
        levels, cur, positions = [], pos, [pos]
        for i, down in enumerate(downs):
            g = down.precompute(cur, backward_tables=backward_tables and i > 0)
            levels.append(g)
            cur = g.new_pos
            positions.append(cur)
        tables, below = [], None
        for up in ups:
            skip_pos = positions.pop()
            tables.append(up.precompute(below, skip_pos, backward_tables=backward_tables))
            below = skip_pos
        return Data(down=levels, up=tables)

    def forward(self, data, category=None, geometry=None):
        x = None
        if data.x is not None:
            x = data.x.transpose(1, 2)
            if not (self.fused and x.is_cuda):
                x = x.contiguous()
        cur = Data(pos=data.pos, x=x)
        if geometry is None:
            out = self.model(cur)  # the nested blocks' own recursion
        else:
            # the same module order as that recursion (down ... global ... up), each stage with its precomputed tables
            downs, inner, ups = self.stages()
            skips = [cur]
            for i, down in enumerate(downs):
                cur = down(cur, precomputed=geometry.down[i])
                skips.append(cur)
            cur = inner(cur)
            for i, up in enumerate(ups):
                cur = up((cur, skips.pop()), precomputed=geometry.up[i])
            out = cur
        return self.classify(out.x, category)


def PointNet2(architecture="unet", input_nc=None, num_layers=3, output_nc=None, multiscale=False, kernels=None):
    """Config-free factory with the reference's signature (applications/pointnet2.py:22-55); unet only."""
    if architecture != "unet":
        raise NotImplementedError("only the unet architecture is on the hot path")
    name = "unet_{}_{}".format(num_layers, "ms" if multiscale else "ss")
    return PointNet2Unet(input_nc, output_nc=output_nc, config=name, kernels=kernels)


def synthetic_batch(B, N, feat, device="cpu", seed=1234):
    """The bench/parity input of SURVEY.md 8d: pos ~ U[-1,1]^3, x ~ N(0,1), fixed seed."""
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand(B, N, 3, generator=g) * 2 - 1
    x = torch.randn(B, N, feat, generator=g)
    return Data(pos=pos.to(device), x=x.to(device))
