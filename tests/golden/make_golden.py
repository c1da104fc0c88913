#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ by running the REFERENCE's own Python modules.

Runs only in the build container (needs /root/reference); the fixtures it writes are plain data (inputs,
indices, per-stage outputs, gradients) and are what travels to the GPU box.

How: torch-points3d's dense PointNet++ modules (PointNetMSGDown, DenseFPModule, GlobalDenseBaseModule, Conv1D)
are imported from /root/reference unmodified.  Their third-party imports that are not installed here
(torch_geometric, torch_scatter, omegaconf) are replaced by empty stand-in modules in sys.modules -- none of
them is called on this path -- and `torch_points_kernels` (torch-points-kernels 0.7.0, not installed, source
not in the reference tree) is bound to the CPU oracle oracle/tpk_ref.py.  So the fixtures pin
"reference modules + oracle kernels"; the kernel boundary itself is pinned by the reference's known-answer
tests restated in tests/test_oracle_kat.py.

The U-Net assembly of applications/pointnet2.py needs a real OmegaConf and cannot be imported; the script
assembles the reference modules by hand in the order unet.py:400-487 / pointnet2.py:154-191 prescribe.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import tpk_ref  # noqa: E402


class _Bag(object):
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None]


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    def _na(*a, **k):
        raise RuntimeError("stubbed third-party function called")

    class _MP(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tg = _stub("torch_geometric")
    tg.nn = _stub("torch_geometric.nn", knn_interpolate=_na, fps=_na, radius=_na, global_max_pool=_na,
                  global_mean_pool=_na, knn=_na, voxel_grid=_na, PointConv=_MP, MessagePassing=_MP)
    tg.nn.pool = _stub("torch_geometric.nn.pool")
    _stub("torch_geometric.nn.pool.consecutive", consecutive_cluster=_na)
    _stub("torch_geometric.nn.pool.pool", pool_pos=_na, pool_batch=_na)
    tg.data = _stub("torch_geometric.data", Data=_Bag, Batch=_Bag)
    _stub("torch_scatter", scatter_add=_na, scatter_mean=_na, scatter_max=_na)
    oc = _stub("omegaconf", OmegaConf=object, DictConfig=dict, ListConfig=list)
    oc.listconfig = _stub("omegaconf.listconfig", ListConfig=type("ListConfig", (list,), {}))
    oc.dictconfig = _stub("omegaconf.dictconfig", DictConfig=type("DictConfig", (dict,), {}))
    _stub("torch_points_kernels", furthest_point_sample=tpk_ref.furthest_point_sample,
          ball_query=tpk_ref.ball_query, three_nn=tpk_ref.three_nn, three_interpolate=tpk_ref.three_interpolate,
          grouping_operation=tpk_ref.grouping_operation)
    sys.path.insert(0, REF)


def build_reference_unet(cfg, output_nc, activation=None):
    """Reference modules assembled in the reference's order (down, inner, up, head)."""
    from torch_points3d.core.base_conv.dense import DenseFPModule, GlobalDenseBaseModule
    from torch_points3d.core.common_modules.base_modules import Seq
    from torch_points3d.core.common_modules.dense_modules import Conv1D
    from torch_points3d.modules.pointnet2.dense import PointNetMSGDown

    kw = {} if activation is None else {"activation": activation}  # default = the modules' LeakyReLU(0.01)
    net = torch.nn.Module()
    net.down_modules = torch.nn.ModuleList()
    for i in range(len(cfg["down_conv_nn"])):
        net.down_modules.append(PointNetMSGDown(
            npoint=cfg["npoint"][i], radii=cfg["radii"][i], nsample=cfg["nsample"][i],
            down_conv_nn=cfg["down_conv_nn"][i], normalize_xyz=cfg["normalize_xyz"][i], index=i, **kw))
    net.inner_modules = torch.nn.ModuleList([GlobalDenseBaseModule(nn=cfg["innermost"], **kw)])
    net.up_modules = torch.nn.ModuleList([DenseFPModule(up_conv_nn=c, index=i, **kw)
                                          for i, c in enumerate(cfg["up_conv_nn"])])
    net.mlp = Seq()
    net.mlp.append(Conv1D(cfg["up_conv_nn"][-1][-1], output_nc, bn=True, bias=False, **kw))
    return net


def run_reference_unet(net, pos, x, record):
    """PointNet2Unet.forward (applications/pointnet2.py:154-191) over the reference modules."""
    data = _Bag(pos=pos, x=x.transpose(1, 2).contiguous())
    stack = [data]
    for i in range(len(net.down_modules) - 1):
        data = net.down_modules[i](data)
        record["down%d_x" % i] = data.x
        record["down%d_pos" % i] = data.pos
        stack.append(data)
    data = net.down_modules[-1](data)
    last = len(net.down_modules) - 1
    record["down%d_x" % last] = data.x
    record["down%d_pos" % last] = data.pos
    stack.append(data)
    data = net.inner_modules[0](data)
    record["inner_x"] = data.x
    for i in range(len(net.up_modules)):
        data = net.up_modules[i]((data, stack.pop()))
        record["up%d_x" % i] = data.x
    data.x = net.mlp(data.x)
    record["out_x"] = data.x
    return data


def state_checksums(sd):
    out = {}
    for k, v in sd.items():
        v = v.double()
        out[k] = np.array([float(v.sum()), float(v.abs().sum())])
    return out


def kernel_level_records(cfg, pos):
    """Indices the reference modules obtained from the kernel boundary, recomputed stage by stage."""
    rec = {}
    cur = pos
    for i in range(len(cfg["down_conv_nn"])):
        fps = tpk_ref.furthest_point_sample(cur, cfg["npoint"][i])
        new = cur.gather(1, fps.unsqueeze(-1).repeat(1, 1, 3))
        rec["fps%d" % i] = fps
        for s, (r, ns) in enumerate(zip(cfg["radii"][i], cfg["nsample"][i])):
            idx, d2 = tpk_ref.ball_query(r, ns, cur, new)
            rec["ball%d_%d_idx" % (i, s)] = idx
            rec["ball%d_%d_d2" % (i, s)] = d2
        cur = new
    return rec


def to_np(rec):
    out = {}
    for k, v in rec.items():
        v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        if v.dtype == np.int64 and v.size and v.max() < 32767 and v.min() > -32768:
            v = v.astype(np.int16)  # indices are small; the tests widen them again
        out[k] = v
    return out


def make_case(name, cfg, feat, output_nc, pos, x, seed, store_weights, activation=None):
    torch.manual_seed(seed)
    net = build_reference_unet(cfg, output_nc, activation)
    net.train()  # the example never calls .eval(): BatchNorm uses batch statistics
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}  # weights BEFORE the forward pass
    x_in = x.clone().requires_grad_(True)
    rec = {}
    out = run_reference_unet(net, pos, x_in, rec)
    # one backward through three_interpolate / grouping / conv / BN for the gradient goldens
    gen = torch.Generator().manual_seed(seed + 1)
    cot = torch.randn(out.x.shape, generator=gen)
    (out.x * cot).sum().backward()
    rec["cotangent"] = cot
    rec["grad_x_in"] = x_in.grad
    rec["grad_first_conv"] = net.down_modules[0].mlps[0][0][0].weight.grad
    rec["grad_last_fp_conv"] = net.up_modules[-1].nn[0][0].weight.grad
    rec["pos"] = pos
    rec["x"] = x
    rec.update(kernel_level_records(cfg, pos))
    arrays = to_np(rec)
    # running statistics after one train-mode forward (BatchNorm momentum 0.1)
    arrays["bn_after/first_running_mean"] = net.down_modules[0].mlps[0][0][1].running_mean.detach().numpy()
    arrays["bn_after/first_running_var"] = net.down_modules[0].mlps[0][0][1].running_var.detach().numpy()
    for k, v in state_checksums(sd).items():
        arrays["cksum/" + k] = v
    if store_weights:
        for k, v in sd.items():
            arrays["state/" + k] = v.detach().cpu().numpy()
    arrays["meta_seed"] = np.array([seed])
    arrays["meta_feat_outnc"] = np.array([feat, output_nc])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KiB)" % (path, os.path.getsize(path) / 1024.0))


def make_kpconv_case():
    """KPConv_ops of the reference (modules/KPConv/convolution_ops.py:19-107, loaded by file path) on a small
    partial-dense neighbourhood table with -1 shadows, all influence / aggregation modes."""
    import importlib.util

    def by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    ops = by_path("ref_kpconv_ops", "torch_points3d/modules/KPConv/convolution_ops.py")
    ply = by_path("ref_plyutils", "torch_points3d/modules/KPConv/plyutils.py")
    kp = ply.read_ply(os.path.join(REF, "torch_points3d/modules/KPConv/kernels/dispositions/k_015_center.ply"))
    kp = np.vstack((kp["x"], kp["y"], kp["z"])).T.astype(np.float32)  # (15, 3) unit disposition
    g = torch.Generator().manual_seed(2024)
    M, Nq, Mn, Cin, Cout = 260, 200, 25, 8, 16
    support = torch.rand(M, 3, generator=g)
    query = support[torch.randperm(M, generator=g)[:Nq]].contiguous()
    point_influence = 0.12
    K_points = torch.from_numpy(kp) * (1.5 * point_influence)  # kernel radius = 1.5 * influence (kernels.py:35,51)
    idx, _ = tpk_ref.ball_query(0.3, Mn, support, query, mode="partial_dense", batch_x=torch.zeros(M, dtype=torch.long),
                                batch_y=torch.zeros(Nq, dtype=torch.long))
    assert (idx == -1).any()
    feats = torch.randn(M, Cin, generator=g)
    W = torch.randn(15, Cin, Cout, generator=g) * 0.2
    arrays = {"support": support, "query": query, "neighbors": idx, "features": feats, "K_points": K_points,
              "K_values": W, "extent": torch.tensor([point_influence])}
    for infl in ("constant", "linear", "gaussian"):
        for aggr in ("sum", "closest"):
            out = ops.KPConv_ops(query, support, idx.clone(), feats, K_points, W, point_influence, infl, aggr)
            arrays["out_%s_%s" % (infl, aggr)] = out
    path = os.path.join(HERE, "kpconv_ops.npz")
    np.savez_compressed(path, **to_np(arrays))
    print("wrote %s (%.1f KiB)" % (path, os.path.getsize(path) / 1024.0))


def make_kpconv_blocks_case():
    """Partial-dense KPConv path through the REFERENCE's own block classes (modules/KPConv/blocks.py: SimpleBlock,
    ResnetBBlock, KPDualBlock; core/base_conv/partial_dense.py: FPModule_PD; core/common_modules/base_modules.py: MLP,
    FastBatchNorm1d) composed as applications/conf/kpconv/unet_4.yaml composes its first two levels and last decoder
    stage (narrow widths).  Third-party pieces that are not installed are bound to the CPU oracle restatements:
    torch_points_kernels.ball_query -> oracle/tpk_ref_cpu.c, GridSampling3D -> oracle/voxel_ref.py (the reference's
    transform module needs torch_cluster), torch_geometric.knn_interpolate -> brute-force kNN + the published
    inverse-squared-distance formula.  So the fixture pins the block LOGIC (radius rule, bottleneck, BatchNorm
    momentum, strided shortcut, skip concatenation, kernel-point scaling) as the reference wrote it."""
    from oracle import voxel_ref

    class CpuGridSampling3D(object):
        def __init__(self, size, quantize_coords=False, mode="mean", verbose=False):
            self._grid_size = size

        def __call__(self, data):
            out = voxel_ref.grid_sampling_mean(data.pos.numpy(), self._grid_size, batch=data.batch.numpy(),
                                               x=data.x.detach().numpy())
            data.pos, data.batch = torch.from_numpy(out["pos"]), torch.from_numpy(out["batch"])
            data.x = torch.from_numpy(out["x"])
            return data

    def knn_interpolate(x, pos_x, pos_y, batch_x=None, batch_y=None, k=3, num_workers=1):
        idx, d2 = tpk_ref.knn(k, pos_x, pos_y, batch_x, batch_y)
        Nq = pos_y.shape[0]
        y_idx = torch.arange(Nq).repeat_interleave(k)
        x_idx = idx.reshape(-1)
        keep = x_idx >= 0
        w = (1.0 / torch.clamp(d2.reshape(-1, 1), min=1e-16))[keep]
        y_idx, x_idx = y_idx[keep], x_idx[keep]
        num = torch.zeros(Nq, x.shape[1]).index_add_(0, y_idx, x[x_idx] * w)
        return num / torch.zeros(Nq, 1).index_add_(0, y_idx, w)

    class _BILM(torch.nn.Module):
        pass

    class _Data(_Bag):  # the reference blocks call data.clone()
        def clone(self):
            out = _Data()
            for k, v in self.__dict__.items():
                setattr(out, k, v.clone() if torch.is_tensor(v) else v)
            return out

    _stub("torch_points3d.core.data_transform", GridSampling3D=CpuGridSampling3D)
    _stub("torch_points3d.models.base_model", BaseInternalLossModule=_BILM)
    try:
        import matplotlib  # noqa: F401
    except Exception:
        _stub("matplotlib").pyplot = _stub("matplotlib.pyplot")
    sys.modules["torch_geometric.nn"].knn_interpolate = knn_interpolate
    from torch_points3d.modules.KPConv.blocks import KPDualBlock
    import torch_points3d.core.base_conv.partial_dense as ref_pd
    import torch_points3d.core.spatial_ops.interpolate as ref_interp
    ref_interp.knn_interpolate = knn_interpolate  # it was imported by name before the stub was replaced
    ref_pd.Batch = _Data

    g = torch.Generator().manual_seed(77)
    N, grid, f = 1800, 0.04, 8
    pos = torch.rand(N, 3, generator=g) * 0.7
    batch = torch.sort(torch.randint(0, 2, (N,), generator=g))[0]
    x = torch.cat([torch.ones(N, 1), torch.randn(N, 3, generator=g)], 1)
    torch.manual_seed(5)
    np.random.seed(5)  # the kernel-point disposition gets a random rotation (kernel_utils.py:251-280)
    level0 = KPDualBlock(block_names=["SimpleBlock", "ResnetBBlock"], down_conv_nn=[[4, f], [f, 2 * f]],
                         grid_size=[grid, grid], prev_grid_size=[grid, grid], has_bottleneck=[False, True],
                         max_num_neighbors=[20, 20], deformable=[False, False], module_name="KPDualBlock", index=0)
    level1 = KPDualBlock(block_names=["ResnetBBlock", "ResnetBBlock"], down_conv_nn=[[2 * f, 2 * f], [2 * f, 4 * f]],
                         grid_size=[2 * grid, 2 * grid], prev_grid_size=[grid, 2 * grid], has_bottleneck=[True, True],
                         max_num_neighbors=[20, 20], deformable=[False, False], module_name="KPDualBlock", index=1)
    up = ref_pd.FPModule_PD(up_k=1, up_conv_nn=[4 * f + 2 * f, f], skip=True, bn_momentum=0.2, module_name="FPModule_PD",
                            index=0)
    net = torch.nn.ModuleDict({"level0": level0, "level1": level1, "up": up})
    net.train()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    xin = x.clone().requires_grad_(True)
    d0 = level0(_Data(pos=pos, batch=batch, x=xin))
    d1 = level1(d0)
    out = up((d1, d0))
    loss = (out.x * torch.linspace(-1.0, 1.0, f)).sum()
    loss.backward()
    arrays = {"pos": pos, "batch": batch, "x": x, "grid": torch.tensor([grid]), "width": torch.tensor([f]),
              "l0_x": d0.x, "l0_idx": d0.idx_neighboors, "l1_pos": d1.pos, "l1_batch": d1.batch, "l1_x": d1.x,
              "l1_idx": d1.idx_neighboors, "out_x": out.x, "loss": loss.reshape(1), "grad_x": xin.grad}
    for k, v in state.items():
        arrays["sd." + k] = v
    for k, v in net.state_dict().items():
        if "running_" in k:
            arrays["after." + k] = v
    for k, p in net.named_parameters():
        if p.grad is not None:
            arrays["grad." + k] = p.grad
    path = os.path.join(HERE, "kpconv_blocks.npz")
    np.savez_compressed(path, **to_np(arrays))
    print("wrote %s (%.1f KiB): levels %d -> %d points, neighbour slots filled %.0f%% / %.0f%%" % (
        path, os.path.getsize(path) / 1024.0, N, d1.pos.shape[0], 100 * float((d0.idx_neighboors >= 0).float().mean()),
        100 * float((d1.idx_neighboors >= 0).float().mean())))


def make_grid_sampling_case():
    """GridSampling3D through the REFERENCE's own transform code (core/data_transform/grid_transform.py:33-141:
    `group_data`, `GridSampling3D._process`), loaded by file path.  Its third-party calls (torch_cluster.grid_cluster,
    torch_geometric voxel_grid / consecutive_cluster, torch_scatter scatter_mean / scatter_add) are bound to the
    numpy restatements of oracle/voxel_ref.py, so the fixture pins the transform's own logic: which attributes are
    grouped how (mean / majority vote over one-hot sums / representative point / bool round trip), `coords`,
    `grid_size`, untouched attributes."""
    import importlib.util
    from oracle import voxel_ref

    class Data(object):  # the slice of torch_geometric.data.Data the transform touches
        def __init__(self, **kw):
            for k, v in kw.items():
                setattr(self, k, v)

        @property
        def keys(self):
            return [k for k, v in self.__dict__.items() if v is not None]

        @property
        def num_nodes(self):
            return self.pos.shape[0]

        def __iter__(self):
            for k in list(self.keys):
                yield k, getattr(self, k)

        def __contains__(self, k):
            return k in self.keys

        def __getitem__(self, k):
            return getattr(self, k)

        def __setitem__(self, k, v):
            setattr(self, k, v)

    def grid_cluster(pos, size, start=None, end=None):
        return torch.from_numpy(voxel_ref.grid_cluster_key(pos.numpy()))

    def voxel_grid(pos, batch, size, start=None, end=None):
        return torch.from_numpy(voxel_ref.grid_cluster_key(pos.numpy(), batch.numpy()))

    def consecutive_cluster(src):
        inv, perm = voxel_ref.consecutive_cluster(src.numpy())
        return torch.from_numpy(inv), torch.from_numpy(perm)

    def scatter_mean(src, index, dim=0, dim_size=None):
        K = int(index.max()) + 1
        if src.is_floating_point():
            return torch.from_numpy(voxel_ref.scatter_mean(src.numpy(), index.numpy(), K))
        sums = torch.zeros((K,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, src)
        cnt = torch.bincount(index, minlength=K).clamp(min=1).reshape((-1,) + (1,) * (src.dim() - 1))
        return torch.div(sums, cnt, rounding_mode="floor")  # torch_scatter: floor division for integer inputs

    def scatter_add(src, index, dim=0, dim_size=None):
        K = int(index.max()) + 1
        return torch.zeros((K,) + tuple(src.shape[1:]), dtype=src.dtype).index_add_(0, index, src)

    sys.modules["torch_scatter"].scatter_mean = scatter_mean
    sys.modules["torch_scatter"].scatter_add = scatter_add
    sys.modules["torch_geometric.nn.pool.consecutive"].consecutive_cluster = consecutive_cluster
    sys.modules["torch_geometric.nn"].voxel_grid = voxel_grid
    sys.modules["torch_geometric.data"].Data = Data
    _stub("torch_cluster", grid_cluster=grid_cluster)
    spec = importlib.util.spec_from_file_location(
        "ref_grid_transform", os.path.join(REF, "torch_points3d/core/data_transform/grid_transform.py"))
    gt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gt)

    g = torch.Generator().manual_seed(31)
    N = 3000
    pos = torch.rand(N, 3, generator=g) * 0.5
    batch = torch.sort(torch.randint(0, 3, (N,), generator=g))[0]
    x = torch.randn(N, 5, generator=g)
    y = torch.randint(-1, 6, (N,), generator=g)
    inst = torch.randint(0, 40, (N,), generator=g)
    flag = torch.rand(N, generator=g) < 0.3
    origin_id = torch.arange(N)
    arrays = {"pos": pos, "batch": batch, "x": x, "y": y, "instance_labels": inst, "flag": flag, "size": torch.tensor([0.06])}
    out = gt.GridSampling3D(0.06, quantize_coords=True, mode="mean")(
        Data(pos=pos.clone(), batch=batch.clone(), x=x.clone(), y=y.clone(), instance_labels=inst.clone(),
             flag=flag.clone(), origin_id=origin_id, scalar=torch.tensor([7.0])))
    for k in ("pos", "batch", "x", "y", "instance_labels", "flag", "origin_id", "coords", "grid_size", "scalar"):
        arrays["out." + k] = getattr(out, k)
    nb = gt.GridSampling3D(0.1, mode="mean")(Data(pos=pos.clone(), x=x.clone()))  # no batch attribute
    arrays["nobatch.pos"], arrays["nobatch.x"] = nb.pos, nb.x
    path = os.path.join(HERE, "grid_sampling.npz")
    np.savez_compressed(path, **to_np(arrays))
    print("wrote %s (%.1f KiB): %d -> %d voxels" % (path, os.path.getsize(path) / 1024.0, N, out.pos.shape[0]))


def make_unet4_config_case():
    """applications/conf/kpconv/unet_4.yaml resolved the way the reference resolves it
    (utils/model_building_utils/model_definition_resolver.py:22-51: every string leaf is eval()'d with FEAT and the
    file's define_constants; applications/modelfactory.py overrides in_grid_size / in_feat from the factory arguments):
    the architecture table torch_points3d_amd.kpconv_unet.unet_config must reproduce."""
    import json
    import yaml
    with open(os.path.join(REF, "torch_points3d/applications/conf/kpconv/unet_4.yaml")) as f:
        cfg = yaml.safe_load(f)

    def resolve(obj, constants):
        if isinstance(obj, dict):
            return {k: resolve(v, constants) for k, v in obj.items()}
        if isinstance(obj, list):
            return [resolve(v, constants) for v in obj]
        if isinstance(obj, str):
            try:
                return eval(obj, dict(constants))
            except (NameError, ValueError, SyntaxError):
                return obj
        return obj

    out = {}
    for feat, in_feat, grid in ((3, 64, 0.02), (1, 32, 0.05)):
        constants = dict(cfg["define_constants"])
        constants.update({"FEAT": feat, "in_feat": in_feat, "in_grid_size": grid})
        out["feat%d_infeat%d_grid%g" % (feat, in_feat, grid)] = {
            "down_conv": resolve(cfg["down_conv"], constants), "up_conv": resolve(cfg["up_conv"], constants)}
    path = os.path.join(HERE, "kpconv_unet4_config.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %s" % path)


def make_rsconv_case():
    """Relation-Shape convolution, dense format (modules/RSConv/dense.py:18-190, 398-476): the reference's own
    RSConvSharedMSGDown / RSConvMSGDown over the oracle kernels, two stacked levels (first-layer mapper with the
    feature-raising branch, then a plain mapper), distinct clouds, train-mode BatchNorm, one backward."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_rsconv_dense",
                                                  os.path.join(REF, "torch_points3d/modules/RSConv/dense.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    g = torch.Generator().manual_seed(4321)
    pos = torch.rand(2, 500, 3, generator=g) * 2 - 1
    feats = torch.randn(2, 500, 4, generator=g)
    torch.manual_seed(21)
    # level 0: shared mapper over two scales; features = [centred xyz (3) | x (4)] raised 7 -> 16
    l0 = mod.RSConvSharedMSGDown(npoint=128, radii=[0.3, 0.45], nsample=[12, 20],
                                 down_conv_nn=[[10, 8, 16], [4 + 3, 16]], channel_raising_nn=[16, 24])
    # level 1: one mapper per scale; features = [centred xyz (3) | 48 channels]
    l1 = mod.RSConvMSGDown(npoint=32, radii=[0.6, 0.9], nsample=[16, 24], down_conv_nn=[10, 16, 48 + 3],
                           channel_raising_nn=[48 + 3, 40])
    l0.train()
    l1.train()
    rec = {"pos": pos, "x": feats}
    for name, m in (("l0", l0), ("l1", l1)):
        for k, v in m.state_dict().items():
            rec["state/%s/%s" % (name, k)] = v.detach().clone()
    x_in = feats.clone().requires_grad_(True)
    d0 = l0(_Bag(pos=pos, x=x_in.transpose(1, 2).contiguous()))
    d1 = l1(d0)
    rec["l0_x"], rec["l0_pos"], rec["l1_x"], rec["l1_pos"] = d0.x, d0.pos, d1.x, d1.pos
    cot = torch.randn(d1.x.shape, generator=torch.Generator().manual_seed(22))
    (d1.x * cot).sum().backward()
    rec["cotangent"] = cot
    rec["grad_x_in"] = x_in.grad
    rec["grad_l0_msg_conv"] = l0._mapper.nn["mlp_msg"][0][0].weight.grad
    rec["grad_l1_raise_conv"] = l1.mlp_out[0].weight.grad
    # what the modules obtained at the kernel boundary
    fps0 = tpk_ref.furthest_point_sample(pos, 128)
    rec["fps0"] = fps0
    for s, (r, ns) in enumerate(zip([0.3, 0.45], [12, 20])):
        rec["ball0_%d_idx" % s] = tpk_ref.ball_query(r, ns, pos, d0.pos)[0]
    rec["fps1"] = tpk_ref.furthest_point_sample(d0.pos, 32)
    for s, (r, ns) in enumerate(zip([0.6, 0.9], [16, 24])):
        rec["ball1_%d_idx" % s] = tpk_ref.ball_query(r, ns, d0.pos, d1.pos)[0]
    path = os.path.join(HERE, "rsconv_dense.npz")
    np.savez_compressed(path, **to_np(rec))
    print("wrote %s (%.1f KiB)" % (path, os.path.getsize(path) / 1024.0))


def main():
    install_stubs()
    if sys.argv[1:] == ["rsconv"]:  # only this fixture (the others are unchanged)
        return make_rsconv_case()
    make_kpconv_case()
    from torch_points3d_amd.pointnet2 import unet_config

    # (1) BASELINE config 1: examples/pointnet2_segmentation_forward.py:5-19 -- randn cloud duplicated to B=2,
    #     FEAT=5, 10 classes, unet_3_ss, weights from torch.manual_seed(0).
    torch.manual_seed(0)
    pos = torch.randn((1024, 3)).unsqueeze(0)
    feats = torch.randn((1024, 5)).unsqueeze(0)
    pos, feats = torch.cat([pos, pos], 0), torch.cat([feats, feats], 0)
    make_case("c1_example", unet_config("unet_3_ss", 5), 5, 10, pos, feats, seed=0, store_weights=False)

    # (2) distinct clouds, uniform cube (realistic full/partial balls), narrow network with stored weights.
    g = torch.Generator().manual_seed(1234)
    pos = torch.rand(3, 700, 3, generator=g) * 2 - 1
    feats = torch.randn(3, 700, 4, generator=g)
    small = dict(npoint=[160, 40], radii=[[0.35], [0.7]], nsample=[[24], [16]],
                 down_conv_nn=[[[4 + 3, 16, 16, 24]], [[24 + 3, 24, 24, 32]]], innermost=[32 + 3, 32, 48],
                 up_conv_nn=[[48 + 32, 32, 32], [32 + 24, 32, 24], [24 + 4, 24, 24, 24]],
                 normalize_xyz=[False, True], save_sampling_id=[False, False])
    make_case("small_ssg", small, 4, 6, pos, feats, seed=7, store_weights=True)

    # (2b) same, with a smooth activation handed to the reference modules (they take `activation=`): without the
    #      LeakyReLU kink a last-bit forward difference cannot flip a gradient mask, so gradients compare tightly.
    make_case("small_ssg_tanh", small, 4, 6, pos, feats, seed=7, store_weights=True, activation=torch.nn.Tanh())

    # (2c) LeakyReLU(negative_slope=1.0) == identity: still a LeakyReLU (so the fused channel-last kernels run it)
    #      but kink-free, so the gradients of the fused path can be compared element-wise as well.
    make_case("small_ssg_slope1", small, 4, 6, pos, feats, seed=7, store_weights=True,
              activation=torch.nn.LeakyReLU(negative_slope=1.0))

    # (3) multi-scale grouping (unet_3_ms.yaml layout, narrow) on distinct clouds.
    g = torch.Generator().manual_seed(99)
    pos = torch.rand(2, 600, 3, generator=g) * 2 - 1
    feats = torch.randn(2, 600, 3, generator=g)
    msg = dict(npoint=[128, 32], radii=[[0.2, 0.4], [0.5, 0.9]], nsample=[[8, 16], [16, 24]],
               down_conv_nn=[[[3 + 3, 8, 12], [3 + 3, 8, 16]], [[12 + 16 + 3, 16, 24], [12 + 16 + 3, 16, 20]]],
               innermost=[24 + 20 + 3, 32, 48], up_conv_nn=[[48 + 44, 32, 32], [32 + 28, 24, 24], [24 + 3, 16, 16]],
               normalize_xyz=[False, False], save_sampling_id=[False, False])
    make_case("small_msg", msg, 3, 5, pos, feats, seed=11, store_weights=True)

    make_rsconv_case()

    # (4) KPConv blocks + FPModule_PD through the reference's own classes (last: it replaces further modules by stubs)
    make_kpconv_blocks_case()
    make_grid_sampling_case()
    make_unet4_config_case()


if __name__ == "__main__":
    main()
