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


def build_reference_fc_layer(cfg, output_nc):
    """models/segmentation/pointnet2.py:50-61: FC_layer = Conv1D(+BN+act) ..., Dropout, Conv1D(bias, no BN, no act)"""
    from torch_points3d.core.common_modules.base_modules import Seq
    from torch_points3d.core.common_modules.dense_modules import Conv1D
    nn_cls = list(cfg["mlp_cls"])
    nn_cls[0] += cfg["num_categories"]
    fc = Seq()
    for i in range(1, len(nn_cls)):
        fc.append(Conv1D(nn_cls[i - 1], nn_cls[i], bn=True, bias=False))
    if cfg["dropout"]:
        fc.append(torch.nn.Dropout(p=cfg["dropout"]))
    fc.append(Conv1D(nn_cls[-1], output_nc, activation=None, bias=True, bn=False))
    return fc


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
    if cfg.get("head") == "pointnet2_d":
        net.FC_layer = build_reference_fc_layer(cfg, output_nc)
        return net
    net.mlp = Seq()
    net.mlp.append(Conv1D(cfg["up_conv_nn"][-1][-1], output_nc, bn=True, bias=False, **kw))
    return net


def build_reference_nested_unet(cfg, output_nc):
    """The segmentation model of models/segmentation/pointnet2.py (PointNet2_D) as UnetBasedModel.__init__ nests it
    (models/base_architectures/unet.py:150-190): innermost block (global module + first up conv) first, then one
    UnetSkipConnectionBlock per remaining down conv from the deepest outwards, then FC_layer.  unet.py is loaded by
    file path with stand-ins for its dataset / base-model imports; the option parsing (OmegaConf) is bypassed, the
    blocks get the argument dicts it would have produced."""
    import importlib.util
    _stub("torch_points3d.datasets.base_dataset", BaseDataset=object)
    _stub("torch_points3d.models.base_model", BaseModel=torch.nn.Module, BaseInternalLossModule=torch.nn.Module)
    spec = importlib.util.spec_from_file_location("_ref_unet", os.path.join(REF, "torch_points3d/models/base_architectures/unet.py"))
    unet = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(unet)
    from torch_points3d.core.base_conv.dense import DenseFPModule, GlobalDenseBaseModule
    from torch_points3d.modules.pointnet2.dense import PointNetMSGDown
    lib = types.SimpleNamespace(PointNetMSGDown=PointNetMSGDown, DenseFPModule=DenseFPModule,
                                GlobalDenseBaseModule=GlobalDenseBaseModule)

    def down_args(i):
        return dict(down_conv_cls=PointNetMSGDown, npoint=cfg["npoint"][i], radii=cfg["radii"][i],
                    nsample=cfg["nsample"][i], down_conv_nn=cfg["down_conv_nn"][i], index=i)

    def up_args(j):
        return dict(up_conv_cls=DenseFPModule, up_conv_nn=cfg["up_conv_nn"][j], skip=True, index=j)

    n = len(cfg["down_conv_nn"])
    block = unet.UnetSkipConnectionBlock(args_up=up_args(0), modules_lib=lib, innermost=True,
                                         args_innermost=dict(module_name="GlobalDenseBaseModule", nn=cfg["innermost"]))
    for index in range(n - 1, 0, -1):
        block = unet.UnetSkipConnectionBlock(args_up=up_args(n - index), args_down=down_args(index), modules_lib=lib,
                                             submodule=block)
    net = torch.nn.Module()
    net.model = unet.UnetSkipConnectionBlock(args_up=up_args(n), args_down=down_args(0), submodule=block, outermost=True)
    downs, ups, b = [], [], net.model
    while not b.innermost:
        downs.append(b.down)
        ups.insert(0, b.up)
        b = b.submodule
    ups.insert(0, b.up)
    # plain lists (not registered as sub-modules): the stage accessors make_case uses
    object.__setattr__(net, "down_modules", downs)
    object.__setattr__(net, "inner_modules", [b.inner])
    object.__setattr__(net, "up_modules", ups)
    net.FC_layer = build_reference_fc_layer(cfg, output_nc)
    return net


def run_reference_unet(net, pos, x, record, category=None):
    """PointNet2Unet.forward (applications/pointnet2.py:154-191) over the reference modules; with a PointNet2_D head
    (models/segmentation/pointnet2.py:87-110) the category one-hot is concatenated and FC_layer applied.  Its Dropout
    is random in train mode: `fc0_x` (before it) is recorded, `out_x` only makes sense in eval mode."""
    data = _Bag(pos=pos, x=x.transpose(1, 2).contiguous())
    if hasattr(net, "model"):  # the reference's own nested UnetSkipConnectionBlock recursion (unet.py:288-297)
        hooks = []
        for i, m in enumerate(net.down_modules):
            hooks.append(m.register_forward_hook(
                lambda mod, inp, out, i=i: record.update({"down%d_x" % i: out.x, "down%d_pos" % i: out.pos})))
        hooks.append(net.inner_modules[0].register_forward_hook(lambda mod, inp, out: record.update({"inner_x": out.x})))
        for i, m in enumerate(net.up_modules):
            hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: record.update({"up%d_x" % i: out.x})))
        data = net.model(data)
        for h in hooks:
            h.remove()
        return _reference_head(net, data, record, category)
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
    return _reference_head(net, data, record, category)


def _reference_head(net, data, record, category):
    if hasattr(net, "FC_layer"):
        last = data.x
        if category is not None:
            ncat = net.FC_layer[0][0].in_channels - last.shape[1]
            onehot = torch.nn.functional.one_hot(category, ncat).to(last.dtype).transpose(1, 2)
            last = torch.cat((last, onehot), dim=1)
        record["fc0_x"] = net.FC_layer[0](last)
        data.x = net.FC_layer(last)
        record["out_x"] = data.x
        return data
    data.x = net.mlp(data.x)
    record["out_x"] = data.x
    return data


def exact_checksum(v):
    """Two int64 numbers that pin a tensor bit for bit and do not depend on the order a machine sums in: the sum of
    the elements' bit patterns and a position-weighted sum of them (integer arithmetic, no overflow below 2^20
    elements x 2^41)."""
    b = v.detach().contiguous()
    b = b.float().view(torch.int32) if b.is_floating_point() else b
    b = b.reshape(-1).to(torch.int64)
    w = torch.arange(b.numel(), dtype=torch.int64) % 1021 + 1
    return np.array([int(b.sum()), int((b * w).sum())], dtype=np.int64)


def state_checksums(sd):
    return {k: exact_checksum(v) for k, v in sd.items()}


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


def _grouping_any(features, idx):
    """grouping_operation for any floating dtype (the fp64 evaluation): out[b,c,j,s] = features[b,c,idx[b,j,s]]"""
    B, C, _ = features.shape
    return features.gather(2, idx.reshape(B, 1, -1).expand(B, C, -1)).reshape(B, C, *idx.shape[1:])


class _Fp64Kernels(object):
    """Context: torch_points_kernels as seen by the reference modules evaluates features in fp64 -- indices come
    from the fp32 oracle on the fp32 coordinates (so they are the golden's indices), distances and feature
    arithmetic are done in double."""
    NAMES = ("furthest_point_sample", "ball_query", "three_nn", "three_interpolate", "grouping_operation")

    def __enter__(self):
        tp = sys.modules["torch_points_kernels"]
        self.saved = {n: getattr(tp, n) for n in self.NAMES}

        def three_nn(unknown, known):
            _, idx = tpk_ref.three_nn(unknown.float(), known.float())
            B, n, _ = idx.shape
            nb = known.gather(1, idx.reshape(B, -1, 1).expand(B, n * 3, 3)).reshape(B, n, 3, 3)
            return ((nb - unknown.unsqueeze(2)) ** 2).sum(-1).sqrt(), idx

        tp.furthest_point_sample = lambda xyz, n: tpk_ref.furthest_point_sample(xyz.float(), n)
        tp.ball_query = lambda r, ns, a, b, **kw: tpk_ref.ball_query(r, ns, a.float(), b.float(), **kw)
        tp.three_nn = three_nn
        tp.three_interpolate = lambda f, idx, w: (_grouping_any(f, idx) * w.unsqueeze(1)).sum(-1)
        tp.grouping_operation = _grouping_any
        return self

    def __exit__(self, *exc):
        tp = sys.modules["torch_points_kernels"]
        for n, f in self.saved.items():
            setattr(tp, n, f)
        return False


def _bn_modules(net):
    return [m for m in net.modules() if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d))]


def _restore_buffers(net, saved):
    for k, v in net.state_dict().items():
        if k in saved and ("running_" in k or "num_batches" in k):
            v.copy_(saved[k])


def probe_conditioning(net, pos, x, category=None):
    """(min |pre-activation| over every BatchNorm output, min non-zero top-2 gap of every max-pool input) of one
    train-mode forward: how far the forward pass is from a LeakyReLU kink / an arg-max switch."""
    saved = {k: v.clone() for k, v in net.state_dict().items()}
    vals = {"pre": float("inf"), "gap": float("inf")}
    hooks = []
    for m in _bn_modules(net):
        hooks.append(m.register_forward_hook(
            lambda mod, i, o: vals.__setitem__("pre", min(vals["pre"], float(o.detach().abs().min())))))

    def pool_hook(mod, i, o):
        o = o.detach()
        o = o if o.shape[-1] > 1 else o.squeeze(-1)  # global module: max over the points axis
        top = o.topk(2, dim=-1)[0]
        gap = (top[..., 0] - top[..., 1])
        gap = gap[gap > 0]
        if gap.numel():
            vals["gap"] = min(vals["gap"], float(gap.min()))

    for d in net.down_modules:
        for mlp in d.mlps:
            hooks.append(mlp.register_forward_hook(pool_hook))
    hooks.append(net.inner_modules[0].nn.register_forward_hook(pool_hook))
    with torch.no_grad():
        run_reference_unet(net, pos, x, {}, category)
    for h in hooks:
        h.remove()
    _restore_buffers(net, saved)
    return vals["pre"], vals["gap"]


def move_off_the_kink(net, pos, x, delta, category=None):
    """Shift every BatchNorm bias by the smallest amount that leaves no pre-activation of its channel within `delta`
    of zero (layers in execution order, one train-mode forward each: a shift changes everything downstream).  The
    fixture keeps LeakyReLU(0.01) and train-mode statistics, but a last-bit difference in a GEMM can no longer flip an
    activation mask, so gradients of two correct fp32 implementations agree element-wise."""
    saved = {k: v.clone() for k, v in net.state_dict().items()}
    order = []
    hooks = [m.register_forward_hook(lambda mod, i, o: order.append(mod)) for m in _bn_modules(net)]
    with torch.no_grad():
        run_reference_unet(net, pos, x, {}, category)
    for h in hooks:
        h.remove()
    _restore_buffers(net, saved)
    for bn in order:
        got = {}
        h = bn.register_forward_hook(lambda mod, i, o: got.__setitem__("o", o.detach()))
        with torch.no_grad():
            run_reference_unet(net, pos, x, {}, category)
        h.remove()
        _restore_buffers(net, saved)
        o = got["o"].transpose(0, 1).reshape(got["o"].shape[1], -1).double().numpy()  # (C, elements)
        for c in range(o.shape[0]):
            u = np.sort(-o[c])  # the shifts that would put an element exactly on the kink
            cands = []
            j = np.searchsorted(u, 0.0)
            # gaps of width >= 2*delta around 0, nearest first: scan outwards over the sorted kink positions
            lo_edges = np.concatenate(([-np.inf], u))
            hi_edges = np.concatenate((u, [np.inf]))
            for g in sorted(range(len(lo_edges)), key=lambda t: abs(t - j))[:4000]:
                a, b = lo_edges[g] + delta, hi_edges[g] - delta
                if a <= b:
                    cands.append(min(max(0.0, a), b))
                    if len(cands) >= 8:
                        break
            shift = min(cands, key=abs)
            if shift != 0.0:
                with torch.no_grad():
                    bn.bias[c] += float(shift)
        saved = {k: v.clone() for k, v in net.state_dict().items()}


def _subsample(t, cap, dup_batch):
    """A variant tensor (eval / fp64) is stored on a subset: cloud 0 only when the batch holds duplicates, and every
    s-th entry of the last (points) axis when it has more than `cap` elements.  Returns (tensor, [s, first_only])."""
    first = 1 if (dup_batch and t.dim() >= 2 and t.shape[0] > 1) else 0
    if first:
        t = t[:1]
    s = 1
    if t.dim() >= 3 and t.numel() > cap:
        s = int(-(-t.numel() // cap))
        t = t[..., ::s]
    return t.contiguous(), np.array([s, first])


def stage_input_grads(net, rec, x, pos, category, target_key):
    """gin/<stage>/<j>: the gradient one stage ALONE sends to its j-th input when the recorded gradient of its output
    (rec[<stage>].grad) enters it.  Each stage is run once more on detached copies of its recorded inputs (a tensor
    that feeds two stages -- a skip connection -- would otherwise collect both paths); BatchNorm buffers are restored."""
    saved = {k: v.clone() for k, v in net.state_dict().items()}
    out = {}

    def leaf(t):
        return t.detach().clone().requires_grad_(True)

    def emit(key, y, leaves):
        assert torch.equal(y.detach(), rec[key].detach()), key  # the re-run IS the recorded stage
        for j, g_ in enumerate(torch.autograd.grad(y, leaves, grad_outputs=rec[key].grad, allow_unused=True)):
            if g_ is not None:
                out["gin/%s/%d" % (key, j)] = g_

    nd = len(net.down_modules)
    level_x = [x] + [rec["down%d_x" % i] for i in range(nd)]       # features entering level i+1 (x is (B,N,C))
    level_pos = [pos] + [rec["down%d_pos" % i].detach() for i in range(nd)]

    def as_bcn(level, l):
        return l.transpose(1, 2).contiguous() if level == 0 else l

    for i in range(nd):
        l = leaf(level_x[i])
        emit("down%d_x" % i, net.down_modules[i](_Bag(pos=level_pos[i], x=as_bcn(i, l))).x, [l])
    l = leaf(level_x[nd])
    emit("inner_x", net.inner_modules[0](_Bag(pos=level_pos[nd], x=l)).x, [l])
    for i in range(len(net.up_modules)):
        lvl = nd - i  # level of the skip connection
        l1 = leaf(rec["inner_x"] if i == 0 else rec["up%d_x" % (i - 1)])
        l2 = leaf(level_x[lvl])
        data = _Bag(pos=None if i == 0 else level_pos[lvl + 1], x=l1)
        emit("up%d_x" % i, net.up_modules[i]((data, _Bag(pos=level_pos[lvl], x=as_bcn(lvl, l2)))).x, [l1, l2])
    l = leaf(rec["up%d_x" % (len(net.up_modules) - 1)])
    if target_key == "out_x":
        emit("out_x", net.mlp(l), [l])
    _restore_buffers(net, saved)
    return out


def make_case(name, cfg, feat, output_nc, pos, x, seed, store_weights, activation=None, category=None, kink_delta=0.0,
              dup_batch=False, variant_cap=65536, stage_grads=True, sub_out=1, variants=True):
    build = build_reference_nested_unet if cfg.get("nested") else (lambda c, o: build_reference_unet(c, o, activation))
    torch.manual_seed(seed)
    net = build(cfg, output_nc)
    net.train()  # the example never calls .eval(): BatchNorm uses batch statistics
    if kink_delta:
        move_off_the_kink(net, pos, x, kink_delta, category)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}  # weights BEFORE the forward pass
    x_in = x.clone().requires_grad_(True)
    rec = {}
    out = run_reference_unet(net, pos, x_in, rec, category)
    # one backward through three_interpolate / grouping / conv / BN for the gradient goldens.  With the PointNet2_D
    # head the cotangent enters BEFORE its Dropout (random in train mode), at fc0_x.
    target_key = "fc0_x" if "fc0_x" in rec else "out_x"
    gen = torch.Generator().manual_seed(seed + 1)
    cot = torch.randn(rec[target_key].shape, generator=gen)
    stage_keys = [k for k in rec if k.endswith("_x") and rec[k].requires_grad]
    for k in stage_keys:
        rec[k].retain_grad()
    (rec[target_key] * cot).sum().backward(retain_graph=True)
    arrays = {}
    if stage_grads:
        for key in stage_keys:
            if rec[key].grad is not None:
                arrays["gout/" + key] = rec[key].grad.detach().clone()
        for k, g_ in stage_input_grads(net, rec, x, pos, category, target_key).items():
            arrays[k] = g_
        if store_weights:
            for k, p_ in net.named_parameters():
                if p_.grad is not None:
                    arrays["pgrad/" + k] = p_.grad.detach().clone()
    rec_out = dict(rec)
    if sub_out > 1:  # per-point head tensors of the large fixture: every sub_out-th point
        for k in ("fc0_x", "out_x"):
            if k in rec_out:
                rec_out[k] = rec_out[k][..., ::sub_out]
        arrays["meta_sub_out"] = np.array([sub_out])
    if "fc0_x" in rec_out:
        del rec_out["out_x"]  # went through train-mode Dropout: not reproducible
    if cot.numel() <= 200000:
        rec_out["cotangent"] = cot
    # (larger cotangents are regenerated by the tests: torch.randn(shape, generator=manual_seed(seed + 1)))
    arrays["meta_cot_shape"] = np.array(list(cot.shape))
    rec_out["grad_x_in"] = x_in.grad
    rec_out["grad_first_conv"] = net.down_modules[0].mlps[0][0][0].weight.grad
    rec_out["grad_last_fp_conv"] = net.up_modules[-1].nn[0][0].weight.grad
    rec_out["pos"] = pos
    rec_out["x"] = x
    if category is not None:
        rec_out["category"] = category
    rec_out.update(kernel_level_records(cfg, pos))
    arrays.update(rec_out)
    arrays = to_np(arrays)
    # running statistics after one train-mode forward (BatchNorm momentum 0.1)
    arrays["bn_after/first_running_mean"] = net.down_modules[0].mlps[0][0][1].running_mean.detach().numpy().copy()
    arrays["bn_after/first_running_var"] = net.down_modules[0].mlps[0][0][1].running_var.detach().numpy().copy()
    after = {k: v.detach().clone() for k, v in net.state_dict().items() if "running_" in k}
    for k, v in after.items():
        arrays["after/" + k] = v.numpy()

    # ---- eval mode (running statistics as they stand after that one training pass): no batch coupling, so the
    #      1e-5 feature bar of BASELINE.json applies element-wise, stage by stage
    net.eval()
    rec_e = {}
    with torch.no_grad():
        if variants:
            run_reference_unet(net, pos, x, rec_e, category)
    for k, v in rec_e.items():
        if k.endswith("_x"):
            t, meta = _subsample(v, variant_cap, dup_batch)
            arrays["eval/" + k] = t.numpy()
            arrays["meta_sub/eval/" + k] = meta

    # ---- the same train-mode pass in fp64 (same indices): the yardstick for "how far may a correct fp32
    #      implementation be from the exact result" -- tests bound |GPU - fp64| by a multiple of |golden - fp64|
    if variants:
        torch.manual_seed(seed)
        net64 = build(cfg, output_nc)
        net64.load_state_dict(sd)
        net64 = net64.double().train()
        x64 = x.double().clone().requires_grad_(True)
        rec64 = {}
        with _Fp64Kernels():
            run_reference_unet(net64, pos.double(), x64, rec64, category)
            (rec64[target_key] * cot.double()).sum().backward()
        for k, v in rec64.items():
            if k.endswith("_x") and not (k == "out_x" and "fc0_x" in rec64):
                t, meta = _subsample(v.detach(), variant_cap, dup_batch)
                arrays["f64/" + k] = t.numpy()
                arrays["meta_sub/f64/" + k] = meta
        arrays["f64/grad_x_in"] = x64.grad.numpy()
        arrays["f64/grad_first_conv"] = net64.down_modules[0].mlps[0][0][0].weight.grad.numpy()
        arrays["f64/grad_last_fp_conv"] = net64.up_modules[-1].nn[0][0].weight.grad.numpy()

    # ---- conditioning of this forward pass (see probe_conditioning)
    torch.manual_seed(seed)
    netp = build(cfg, output_nc)
    netp.load_state_dict(sd)
    netp.train()
    pre, gap = probe_conditioning(netp, pos, x, category)
    arrays["meta_min_preact"] = np.array([pre])
    arrays["meta_min_pool_gap"] = np.array([gap])

    for k, v in state_checksums(sd).items():
        arrays["cksum/" + k] = v
    if store_weights:
        for k, v in sd.items():
            arrays["state/" + k] = v.detach().cpu().numpy()
    arrays["meta_seed"] = np.array([seed])
    arrays["meta_feat_outnc"] = np.array([feat, output_nc])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KiB)  min |pre-activation| %.3g, min pool gap %.3g" % (
        path, os.path.getsize(path) / 1024.0, pre, gap))
    return pre, gap


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


def make_kpconv_blocks_case(name="kpconv_blocks", slope=None):
    """Partial-dense KPConv path through the REFERENCE's own block classes (modules/KPConv/blocks.py: SimpleBlock,
    ResnetBBlock, KPDualBlock; core/base_conv/partial_dense.py: FPModule_PD; core/common_modules/base_modules.py: MLP,
    FastBatchNorm1d) composed as applications/conf/kpconv/unet_4.yaml composes its first two levels and last decoder
    stage (narrow widths).  Third-party pieces that are not installed are bound to the CPU oracle restatements:
    torch_points_kernels.ball_query -> oracle/tpk_ref_cpu.c, GridSampling3D -> oracle/voxel_ref.py (the reference's
    transform module needs torch_cluster), torch_geometric.knn_interpolate -> brute-force kNN + the published
    inverse-squared-distance formula.  So the fixture pins the block LOGIC (radius rule, bottleneck, BatchNorm
    momentum, strided shortcut, skip concatenation, kernel-point scaling) as the reference wrote it.

    Per stage (level0, level1, decoder) the fixture holds the train-mode fp32 output, the same pass in float64 (same
    sampled clouds and neighbour tables: positions go through the fp32 searches on both sides) and the eval-mode output
    after the one training pass.  slope: every activation replaced by LeakyReLU(slope) -- slope 1.0 gives the kink-free
    twin whose gradients two correct fp32 implementations agree on element-wise."""
    import copy
    from oracle import voxel_ref

    class CpuGridSampling3D(object):
        def __init__(self, size, quantize_coords=False, mode="mean", verbose=False):
            self._grid_size = size

        def __call__(self, data):
            dt = data.pos.dtype  # the float64 evaluation samples the SAME fp32 cloud
            out = voxel_ref.grid_sampling_mean(data.pos.float().numpy(), self._grid_size, batch=data.batch.numpy(),
                                               x=data.x.detach().float().numpy())
            data.pos, data.batch = torch.from_numpy(out["pos"]).to(dt), torch.from_numpy(out["batch"])
            data.x = torch.from_numpy(out["x"]).to(dt)  # (overwritten by the block: blocks.py:88)
            return data

    def knn_interpolate(x, pos_x, pos_y, batch_x=None, batch_y=None, k=3, num_workers=1):
        idx, _ = tpk_ref.knn(k, pos_x.float(), pos_y.float(), batch_x, batch_y)
        Nq = pos_y.shape[0]
        y_idx = torch.arange(Nq).repeat_interleave(k)
        x_idx = idx.reshape(-1)
        keep = x_idx >= 0
        y_idx, x_idx = y_idx[keep], x_idx[keep]
        d2 = ((pos_x[x_idx] - pos_y[y_idx]) ** 2).sum(-1, keepdim=True)
        w = 1.0 / torch.clamp(d2, min=1e-16)
        num = torch.zeros(Nq, x.shape[1], dtype=x.dtype).index_add_(0, y_idx, x[x_idx] * w)
        return num / torch.zeros(Nq, 1, dtype=x.dtype).index_add_(0, y_idx, w)

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
    import torch_points3d.modules.KPConv.blocks as ref_blocks
    import torch_points3d.core.base_conv.partial_dense as ref_pd
    import torch_points3d.core.spatial_ops.interpolate as ref_interp
    ref_blocks.GridSampling3D = CpuGridSampling3D  # (bound by name at the first import of the module)
    ref_interp.knn_interpolate = knn_interpolate  # it was imported by name before the stub was replaced
    ref_pd.Batch = _Data

    g = torch.Generator().manual_seed(77)
    N, grid, f = 1800, 0.04, 8
    pos = torch.rand(N, 3, generator=g) * 0.7
    batch = torch.sort(torch.randint(0, 2, (N,), generator=g))[0]
    x = torch.cat([torch.ones(N, 1), torch.randn(N, 3, generator=g)], 1)
    torch.manual_seed(5)
    np.random.seed(5)  # the kernel-point disposition gets a random rotation (kernel_utils.py:251-280)
    kw = {} if slope is None else {"activation": torch.nn.LeakyReLU(negative_slope=slope)}
    level0 = KPDualBlock(block_names=["SimpleBlock", "ResnetBBlock"], down_conv_nn=[[4, f], [f, 2 * f]],
                         grid_size=[grid, grid], prev_grid_size=[grid, grid], has_bottleneck=[False, True],
                         max_num_neighbors=[20, 20], deformable=[False, False], module_name="KPDualBlock", index=0, **kw)
    level1 = KPDualBlock(block_names=["ResnetBBlock", "ResnetBBlock"], down_conv_nn=[[2 * f, 2 * f], [2 * f, 4 * f]],
                         grid_size=[2 * grid, 2 * grid], prev_grid_size=[grid, 2 * grid], has_bottleneck=[True, True],
                         max_num_neighbors=[20, 20], deformable=[False, False], module_name="KPDualBlock", index=1, **kw)
    up = ref_pd.FPModule_PD(up_k=1, up_conv_nn=[4 * f + 2 * f, f], skip=True, bn_momentum=0.2, module_name="FPModule_PD",
                            index=0)
    if slope is not None:  # FPModule_PD's MLP takes no activation argument: swap the module
        for blk in up.nn:
            blk[2] = torch.nn.LeakyReLU(negative_slope=slope)
    net = torch.nn.ModuleDict({"level0": level0, "level1": level1, "up": up})
    net.train()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    net64 = copy.deepcopy(net).double()

    def run(model, xin, pp):
        d0 = model["level0"](_Data(pos=pp, batch=batch, x=xin))
        d1 = model["level1"](d0)
        return d0, d1, model["up"]((d1, d0))

    xin = x.clone().requires_grad_(True)
    d0, d1, out = run(net, xin, pos)
    # cotangent: per-channel weights (the original fixture); for the kink-free twin a random one -- with identity
    # activations the network ends in a BatchNorm, whose per-channel sum over the rows is a constant (zero gradient)
    weights = torch.linspace(-1.0, 1.0, f)
    if slope is not None:
        weights = torch.randn(out.x.shape, generator=torch.Generator().manual_seed(78))
    loss = (out.x * weights).sum()
    loss.backward()
    arrays = {"pos": pos, "batch": batch, "x": x, "grid": torch.tensor([grid]), "width": torch.tensor([f]),
              "l0_x": d0.x, "l0_idx": d0.idx_neighboors, "l1_pos": d1.pos, "l1_batch": d1.batch, "l1_x": d1.x,
              "l1_idx": d1.idx_neighboors, "out_x": out.x, "loss": loss.reshape(1), "grad_x": xin.grad}
    if slope is not None:
        arrays["slope"] = torch.tensor([slope])
        arrays["cotangent"] = weights
    for k, v in state.items():
        arrays["sd." + k] = v
    for k, v in net.state_dict().items():
        if "running_" in k:
            arrays["after." + k] = v
    for k, p in net.named_parameters():
        if p.grad is not None:
            arrays["grad." + k] = p.grad
    # ---- the same pass in float64 (same clouds, same tables), with its gradients
    x64 = x.double().clone().requires_grad_(True)
    with _Fp64Kernels():
        e0, e1, eout = run(net64, x64, pos.double())
        (eout.x * weights.double()).sum().backward()
    assert torch.equal(e1.pos.float(), d1.pos) and torch.equal(e1.idx_neighboors, d1.idx_neighboors)
    for key, v in (("l0_x", e0.x), ("l1_x", e1.x), ("out_x", eout.x), ("grad_x", x64.grad)):
        arrays["f64/" + key] = v.detach().numpy()
    for k, p in net64.named_parameters():
        if p.grad is not None and slope is not None:
            arrays["f64/grad." + k] = p.grad.numpy()
    # ---- eval mode (running statistics as they stand after the one training pass)
    net.eval()
    with torch.no_grad():
        v0, v1, vout = run(net, x, pos)
    for key, v in (("l0_x", v0.x), ("l1_x", v1.x), ("out_x", vout.x)):
        arrays["eval/" + key] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **to_np(arrays))
    dist = lambda a, b: float((a.detach().double() - b.detach()).abs().max())  # noqa: E731
    print("wrote %s (%.1f KiB): levels %d -> %d points, neighbour slots filled %.0f%% / %.0f%%; fp32-vs-fp64 distance of the "
          "reference pass: l0 %.2e, l1 %.2e, out %.2e" % (
              path, os.path.getsize(path) / 1024.0, N, d1.pos.shape[0], 100 * float((d0.idx_neighboors >= 0).float().mean()),
              100 * float((d1.idx_neighboors >= 0).float().mean()), dist(d0.x, e0.x), dist(d1.x, e1.x), dist(out.x, eout.x)))


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
    for name, m in (("l0", l0), ("l1", l1)):
        for k, v in m.state_dict().items():
            if "running_" in k:
                rec["after/%s/%s" % (name, k)] = v.detach().clone()
    # the same pass in float64 (same indices: the searches run on the fp32 coordinates), then the eval-mode pass
    import copy
    m0, m1 = copy.deepcopy(l0).double(), copy.deepcopy(l1).double()
    for m, src in ((m0, l0), (m1, l1)):  # the copies were taken after the training pass: rewind their statistics
        m.load_state_dict({k: rec["state/%s/%s" % ("l0" if src is l0 else "l1", k)].double()
                           if rec["state/%s/%s" % ("l0" if src is l0 else "l1", k)].is_floating_point()
                           else rec["state/%s/%s" % ("l0" if src is l0 else "l1", k)] for k in src.state_dict()})
        m.train()
    with _Fp64Kernels():
        e0 = m0(_Bag(pos=pos.double(), x=feats.double().transpose(1, 2).contiguous()))
        e1 = m1(e0)
    rec["f64/l0_x"], rec["f64/l1_x"] = e0.x.detach().numpy(), e1.x.detach().numpy()
    l0.eval()
    l1.eval()
    with torch.no_grad():
        v0 = l0(_Bag(pos=pos, x=feats.transpose(1, 2).contiguous()))
        v1 = l1(v0)
    rec["eval/l0_x"], rec["eval/l1_x"] = v0.x, v1.x
    print("rsconv: fp32-vs-fp64 distance of the reference pass: l0 %.2e, l1 %.2e" % (
        float((d0.x.detach().double() - e0.x.detach()).abs().max()), float((d1.x.detach().double() - e1.x.detach()).abs().max())))
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


def make_randla_case():
    """RandLA-Net local feature aggregation through the REFERENCE's own classes (modules/RandLANet/modules.py:
    RandlaKernel :9-54, RandlaConv :57-67, DilatedResidualBlock :70-101, RandLANetRes :104-123;
    core/base_conv/message_passing.py: BaseConvolutionDown :35-58, BaseResnetBlock :212-255; base_modules.MLP /
    FastBatchNorm1d), loaded from the reference tree.  What is not installed is bound as follows:

    * torch_geometric.nn.MessagePassing -> a stand-in whose propagate() does what PyG documents for aggr="add",
      flow="source_to_target": `<arg>_j` = arg[edge_index[0]] (source), `<arg>_i` = arg[edge_index[1]] (target),
      out = scatter-add of message(...) over edge_index[1], then update(out).  For a TUPLE argument PyG takes
      element 0 for `_j` and element 1 for `_i` (`tuple_order = "pyg"`; used for the kernel-level fixtures, which call
      RandlaKernel.forward directly with pos = (support positions, query positions)).
    * BaseConvolutionDown.forward (:55) hands the kernel `(pos[idx], pos)` -- target FIRST.  Under PyG's rule that reads
      pos_j = pos[idx][support index] and PyG's size check rejects it with a ValueError whenever the sample count
      differs from the cloud size (ratio < 1), which is every RandLA-Net level but the first; the reference's own model
      test skips randlanet (test/test_models.py:116-125).  The block-level fixtures therefore evaluate that tuple the
      way the call site evidently means it (`tuple_order = "callsite"`: element 0 = targets, element 1 = sources), which
      is also the paper's definition (pos_i = the sampled point, pos_j = its neighbour) and what torch_points3d_amd/randla.py
      computes.  Nothing else of the arithmetic is touched.
    * torch_geometric.nn.knn -> oracle/tpk_ref_cpu.c exact kNN as (row = query, col = support), query-major, closest
      first.  torch_cluster is absent, so the EDGE ORDER is unpinned; the sum over a query's 16 edges runs in that order
      on both sides.
    * RandomSampler draws with torch.randint: the drawn indices are stored and replayed by the GPU test.

    Every fixture: train-mode fp32 pass (+ running statistics after it), the same pass in float64, the eval-mode pass
    after it, one backward (input and parameter gradients)."""
    import copy
    import importlib.util
    import inspect

    class MessagePassing(torch.nn.Module):
        tuple_order = "pyg"

        def __init__(self, aggr="add", flow="source_to_target", node_dim=0):
            super().__init__()
            assert aggr == "add" and flow == "source_to_target" and node_dim == 0

        def propagate(self, edge_index, size=None, **kwargs):
            src, dst = edge_index[0], edge_index[1]
            n_dst = None
            feed = {}
            for name in inspect.signature(self.message).parameters:
                value, is_j = kwargs.get(name[:-2]), name.endswith("_j")
                if isinstance(value, (tuple, list)):
                    j_elem = 0 if self.tuple_order == "pyg" else 1
                    n_dst = value[1 - j_elem].shape[0]
                    value = value[j_elem if is_j else 1 - j_elem]
                feed[name] = None if value is None else value[src if is_j else dst]
            msg = self.message(**feed)
            if n_dst is None:
                n_dst = int(dst.max()) + 1
            out = torch.zeros(n_dst, msg.shape[1], dtype=msg.dtype).index_add_(0, dst, msg)
            return self.update(out)

    def knn_edges(x, y, k, batch_x=None, batch_y=None, **kw):
        idx, _ = tpk_ref.knn(k, x.float(), y.float(), batch_x, batch_y)
        row = torch.arange(y.shape[0]).repeat_interleave(k)
        col = idx.reshape(-1)
        keep = col >= 0
        return torch.stack([row[keep], col[keep]], 0)

    sys.modules["torch_geometric.nn"].MessagePassing = MessagePassing
    sys.modules["torch_geometric.nn"].knn = knn_edges
    import torch_points3d.core.spatial_ops.neighbour_finder as ref_nf
    ref_nf.knn = knn_edges  # bound by name at import
    spec = importlib.util.spec_from_file_location("_ref_randla", os.path.join(REF, "torch_points3d/modules/RandLANet/modules.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import torch_points3d.core.base_conv.message_passing as ref_mp

    class _Data(_Bag):
        pass

    ref_mp.Batch = _Data
    rec = {}

    def put_state(prefix, m):
        for k, v in m.state_dict().items():
            rec["%s/sd/%s" % (prefix, k)] = v.detach().clone()

    def put_after(prefix, m):
        for k, v in m.state_dict().items():
            if "running_" in k:
                rec["%s/after/%s" % (prefix, k)] = v.detach().clone()

    def put_grads(prefix, m):
        for k, p in m.named_parameters():
            if p.grad is not None:
                rec["%s/grad/%s" % (prefix, k)] = p.grad.detach().clone()

    # ---- (1) kernel level: RandlaKernel.forward / message / update on a fixed edge list, PyG tuple order
    g = torch.Generator().manual_seed(515)
    M, Nq, k, F = 900, 300, 16, 6
    pos_s = torch.rand(M, 3, generator=g) * 2 - 1
    batch_s = torch.sort(torch.randint(0, 2, (M,), generator=g))[0]
    qsel = torch.randint(0, M, (Nq,), generator=g)
    pos_q, batch_q = pos_s[qsel], batch_s[qsel]
    edges = knn_edges(pos_s, pos_q, k, batch_s, batch_q)
    edge_index = torch.stack([edges[1], edges[0]], 0)  # message_passing.py:50: [col (support), row (query)]
    rec.update({"k/pos_s": pos_s, "k/batch_s": batch_s, "k/qsel": qsel, "k/nbr": edges[1].view(Nq, k)})
    MessagePassing.tuple_order = "pyg"
    for tag, with_x in (("kx", True), ("kpos", False)):
        cin = F if with_x else 3
        torch.manual_seed(31 if with_x else 32)
        ker = mod.RandlaKernel(point_pos_nn=[10, 8, F], attention_nn=[cin + F, 8, cin + F], global_nn=[cin + F, 8, 16])
        ker.train()
        put_state(tag, ker)
        x = torch.randn(M, F, generator=g) if with_x else None
        cot = torch.randn(Nq, 16, generator=g)
        ker64 = copy.deepcopy(ker).double()
        xin = x.clone().requires_grad_(True) if with_x else None
        out = ker(xin, (pos_s, pos_q), edge_index)
        (out * cot).sum().backward()
        out64 = ker64(None if x is None else x.double(), (pos_s.double(), pos_q.double()), edge_index)
        put_after(tag, ker)
        put_grads(tag, ker)
        ker.eval()
        with torch.no_grad():
            out_eval = ker(x, (pos_s, pos_q), edge_index)
        rec.update({tag + "/out": out, tag + "/out64": out64.detach().numpy(), tag + "/out_eval": out_eval, tag + "/cot": cot})
        if with_x:
            rec.update({tag + "/x": x, tag + "/grad_x": xin.grad})

    # ---- (2) block level: the two RandLANetRes down modules of conf/models/segmentation/randlanet.yaml Randlanet_Res
    #          (ratio [1,1] then [0.5,0.5], FEAT = 6), call-site tuple order (see the docstring)
    MessagePassing.tuple_order = "callsite"
    N = 1200
    pos = torch.rand(N, 3, generator=g) * 2 - 1
    # ONE cloud: RandomSampler's indices are unsorted, so from the second convolution on `batch[idx]` of a multi-cloud
    # batch is unsorted, which torch_cluster's knn (and the oracle) do not accept as batch_x
    batch = torch.zeros(N, dtype=torch.long)
    x = torch.randn(N, F, generator=g)
    torch.manual_seed(41)
    b0 = mod.RandLANetRes(indim=3, outdim=32, ratio=[1, 1], point_pos_nn=[[10, 8, F], [10, 16, 16]],
                          attention_nn=[[2 * F, 8, 2 * F], [32, 64, 32]], down_conv_nn=[[2 * F, 8, 16], [32, 64, 32]],
                          index=0, nb_feature=F)
    b1 = mod.RandLANetRes(indim=32, outdim=128, ratio=[0.5, 0.5], point_pos_nn=[[10, 16, 32], [10, 32, 64]],
                          attention_nn=[[64, 128, 64], [128, 256, 128]], down_conv_nn=[[64, 64, 64], [128, 128, 128]],
                          index=1, nb_feature=F)
    net = torch.nn.ModuleDict({"b0": b0, "b1": b1})
    net.train()
    put_state("blk", net)
    net64 = copy.deepcopy(net).double()

    drawn = []
    real_randint = torch.randint

    def recording_randint(*a, **kw):
        out = real_randint(*a, **kw)
        drawn.append(out.clone())
        return out

    def run(model, xx, pp, record_draws):
        torch.manual_seed(43)  # RandomSampler is the only consumer of the global generator in forward
        if record_draws:
            torch.randint = recording_randint
        try:
            d0 = model["b0"](_Data(pos=pp, batch=batch, x=xx))
            d1 = model["b1"](d0)
        finally:
            torch.randint = real_randint
        return d0, d1

    xin = x.clone().requires_grad_(True)
    d0, d1 = run(net, xin, pos, True)
    cot = torch.randn(d1.x.shape, generator=g)
    (d1.x * cot).sum().backward()
    e0, e1 = run(net64, x.double(), pos.double(), False)
    put_after("blk", net)
    put_grads("blk", net)
    net.eval()
    with torch.no_grad():
        v0, v1 = run(net, x, pos, False)
    assert len(drawn) == 4 and torch.equal(d1.idx, drawn[3])
    rec.update({"blk/pos": pos, "blk/batch": batch, "blk/x": x, "blk/cot": cot, "blk/grad_x": xin.grad,
                "blk/b0_x": d0.x, "blk/b0_pos": d0.pos, "blk/b1_x": d1.x, "blk/b1_pos": d1.pos, "blk/b1_batch": d1.batch,
                "blk/b0_x64": e0.x.detach().numpy(), "blk/b1_x64": e1.x.detach().numpy(),
                "blk/b0_x_eval": v0.x, "blk/b1_x_eval": v1.x})
    for i, t in enumerate(drawn):
        rec["blk/draw%d" % i] = t
    path = os.path.join(HERE, "randla.npz")
    np.savez_compressed(path, **to_np(rec))
    print("wrote %s (%.1f KiB): kernel fixtures %d edges; blocks %d -> %d -> %d points; fp32-vs-fp64 distance of the "
          "reference pass: kernel %.2e, b0 %.2e, b1 %.2e" % (
              path, os.path.getsize(path) / 1024.0, edge_index.shape[1], N, d0.pos.shape[0], d1.pos.shape[0],
              float((rec["kx/out"].double() - torch.from_numpy(rec["kx/out64"])).abs().max()),
              float((d0.x.double() - e0.x).abs().max()), float((d1.x.double() - e1.x).abs().max())))


SMALL_CFG = dict(npoint=[160, 40], radii=[[0.35], [0.7]], nsample=[[24], [16]],
                 down_conv_nn=[[[4 + 3, 16, 16, 24]], [[24 + 3, 24, 24, 32]]], innermost=[32 + 3, 32, 48],
                 up_conv_nn=[[48 + 32, 32, 32], [32 + 24, 32, 24], [24 + 4, 24, 24, 24]],
                 normalize_xyz=[False, True], save_sampling_id=[False, False])
MSG_CFG = dict(npoint=[128, 32], radii=[[0.2, 0.4], [0.5, 0.9]], nsample=[[8, 16], [16, 24]],
               down_conv_nn=[[[3 + 3, 8, 12], [3 + 3, 8, 16]], [[12 + 16 + 3, 16, 24], [12 + 16 + 3, 16, 20]]],
               innermost=[24 + 20 + 3, 32, 48], up_conv_nn=[[48 + 44, 32, 32], [32 + 28, 24, 24], [24 + 3, 16, 16]],
               normalize_xyz=[False, False], save_sampling_id=[False, False])


def probe_seed(cfg, output_nc, pos, x, seed):
    torch.manual_seed(seed)
    net = build_reference_unet(cfg, output_nc)
    net.train()
    move_off_the_kink(net, pos, x, 1e-3)
    return probe_conditioning(net, pos, x)[1]


def make_conditioned_cases(small, msg):
    """LeakyReLU(0.01) fixtures whose forward pass stays clear of every kink (BatchNorm biases shifted, see
    move_off_the_kink) and of every max-pool arg-max switch (seed chosen so): gradients can be compared element-wise."""
    for name, cfg, feat, nc, shape, pseed in (("small_ssg_kinkfree", small, 4, 6, (3, 700), 1234),
                                              ("small_msg_kinkfree", msg, 3, 5, (2, 600), 99)):
        g = torch.Generator().manual_seed(pseed)
        pos = torch.rand(shape[0], shape[1], 3, generator=g) * 2 - 1
        feats = torch.randn(shape[0], shape[1], feat, generator=g)
        # the arg-max of a pooled group can switch only when its two largest values are closer than the forward error of
        # an implementation (a few 1e-7 here): take the initialisation (of eight) whose smallest such gap is largest
        best = max(range(100, 108), key=lambda sd_: probe_seed(cfg, nc, pos, feats, sd_))
        pre, gap = make_case(name, cfg, feat, nc, pos, feats, seed=best, store_weights=True, kink_delta=1e-3)
        if pre < 9e-4 or gap < 5e-6:
            raise RuntimeError("no seed gave a well-conditioned %s (%g, %g)" % (name, pre, gap))


def make_c3_case():
    """BASELINE config 3: pointnet2_charlesmsg (conf/models/segmentation/pointnet2.yaml:95-130) with the PointNet2_D
    head (models/segmentation/pointnet2.py:40-61,87-110; ShapeNet part segmentation: 16 categories one-hot, 50 part
    classes, N = 2048), B = 2 distinct clouds.  Weights are not stored (1.7 M parameters): they are the modules'
    default initialisation under torch.manual_seed(3) in the reference's construction order, pinned by checksums."""
    from torch_points3d_amd.pointnet2 import unet_config
    cfg = dict(unet_config("unet_3_ms", 3), nested=True, head="pointnet2_d", mlp_cls=[128, 128], dropout=0.5,
               num_categories=16)
    g = torch.Generator().manual_seed(2048)
    pos = torch.rand(2, 2048, 3, generator=g) * 2 - 1
    feats = torch.randn(2, 2048, 3, generator=g)
    category = torch.randint(0, 16, (2, 1), generator=g).repeat(1, 2048)  # one object category per cloud
    make_case("c3_charlesmsg", cfg, 3, 50, pos, feats, seed=3, store_weights=False, category=category, stage_grads=False,
              variant_cap=16384, sub_out=4)


def main():
    install_stubs()
    if sys.argv[1:] == ["c3"]:
        return make_c3_case()
    if sys.argv[1:] == ["conditioned"]:
        return make_conditioned_cases(SMALL_CFG, MSG_CFG)
    if sys.argv[1:] == ["rsconv"]:  # only this fixture (the others are unchanged)
        return make_rsconv_case()
    if sys.argv[1:] == ["randla"]:
        return make_randla_case()
    if sys.argv[1:] == ["kpconv_blocks"]:
        make_kpconv_blocks_case()
        return make_kpconv_blocks_case("kpconv_blocks_slope1", slope=1.0)
    make_kpconv_case()
    from torch_points3d_amd.pointnet2 import unet_config

    # (1) BASELINE config 1: examples/pointnet2_segmentation_forward.py:5-19 -- randn cloud duplicated to B=2,
    #     FEAT=5, 10 classes, unet_3_ss, weights from torch.manual_seed(0).
    torch.manual_seed(0)
    pos = torch.randn((1024, 3)).unsqueeze(0)
    feats = torch.randn((1024, 5)).unsqueeze(0)
    pos, feats = torch.cat([pos, pos], 0), torch.cat([feats, feats], 0)
    make_case("c1_example", unet_config("unet_3_ss", 5), 5, 10, pos, feats, seed=0, store_weights=False, dup_batch=True,
              stage_grads=False, variant_cap=16384)

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
    make_case("small_ssg_tanh", small, 4, 6, pos, feats, seed=7, store_weights=True, activation=torch.nn.Tanh(),
              stage_grads=False, variants=False)

    # (2c) LeakyReLU(negative_slope=1.0) == identity: still a LeakyReLU (so the fused channel-last kernels run it)
    #      but kink-free, so the gradients of the fused path can be compared element-wise as well.
    make_case("small_ssg_slope1", small, 4, 6, pos, feats, seed=7, store_weights=True,
              activation=torch.nn.LeakyReLU(negative_slope=1.0), stage_grads=False, variants=False)

    # (3) multi-scale grouping (unet_3_ms.yaml layout, narrow) on distinct clouds.
    g = torch.Generator().manual_seed(99)
    pos = torch.rand(2, 600, 3, generator=g) * 2 - 1
    feats = torch.randn(2, 600, 3, generator=g)
    msg = dict(npoint=[128, 32], radii=[[0.2, 0.4], [0.5, 0.9]], nsample=[[8, 16], [16, 24]],
               down_conv_nn=[[[3 + 3, 8, 12], [3 + 3, 8, 16]], [[12 + 16 + 3, 16, 24], [12 + 16 + 3, 16, 20]]],
               innermost=[24 + 20 + 3, 32, 48], up_conv_nn=[[48 + 44, 32, 32], [32 + 28, 24, 24], [24 + 3, 16, 16]],
               normalize_xyz=[False, False], save_sampling_id=[False, False])
    make_case("small_msg", msg, 3, 5, pos, feats, seed=11, store_weights=True)
    make_conditioned_cases(small, msg)
    make_c3_case()

    make_rsconv_case()
    make_randla_case()

    # (4) KPConv blocks + FPModule_PD through the reference's own classes (last: it replaces further modules by stubs)
    make_kpconv_blocks_case()
    make_kpconv_blocks_case("kpconv_blocks_slope1", slope=1.0)
    make_grid_sampling_case()
    make_unet4_config_case()


if __name__ == "__main__":
    main()
