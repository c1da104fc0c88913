"""Shared helpers of the golden-fixture tests (CPU mirror tests and GPU parity tests).

A fixture (tests/golden/<name>.npz, written by tests/golden/make_golden.py from the REFERENCE's own modules) holds, per
network stage, the train-mode fp32 output, an fp64 evaluation of the same pass (`f64/`), the eval-mode output (`eval/`),
and for the small nets the gradient that enters / leaves each stage (`gout/`, `gin/`) and every parameter gradient
(`pgrad/`).  The helpers below build the mirror with the fixture's weights and run it whole or stage by stage on the
fixture's own stage inputs ("teacher forcing": the error of a stage is then the error of that stage alone)."""
import json
import os

import numpy as np
import torch

from torch_points3d_amd.dense import Data
from torch_points3d_amd.pointnet2 import PointNet2_D, PointNet2Unet, unet_config

SMALL_SSG = dict(npoint=[160, 40], radii=[[0.35], [0.7]], nsample=[[24], [16]],
                 down_conv_nn=[[[4 + 3, 16, 16, 24]], [[24 + 3, 24, 24, 32]]], innermost=[32 + 3, 32, 48],
                 up_conv_nn=[[48 + 32, 32, 32], [32 + 24, 32, 24], [24 + 4, 24, 24, 24]],
                 normalize_xyz=[False, True], save_sampling_id=[False, False])
SMALL_MSG = dict(npoint=[128, 32], radii=[[0.2, 0.4], [0.5, 0.9]], nsample=[[8, 16], [16, 24]],
                 down_conv_nn=[[[3 + 3, 8, 12], [3 + 3, 8, 16]], [[12 + 16 + 3, 16, 24], [12 + 16 + 3, 16, 20]]],
                 innermost=[24 + 20 + 3, 32, 48], up_conv_nn=[[48 + 44, 32, 32], [32 + 28, 24, 24], [24 + 3, 16, 16]],
                 normalize_xyz=[False, False], save_sampling_id=[False, False])

CASES = {
    "c1_example": lambda: unet_config("unet_3_ss", 5),
    "small_ssg": lambda: SMALL_SSG,
    "small_msg": lambda: SMALL_MSG,
    "small_ssg_tanh": lambda: SMALL_SSG,
    "small_ssg_slope1": lambda: SMALL_SSG,
    "small_ssg_kinkfree": lambda: SMALL_SSG,
    "small_msg_kinkfree": lambda: SMALL_MSG,
    "c3_charlesmsg": lambda: unet_config("unet_3_ms", 3),
}
ACTIVATION = {"small_ssg_tanh": torch.nn.Tanh, "small_ssg_slope1": lambda: torch.nn.LeakyReLU(negative_slope=1.0)}
UNET_CASES = [n for n in sorted(CASES) if n != "c3_charlesmsg"]
KINKFREE = ["small_ssg_kinkfree", "small_msg_kinkfree"]


def build_from_golden(g, name, kernels, device="cpu", fused=True):
    """Mirror model carrying exactly the reference modules' weights (stored, or re-created from the seed and pinned
    by the stored checksums)."""
    feat, out_nc = [int(v) for v in g["meta_feat_outnc"]]
    torch.manual_seed(int(g["meta_seed"][0]))
    if name == "c3_charlesmsg":
        net = PointNet2_D(feat, out_nc, config="pointnet2_charlesmsg", num_categories=16, kernels=kernels, fused=fused)
    else:
        net = PointNet2Unet(feat, output_nc=out_nc, config=CASES[name](), kernels=kernels,
                            activation=ACTIVATION.get(name, lambda: None)(), fused=fused)
    stored = {k[len("state/"):]: v for k, v in g.items() if k.startswith("state/")}
    if stored:
        net.load_state_dict(stored, strict=True)
    sd = net.state_dict()
    cks = {k[len("cksum/"):]: v for k, v in g.items() if k.startswith("cksum/")}
    assert set(cks) == set(sd), "state_dict keys differ from the reference modules'"
    for k, v in sd.items():
        assert np.array_equal(exact_checksum(v), np.asarray(cks[k])), "weights differ at " + k
    return net.to(device).train()


def exact_checksum(v):
    """as tests/golden/make_golden.py: order-independent integer checksums of the bit patterns"""
    b = v.detach().contiguous()
    b = b.float().view(torch.int32) if b.is_floating_point() else b
    b = b.reshape(-1).to(torch.int64)
    w = torch.arange(b.numel(), dtype=torch.int64) % 1021 + 1
    return np.array([int(b.sum()), int((b * w).sum())], dtype=np.int64)


def stage_lists(net):
    if isinstance(net, PointNet2_D):
        return net.stages()
    return list(net.down_modules), net.inner_modules[0], list(net.up_modules)


def head_output(net, x, g, dev):
    """(key, tensor) of the head stage for features x (B,C,N): out_x of the applications U-Net; fc0_x (the classifier's
    hidden layer, before its Dropout) of PointNet2_D."""
    if isinstance(net, PointNet2_D):
        rows = net.classifier_hidden(x, g["category"].to(dev))
        return "fc0_x", rows.view(x.shape[0], x.shape[2], -1).transpose(1, 2)
    return "out_x", net._head(x)


def cotangent(g):
    if "cotangent" in g:
        return g["cotangent"]
    shape = [int(v) for v in g["meta_cot_shape"]]
    return torch.randn(shape, generator=torch.Generator().manual_seed(int(g["meta_seed"][0]) + 1))


def run_stages(net, g, dev, x_in=None):
    """Whole forward pass with per-stage capture (keys as in make_golden.run_reference_unet)."""
    rec, hooks = {}, []
    downs, inner, ups = stage_lists(net)
    for i, m in enumerate(downs):
        hooks.append(m.register_forward_hook(
            lambda mod, inp, out, i=i: rec.update({"down%d_x" % i: out.x, "down%d_pos" % i: out.pos})))
    hooks.append(inner.register_forward_hook(lambda mod, inp, out: rec.update({"inner_x": out.x})))
    for i, m in enumerate(ups):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: rec.update({"up%d_x" % i: out.x})))
    pos = g["pos"].to(dev)
    x = g["x"].to(dev) if x_in is None else x_in
    if isinstance(net, PointNet2_D):
        feats = net.model(Data(pos=pos, x=x.transpose(1, 2) if net.fused and x.is_cuda else x.transpose(1, 2).contiguous())).x
        key, val = head_output(net, feats, g, dev)
        rec[key] = val
        if not net.training:
            B, n = pos.shape[0], pos.shape[1]
            rec["out_x"] = net.classify(feats, g["category"].to(dev)).view(B, n, -1).transpose(1, 2)
    else:
        rec["out_x"] = net(Data(pos=pos, x=x)).x
    for h in hooks:
        h.remove()
    return rec


def run_teacher_forced(net, g, dev, grads=False):
    """Every stage on the FIXTURE's inputs for that stage.  Returns {stage key: output}; with grads=True also
    {stage key: [input tensors that require grad]} for the per-stage backward checks."""
    downs, inner, ups = stage_lists(net)
    out, ins = {}, {}

    def leaf(t):
        t = t.to(dev)
        return t.clone().requires_grad_(True) if grads else t

    x0 = leaf(g["x"])  # (B, N, C) as the data loader hands it over
    fused = getattr(net, "fused", False) and x0.is_cuda
    cur = Data(pos=g["pos"].to(dev), x=x0.transpose(1, 2) if fused else x0.transpose(1, 2).contiguous())
    cur_leaf = x0
    skips = [(cur, cur_leaf)]
    for i, d in enumerate(downs):
        o = d(cur)
        out["down%d_x" % i] = o.x
        out["down%d_pos" % i] = o.pos
        ins["down%d_x" % i] = [cur_leaf]
        cur_leaf = leaf(g["down%d_x" % i])
        cur = Data(pos=g["down%d_pos" % i].to(dev), x=cur_leaf)
        skips.append((cur, cur_leaf))
    out["inner_x"] = inner(cur).x
    ins["inner_x"] = [cur_leaf]
    cur_leaf = leaf(g["inner_x"])
    cur = Data(pos=None, x=cur_leaf)
    for i, u in enumerate(ups):
        skip, skip_leaf = skips.pop()
        if grads:  # a tensor feeding two stages gets its own leaf per use (down path above, skip here)
            skip_leaf = skip_leaf.detach().clone().requires_grad_(True)
            sx = skip_leaf.transpose(1, 2) if skip_leaf.shape[-1] == g["x"].shape[-1] and skip_leaf.dim() == 3 and \
                skip_leaf.shape[1] == g["x"].shape[1] and i == len(ups) - 1 else skip_leaf
            if i == len(ups) - 1 and not fused:
                sx = sx.contiguous()
            skip = Data(pos=skip.pos, x=sx)
        out["up%d_x" % i] = u((cur, skip)).x
        ins["up%d_x" % i] = [cur_leaf, skip_leaf]
        cur_leaf = leaf(g["up%d_x" % i])
        cur = Data(pos=skip.pos, x=cur_leaf)
    key, val = head_output(net, cur.x, g, dev)
    out[key] = val
    ins[key] = [cur_leaf]
    return (out, ins) if grads else out


def variant(g, prefix, key, full):
    """(stored variant tensor, the same subset of `full`) for a key of the `eval/` or `f64/` families."""
    ref = g[prefix + key]
    s, first = [int(v) for v in g["meta_sub/" + prefix + key]]
    t = full
    if first:
        t = t[:1]
    if s > 1:
        t = t[..., ::s]
    return ref, t


def head_subsample(g, key, t):
    """fc0_x of the large fixture is stored on every meta_sub_out-th point."""
    if key in ("fc0_x", "out_x") and "meta_sub_out" in g:
        return t[..., ::int(g["meta_sub_out"][0])]
    return t


def load_after_state(net, g):
    """BatchNorm running statistics as they stood after the fixture's one training pass (for the eval-mode goldens)."""
    sd = net.state_dict()
    for k, v in g.items():
        if k.startswith("after/"):
            sd[k[len("after/"):]].copy_(torch.as_tensor(v))
    return net


REPORT = {}


def report(section, key, value):
    """Measured errors are collected and written to gpurun_out/parity_report.json (when that directory exists), so the
    numbers behind the tolerances are on record next to the test log."""
    REPORT.setdefault(section, {})[key] = value
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_report.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
