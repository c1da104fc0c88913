"""The host-side mirror (torch_points3d_amd.dense / .pointnet2) driven by the CPU oracle must reproduce the
tensors the REFERENCE's own modules produced (tests/golden/*.npz, written by tests/golden/make_golden.py).
CPU only: pins module order, channel order, state_dict keys and train-mode BatchNorm of the mirror."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from torch_points3d_amd.dense import Data
from torch_points3d_amd.pointnet2 import PointNet2Unet, unet_config

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
}
ACTIVATION = {"small_ssg_tanh": torch.nn.Tanh, "small_ssg_slope1": lambda: torch.nn.LeakyReLU(negative_slope=1.0)}


def build_from_golden(g, cfg, kernels, device="cpu", activation=None, fused=True):
    """Mirror model carrying exactly the reference modules' weights."""
    feat, out_nc = [int(v) for v in g["meta_feat_outnc"]]
    torch.manual_seed(int(g["meta_seed"][0]))
    net = PointNet2Unet(feat, output_nc=out_nc, config=cfg, kernels=kernels, activation=activation, fused=fused)
    stored = {k[len("state/"):]: v for k, v in g.items() if k.startswith("state/")}
    if stored:
        net.load_state_dict(stored, strict=True)
    sd = net.state_dict()
    cks = {k[len("cksum/"):]: v for k, v in g.items() if k.startswith("cksum/")}
    assert set(cks) == set(sd), "state_dict keys differ from the reference modules'"
    for k, v in sd.items():
        got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
        np.testing.assert_allclose(got, cks[k], rtol=0, atol=0, err_msg="weights differ at " + k)
    return net.to(device).train()


def run_stages(net, pos, x):
    """Forward with per-stage capture, same names as make_golden.run_reference_unet."""
    rec = {}
    hooks = []
    for i, m in enumerate(net.down_modules):
        hooks.append(m.register_forward_hook(
            lambda mod, inp, out, i=i: rec.update({"down%d_x" % i: out.x, "down%d_pos" % i: out.pos})))
    hooks.append(net.inner_modules[0].register_forward_hook(lambda mod, inp, out: rec.update({"inner_x": out.x})))
    for i, m in enumerate(net.up_modules):
        hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: rec.update({"up%d_x" % i: out.x})))
    out = net(Data(pos=pos, x=x))
    rec["out_x"] = out.x
    for h in hooks:
        h.remove()
    return out, rec


@pytest.mark.parametrize("name", sorted(CASES))
def test_mirror_reproduces_reference_modules(oracle, name):
    g = load_golden(name)
    net = build_from_golden(g, CASES[name](), oracle, activation=ACTIVATION.get(name, lambda: None)())
    x_in = g["x"].clone().requires_grad_(True)
    out, rec = run_stages(net, g["pos"], x_in)
    for k, v in rec.items():
        # same PyTorch CPU ops in the same order as the reference modules: the tolerance only covers oneDNN
        # picking a different reduction split for another thread count
        torch.testing.assert_close(v.detach(), g[k], rtol=1e-4, atol=1e-5, msg=lambda m, k=k: k + ": " + m)
    bn = net.down_modules[0].mlps[0][0][1]
    torch.testing.assert_close(bn.running_mean, g["bn_after/first_running_mean"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, g["bn_after/first_running_var"], rtol=1e-5, atol=1e-6)
    (out.x * g["cotangent"]).sum().backward()
    torch.testing.assert_close(x_in.grad, g["grad_x_in"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(net.down_modules[0].mlps[0][0][0].weight.grad, g["grad_first_conv"], rtol=1e-4,
                               atol=1e-5)
    torch.testing.assert_close(net.up_modules[-1].nn[0][0].weight.grad, g["grad_last_fp_conv"], rtol=1e-4,
                               atol=1e-5)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden_indices(oracle, name):
    g = load_golden(name)
    cfg = CASES[name]()
    cur = g["pos"]
    for i in range(len(cfg["npoint"])):
        fps = oracle.furthest_point_sample(cur, cfg["npoint"][i])
        assert torch.equal(fps, g["fps%d" % i])
        new = cur.gather(1, fps.unsqueeze(-1).repeat(1, 1, 3))
        for s, (r, ns) in enumerate(zip(cfg["radii"][i], cfg["nsample"][i])):
            idx, d2 = oracle.ball_query(r, ns, cur, new)
            assert torch.equal(idx, g["ball%d_%d_idx" % (i, s)])
            assert torch.equal(d2, g["ball%d_%d_d2" % (i, s)])
        cur = new
