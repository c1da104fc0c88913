"""The host-side mirror (torch_points3d_amd.dense / .pointnet2) driven by the CPU oracle must reproduce the
tensors the REFERENCE's own modules produced (tests/golden/*.npz, written by tests/golden/make_golden.py).
CPU only: pins module order, channel order, state_dict keys, train- and eval-mode BatchNorm of the mirror, and checks
the staged ("teacher-forced") harness the GPU parity tests use against the same fixtures."""
import pytest
import torch

from conftest import load_golden
from golden_util import (ACTIVATION, CASES, KINKFREE, UNET_CASES, build_from_golden, cotangent, head_subsample,
                         load_after_state, run_stages, run_teacher_forced, stage_lists, variant)

__all__ = ["ACTIVATION", "CASES"]


def _first_conv(net):
    return stage_lists(net)[0][0].mlps[0][0][0]


def _last_fp_conv(net):
    return stage_lists(net)[2][-1].nn[0][0]


@pytest.mark.parametrize("name", sorted(CASES))
def test_mirror_reproduces_reference_modules(oracle, name):
    g = load_golden(name)
    net = build_from_golden(g, name, oracle)
    x_in = g["x"].clone().requires_grad_(True)
    rec = run_stages(net, g, "cpu", x_in=x_in)
    for k, v in rec.items():
        # same PyTorch CPU ops in the same order as the reference modules: the tolerance only covers oneDNN
        # picking a different reduction split for another thread count
        torch.testing.assert_close(head_subsample(g, k, v.detach()), g[k], rtol=1e-4, atol=1e-5,
                                   msg=lambda m, k=k: k + ": " + m)
    bn = stage_lists(net)[0][0].mlps[0][0][1]
    torch.testing.assert_close(bn.running_mean, g["bn_after/first_running_mean"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var, g["bn_after/first_running_var"], rtol=1e-5, atol=1e-6)
    target = rec["fc0_x"] if "fc0_x" in rec else rec["out_x"]
    (target * cotangent(g)).sum().backward()
    torch.testing.assert_close(x_in.grad, g["grad_x_in"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(_first_conv(net).weight.grad, g["grad_first_conv"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(_last_fp_conv(net).weight.grad, g["grad_last_fp_conv"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", [n for n in sorted(CASES) if n not in ACTIVATION])
def test_mirror_eval_mode_matches_reference(oracle, name):
    """running statistics after the fixture's training pass, then eval(): the mirror's eval path == the reference's"""
    g = load_golden(name)
    net = load_after_state(build_from_golden(g, name, oracle), g).eval()
    with torch.no_grad():
        rec = run_stages(net, g, "cpu")
    checked = 0
    for k, v in rec.items():
        if "eval/" + k in g:
            ref, got = variant(g, "eval/", k, v)
            torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5, msg=lambda m, k=k: k + ": " + m)
            checked += 1
    assert checked >= 7


@pytest.mark.parametrize("name", ["small_ssg", "small_msg", "c3_charlesmsg"] + KINKFREE)
def test_teacher_forced_stages_on_cpu(oracle, name):
    """each stage on the fixture's own inputs reproduces the fixture's output for that stage"""
    g = load_golden(name)
    net = build_from_golden(g, name, oracle)
    out = run_teacher_forced(net, g, "cpu")
    for k, v in out.items():
        torch.testing.assert_close(head_subsample(g, k, v.detach()), g[k], rtol=1e-5, atol=1e-5,
                                   msg=lambda m, k=k: k + ": " + m)


@pytest.mark.parametrize("name", ["small_ssg", "small_msg"] + KINKFREE)
def test_teacher_forced_stage_gradients_on_cpu(oracle, name):
    """per stage: gradient entering (gout/) -> gradients leaving towards the stage's inputs (gin/), element-wise"""
    g = load_golden(name)
    net = build_from_golden(g, name, oracle)
    out, ins = run_teacher_forced(net, g, "cpu", grads=True)
    checked = 0
    for key, inputs in ins.items():
        if "gout/" + key not in g:
            continue
        grads = torch.autograd.grad(out[key], inputs, grad_outputs=g["gout/" + key], allow_unused=True, retain_graph=True)
        for j, got in enumerate(grads):
            want = g.get("gin/%s/%d" % (key, j))
            if want is None:
                continue
            torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-6, msg=lambda m, k=key, j=j: "%s/%d: %s" % (k, j, m))
            checked += 1
    assert checked >= 8


@pytest.mark.parametrize("name", UNET_CASES)
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


@pytest.mark.parametrize("name", KINKFREE)
def test_conditioned_fixtures_are_conditioned(name):
    g = load_golden(name)
    assert float(g["meta_min_preact"][0]) >= 9e-4   # no LeakyReLU input within 1e-3 of the kink
    assert float(g["meta_min_pool_gap"][0]) >= 5e-6  # no pooled group whose two largest values nearly tie
