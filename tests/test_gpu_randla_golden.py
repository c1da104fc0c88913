"""HIP path of the RandLA-Net modules (csrc/randla.hip, gemm_skinny / rows kernels, exact grid kNN) against the fixture
the REFERENCE's own RandlaKernel / DilatedResidualBlock / RandLANetRes produced (tests/golden/randla.npz, written by
tests/golden/make_golden.py `make_randla_case`; see its docstring for what stands in for torch_geometric).

Bars (floating point, written here): kernel-level and stage-level outputs rtol = atol = 1e-5, relaxed to twice the
reference pass's own distance to its float64 evaluation where that is larger; the two chained blocks by their distance
to the float64 evaluation (<= 4x max / 2x RMS of the reference pass's own); eval mode 1e-5 * scale.  Neighbour tables:
torch.equal with the table the reference's edges were built from (oracle kNN; order unpinned in the reference)."""
import pytest
import torch

from conftest import load_golden
from randla_golden_util import F, bound, build_blocks, build_kernel, replay_draws

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def gold():
    return load_golden("randla")


def _dist64(t, ref64):
    d = t.detach().double().cpu() - torch.as_tensor(ref64)
    return float(d.abs().max()), float(d.pow(2).mean().sqrt())


def _check_param_grads(module, grads, rel_tol):
    assert grads
    for name, p in module.named_parameters():
        if name not in grads:
            continue
        want = grads[name].to(DEV)
        if name.endswith(".0.bias"):  # Linear bias under train-mode BatchNorm: analytically zero
            wn = float(grads[name[:-4] + "weight"].norm())
            assert p.grad is None or float(p.grad.norm()) < 1e-4 * wn + 1e-6, name
            continue
        rel = float((p.grad - want).norm() / (want.norm() + 1e-30))
        assert rel < rel_tol, (name, rel)


@pytest.mark.parametrize("tag,with_x", [("kx", True), ("kpos", False)])
def test_randla_kernel_matches_reference(hip, gold, tag, with_x):
    ker = build_kernel(gold, tag, with_x, DEV).train()
    pos_s = gold["k/pos_s"].to(DEV)
    qsel = gold["k/qsel"].to(DEV)
    pos_q = pos_s[qsel]
    batch_s = gold["k/batch_s"].to(DEV)
    # the kernel's neighbour table: the HIP kNN must reproduce the table the reference's edge list was built from
    nbr, _ = hip.knn(16, pos_s, pos_q, batch_s, batch_s[qsel])
    assert torch.equal(nbr.cpu(), gold["k/nbr"])
    x = gold[tag + "/x"].to(DEV).requires_grad_(True) if with_x else None
    out = ker(x, (pos_q, pos_s), nbr)
    tol = bound(gold[tag + "/out"], gold[tag + "/out64"])
    torch.testing.assert_close(out.cpu(), gold[tag + "/out"], rtol=1e-5, atol=tol)
    (out * gold[tag + "/cot"].to(DEV)).sum().backward()
    grads = {k[len(tag) + 6:]: v for k, v in gold.items() if k.startswith(tag + "/grad/")}
    _check_param_grads(ker, grads, 2e-3)
    if with_x:
        want = gold[tag + "/grad_x"].to(DEV)
        assert float((x.grad - want).norm() / want.norm()) < 2e-3
    after = {k[len(tag) + 7:]: v for k, v in gold.items() if k.startswith(tag + "/after/")}
    sd = ker.state_dict()
    for name, v in after.items():
        torch.testing.assert_close(sd[name].cpu(), v, rtol=1e-4, atol=1e-5, msg=name)
    ker.eval()
    with torch.no_grad():
        ev = ker(None if x is None else x.detach(), (pos_q, pos_s), nbr)
    torch.testing.assert_close(ev.cpu(), gold[tag + "/out_eval"], rtol=1e-5, atol=1e-5)


def _run(net, gold, replay, grad):
    from torch_points3d_amd.kpconv_blocks import PDData
    replay.i = 0
    x = gold["blk/x"].to(DEV).clone().requires_grad_(grad)
    d0 = net["b0"](PDData(pos=gold["blk/pos"].to(DEV), batch=gold["blk/batch"].to(DEV), x=x))
    return x, d0, net["b1"](d0)


def test_randlanet_res_blocks_match_reference(hip, gold):
    from torch_points3d_amd.kpconv_blocks import PDData
    net = build_blocks(gold, DEV).train()
    replay = replay_draws(net, gold, DEV)
    x, d0, d1 = _run(net, gold, replay, True)
    assert torch.equal(d0.pos.cpu(), gold["blk/b0_pos"]) and torch.equal(d1.pos.cpu(), gold["blk/b1_pos"])
    assert torch.equal(d1.idx.cpu(), gold["blk/draw3"])
    # first block: its inputs are the fixture's
    torch.testing.assert_close(d0.x.cpu(), gold["blk/b0_x"], rtol=1e-5, atol=bound(gold["blk/b0_x"], gold["blk/b0_x64"]))
    # both blocks chained: distance to the float64 evaluation against the reference pass's own
    own_max, own_rms = _dist64(gold["blk/b1_x"], gold["blk/b1_x64"])
    got_max, got_rms = _dist64(d1.x, gold["blk/b1_x64"])
    assert got_max <= 4 * own_max and got_rms <= 2 * own_rms, (got_max, own_max, got_rms, own_rms)
    (d1.x * gold["blk/cot"].to(DEV)).sum().backward()
    grads = {k[len("blk/grad/"):]: v for k, v in gold.items() if k.startswith("blk/grad/")}
    assert len(grads) > 40
    _check_param_grads(net, grads, 5e-3)
    want = gold["blk/grad_x"].to(DEV)
    assert float((x.grad - want).norm() / want.norm()) < 5e-3
    # second block on the fixture's first-block output (teacher forcing)
    forced = build_blocks(gold, DEV).train()
    replay_draws(forced, gold, DEV).i = 2
    t1 = forced["b1"](PDData(pos=gold["blk/b0_pos"].to(DEV), batch=gold["blk/batch"].to(DEV), x=gold["blk/b0_x"].to(DEV)))
    torch.testing.assert_close(t1.x.cpu(), gold["blk/b1_x"], rtol=1e-5, atol=bound(gold["blk/b1_x"], gold["blk/b1_x64"]))
    # running statistics, then the eval-mode chain (gemm_skinny eval epilogue, cached statistics)
    after = {k[len("blk/after/"):]: v for k, v in gold.items() if k.startswith("blk/after/")}
    sd = net.state_dict()
    for name, v in after.items():
        torch.testing.assert_close(sd[name].cpu(), v, rtol=1e-4, atol=1e-5, msg=name)
    net.eval()
    with torch.no_grad():
        _, v0, v1 = _run(net, gold, replay, False)
    s0, s1 = float(gold["blk/b0_x_eval"].abs().max()), float(gold["blk/b1_x_eval"].abs().max())
    torch.testing.assert_close(v0.x.cpu(), gold["blk/b0_x_eval"], rtol=1e-5, atol=1e-5 * max(1.0, s0))
    torch.testing.assert_close(v1.x.cpu(), gold["blk/b1_x_eval"], rtol=1e-5, atol=1e-5 * max(1.0, s1))
