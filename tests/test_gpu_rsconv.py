"""Free drop-in beneficiaries of the kernel API: the RSConv dense modules and VoteNet's
proposal sampling reach the same furthest_point_sample / ball_query / grouping_operation entry points with other
shapes and another channel order.  HIP path vs the reference modules' recorded tensors and vs the oracle."""
import pytest
import torch

from conftest import load_golden
from torch_points3d_amd.dense import Data

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_rsconv_goldens_on_gpu(hip):
    from test_rsconv_cpu import build_levels
    g = load_golden("rsconv_dense")
    l0, l1 = build_levels(g, None, device=DEV)  # kernels=None -> the HIP product path
    pos = g["pos"].to(DEV)
    # kernel boundary first: bit-exact indices
    fps0 = hip.furthest_point_sample(pos, 128)
    assert torch.equal(fps0.cpu().long(), g["fps0"])
    new0 = pos.gather(1, fps0.long().unsqueeze(-1).repeat(1, 1, 3))
    for s, (r, ns) in enumerate(zip([0.3, 0.45], [12, 20])):
        assert torch.equal(hip.ball_query(r, ns, pos, new0)[0].cpu(), g["ball0_%d_idx" % s])
    fps1 = hip.furthest_point_sample(new0, 32)
    assert torch.equal(fps1.cpu().long(), g["fps1"])
    new1 = new0.gather(1, fps1.long().unsqueeze(-1).repeat(1, 1, 3))
    for s, (r, ns) in enumerate(zip([0.6, 0.9], [16, 24])):
        assert torch.equal(hip.ball_query(r, ns, new0, new1)[0].cpu(), g["ball1_%d_idx" % s])

    x_in = g["x"].to(DEV).requires_grad_(True)
    d0 = l0(Data(pos=pos, x=x_in.transpose(1, 2).contiguous()))
    d1 = l1(d0)
    assert torch.equal(d0.pos.cpu(), g["l0_pos"]) and torch.equal(d1.pos.cpu(), g["l1_pos"])

    def own(key):  # the reference pass's own distance to its float64 evaluation
        d = g[key].double() - torch.as_tensor(g["f64/" + key])
        return float(d.abs().max()), float(d.pow(2).mean().sqrt())

    def dist64(t, key):
        d = t.detach().double().cpu() - torch.as_tensor(g["f64/" + key])
        return float(d.abs().max()), float(d.pow(2).mean().sqrt())

    # first level (its inputs are the fixture's): rtol = atol = 1e-5, relaxed to twice the reference pass's own fp64
    # distance where that is larger; second level chained behind it: by its distance to the float64 evaluation
    torch.testing.assert_close(d0.x.detach().cpu(), g["l0_x"], rtol=1e-5, atol=max(1e-5, 2 * own("l0_x")[0]))
    got_max, got_rms = dist64(d1.x, "l1_x")
    assert got_max <= 4 * own("l1_x")[0] and got_rms <= 2 * own("l1_x")[1], (got_max, got_rms, own("l1_x"))
    (d1.x * g["cotangent"].to(DEV)).sum().backward()
    # ReLU kinks + train-mode BN: one flipped mask couples to the whole batch, so only the relative L2 error is
    # bounded here (the scatter-add backward kernels are pinned exactly in test_gpu_parity.py)
    for got, key in ((x_in.grad, "grad_x_in"), (l0._mapper.nn["mlp_msg"][0][0].weight.grad, "grad_l0_msg_conv"),
                     (l1.mlp_out[0].weight.grad, "grad_l1_raise_conv")):
        assert float((got.cpu() - g[key]).norm() / g[key].norm()) < 0.1, key
    # running statistics after the training pass
    for name, m in (("l0", l0), ("l1", l1)):
        sd = m.state_dict()
        pre = "after/%s/" % name
        for k, v in g.items():
            if k.startswith(pre):
                torch.testing.assert_close(sd[k[len(pre):]].cpu(), v, rtol=1e-4, atol=1e-6, msg=k)
    # second level on the fixture's first-level output (teacher forcing, fresh statistics)
    _, t1 = build_levels(g, None, device=DEV)
    f1 = t1(Data(pos=g["l0_pos"].to(DEV), x=g["l0_x"].to(DEV)))
    torch.testing.assert_close(f1.x.detach().cpu(), g["l1_x"], rtol=1e-5, atol=max(1e-5, 2 * own("l1_x")[0]))
    # eval mode, both levels chained
    l0.eval()
    l1.eval()
    with torch.no_grad():
        v0 = l0(Data(pos=pos, x=g["x"].to(DEV).transpose(1, 2).contiguous()))
        v1 = l1(v0)
    torch.testing.assert_close(v0.x.cpu(), g["eval/l0_x"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(v1.x.cpu(), g["eval/l1_x"], rtol=1e-5, atol=1e-5)


def test_votenet_proposal_sampling_shape(hip, oracle):
    """VoteNet's ProposalModule (modules/VoteNet/proposal_module.py:57): furthest_point_sample(seed_pos (B,1024,3),
    num_proposal=256) on vote-shifted seeds (clustered around object centres, with exact duplicates)."""
    g = torch.Generator().manual_seed(5)
    centres = torch.rand(4, 12, 3, generator=g) * 6 - 3
    which = torch.randint(0, 12, (4, 1024), generator=g)
    seed_pos = centres.gather(1, which.unsqueeze(-1).repeat(1, 1, 3)) + 0.05 * torch.randn(4, 1024, 3, generator=g)
    seed_pos[:, 512:520] = seed_pos[:, 0:8]  # duplicated votes
    want = oracle.furthest_point_sample(seed_pos, 256)
    got = hip.furthest_point_sample(seed_pos.to(DEV), 256)
    assert torch.equal(got.cpu().long(), want.long())
