"""RandLA-Net local feature aggregation (csrc/randla.hip): relative position rows and attentive pooling against plain
torch evaluations of modules/RandLANet/modules.py:36-52, and the fused RandlaKernel path against the unfused one.
Floating point: tolerances are written at each check (fp32, 1e-5 relative unless stated)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _table(Nq, M, k, seed, missing=False):
    g = torch.Generator().manual_seed(seed)
    nbr = torch.randint(0, M, (Nq, k), generator=g)
    if missing:
        nbr[::7, k // 2:] = -1  # clouds smaller than k: trailing slots empty
    return nbr


@pytest.mark.parametrize("Nq,M,k", [(1000, 3000, 16), (257, 50, 5), (1, 4, 1)])
def test_relative_position_rows(hip, Nq, M, k):
    from torch_points3d_amd.randla import relative_position_rows
    g = torch.Generator().manual_seed(Nq)
    pos_s = torch.rand(M, 3, generator=g) * 4 - 2
    pos_q = torch.rand(Nq, 3, generator=g) * 4 - 2
    nbr = _table(Nq, M, k, 1, missing=k > 1)
    rows = relative_position_rows(pos_q.to(DEV), pos_s.to(DEV), nbr.to(DEV)).cpu()
    assert rows.shape == (Nq * k, 12)
    j = nbr.reshape(-1)
    ok = j >= 0
    pos_i, pos_j = pos_q.repeat_interleave(k, 0), pos_s[j.clamp(min=0)]
    want = torch.cat([pos_i, pos_j, pos_i - pos_j], 1)
    assert torch.equal(rows[ok, :9], want[ok])  # copies and one subtraction: exact
    torch.testing.assert_close(rows[ok, 9], (pos_i - pos_j).norm(dim=1)[ok], rtol=1e-6, atol=1e-7)
    assert torch.count_nonzero(rows[:, 10:]) == 0 and torch.count_nonzero(rows[~ok]) == 0


@pytest.mark.parametrize("C,k,missing", [(6, 16, False), (16, 16, True), (32, 16, False), (64, 16, True), (100, 7, False),
                                         (128, 16, True), (200, 3, False), (3, 1, False), (256, 2, True)])
def test_attentive_pool_fwd_bwd(hip, C, k, missing):
    from torch_points3d_amd.randla import attentive_pool
    Nq = 523
    gen = torch.Generator().manual_seed(C * 31 + k)
    ld = (C + 3) // 4 * 4
    g = (torch.randn(Nq * k, C, generator=gen) * 3).to(DEV).requires_grad_(True)
    f = torch.randn(Nq * k, ld, generator=gen).to(DEV).requires_grad_(True)
    nbr = _table(Nq, 1000, k, 5, missing).to(DEV)
    cot = torch.randn(Nq, C, generator=gen).to(DEV)
    out = attentive_pool(g, f, nbr, C)
    (out * cot).sum().backward()
    # fp64 evaluation of softmax(g, -1) * f summed over the k edges of each query
    g64, f64 = g.detach().double().requires_grad_(True), f.detach().double().requires_grad_(True)
    keep = (nbr.reshape(-1, 1) >= 0).double()
    ref = (torch.softmax(g64, -1) * f64[:, :C] * keep).reshape(Nq, k, C).sum(1)
    (ref * cot.double()).sum().backward()
    torch.testing.assert_close(out.detach().double(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(g.grad.double(), g64.grad, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(f.grad.double(), f64.grad, rtol=1e-5, atol=1e-5)
    assert torch.count_nonzero(f.grad[:, C:]) == 0
    out2 = attentive_pool(g, f, nbr, C)
    assert torch.equal(out, out2)  # fixed summation order


def test_attentive_pool_rejects_wide_rows(hip):
    from torch_points3d_amd import _lib
    from torch_points3d_amd.randla import attentive_pool
    g = torch.zeros(8, 300, device=DEV)
    with pytest.raises(_lib.Tp3dError):
        attentive_pool(g, g.clone(), torch.zeros(4, 2, dtype=torch.long, device=DEV), 300)


@pytest.mark.parametrize("with_x", [True, False])
def test_randla_kernel_fused_matches_unfused(hip, with_x):
    """fused=True (HIP row kernels between the MLPs) vs fused=False (the reference's chain of torch ops on the GPU):
    forward, input gradient and weight gradients.  Tanh instead of the LeakyReLU kink keeps gradients comparable."""
    import copy
    from torch_points3d_amd.randla import RandlaKernel
    torch.manual_seed(3)
    M, Nq, k, F = 6000, 1500, 16, 8
    pos_s = torch.rand(M, 3, device=DEV)
    pos_q = pos_s[torch.randint(0, M, (Nq,), device=DEV)]
    nbr = hip.knn(k, pos_s, pos_q)[0]
    cin = F if with_x else 3
    a = RandlaKernel(point_pos_nn=[10, 8, F], attention_nn=[cin + F, 8, cin + F], global_nn=[cin + F, 8, 16]).to(DEV)
    b = copy.deepcopy(a)
    b.fused = False
    x1 = torch.randn(M, F, device=DEV).requires_grad_(True) if with_x else None
    x2 = x1.detach().clone().requires_grad_(True) if with_x else None
    cot = torch.randn(Nq, 16, device=DEV)
    o1 = a(x1, (pos_q, pos_s), nbr)
    o2 = b(x2, (pos_q, pos_s), nbr)
    torch.testing.assert_close(o1, o2, rtol=1e-4, atol=1e-4)
    (o1 * cot).sum().backward()
    (o2 * cot).sum().backward()
    scale = lambda t: max(1.0, float(t.abs().max()))  # noqa: E731
    if with_x:
        torch.testing.assert_close(x1.grad, x2.grad, rtol=1e-3, atol=1e-4 * scale(x2.grad))
    for (n, p1), (_, p2) in zip(a.named_parameters(), b.named_parameters()):
        if p2.grad is None:
            assert p1.grad is None, n
            continue
        # LeakyReLU(0.2) kinks and train-mode BatchNorm: bound the relative L2 error
        assert float((p1.grad - p2.grad).norm()) <= 2e-2 * float(p2.grad.norm()) + 1e-5, n


@pytest.mark.parametrize("M,N,K", [(100000, 8, 12), (70001, 3, 8), (40000, 6, 8), (33000, 16, 8), (50000, 32, 32),
                                   (1000, 5, 7), (65536, 8, 10), (3, 1, 1)])
def test_gemm_skinny(hip, M, N, K):
    """row-per-lane GEMM of the edge MLPs vs fp64 (and exact on small-integer data: fp32 sums of integers)"""
    from torch_points3d_amd import fused
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = torch.randn(N, K, generator=g).to(DEV)
    got = fused.gemm_skinny(A, W)
    ref = A.double() @ W.double().t()
    assert got.shape == (M, N)
    torch.testing.assert_close(got.double(), ref, rtol=1e-5, atol=1e-5)
    Ai = torch.randint(-4, 5, (M, K), generator=g).float().to(DEV)
    Wi = torch.randint(-4, 5, (N, K), generator=g).float().to(DEV)
    assert torch.equal(fused.gemm_skinny(Ai, Wi), (Ai.double() @ Wi.double().t()).float())
    # a strided view is made contiguous by the wrapper
    wide = torch.randn(M, K + 5, generator=g).to(DEV)
    torch.testing.assert_close(fused.gemm_skinny(wide[:, 2:2 + K], W).double(), wide[:, 2:2 + K].double() @ W.double().t(),
                               rtol=1e-5, atol=1e-5)


def test_eval_mode_edge_mlp_runs_in_one_pass(hip):
    """Linear -> BatchNorm (running statistics) -> LeakyReLU of a few-channel edge MLP layer under torch.no_grad():
    tp3d_gemm_skinny_bnact_f32 applies affine and activation in the GEMM's epilogue; same values as the plain modules"""
    from torch_points3d_amd import fused
    from torch_points3d_amd.partial_dense import MLP
    torch.manual_seed(1)
    mlp = MLP([12, 8, 6], bn_momentum=0.1).to(DEV)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_()
                m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_()
    mlp.eval()
    x = torch.randn(70000, 12, device=DEV)
    with torch.no_grad():
        got = fused.rows_mlp(mlp, x)
        want = mlp(x)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # with a gradient wanted the two-kernel path (which keeps the pre-activation for the backward pass) serves it
    xg = x.clone().requires_grad_(True)
    fused.rows_mlp(mlp, xg).sum().backward()
    xr = x.clone().requires_grad_(True)
    mlp(xr).sum().backward()
    torch.testing.assert_close(xg.grad, xr.grad, rtol=1e-4, atol=1e-5)


def test_dilated_residual_block_and_edge_list(hip):
    """RandLANetRes (conf/models/segmentation/randlanet.yaml Randlanet_Res, first down module) mirrors the reference's
    module tree (state_dict keys of DilatedResidualBlock / BaseResnetBlock) and runs on the fused row kernels like the
    unfused chain; RandlaConv(edge_list=True) exposes the reference's edge_index = [support index, query index]."""
    import copy
    from torch_points3d_amd.kpconv_blocks import PDData
    from torch_points3d_amd.randla import RandLANetRes, RandlaConv
    torch.manual_seed(2)
    F = 6
    block = RandLANetRes(indim=3, outdim=32, ratio=[1, 1], point_pos_nn=[[10, 8, F], [10, 16, 16]],
                         attention_nn=[[2 * F, 8, 2 * F], [32, 64, 32]], down_conv_nn=[[2 * F, 8, 16], [32, 64, 32]],
                         index=0, nb_feature=F).to(DEV).eval()
    keys = set(block.state_dict())
    for prefix in ("_conv.features_downsample_nn.0.0.weight", "_conv.features_upsample_nn.0.1.batch_norm.running_mean"
                   if False else "_conv.features_upsample_nn.0.0.weight", "_conv.shortcut_feature_resize_nn.0.0.weight",
                   "_conv.conv1._conv.point_pos_nn.0.0.weight", "_conv.conv2._conv.global_nn.1.0.weight"):
        assert prefix in keys, prefix
    n = 3000
    pos = torch.rand(n, 3, device=DEV)
    batch = torch.zeros(n, dtype=torch.long, device=DEV)
    x = torch.randn(n, F, device=DEV)
    twin = copy.deepcopy(block)
    for m in twin.modules():
        if hasattr(m, "fused"):
            m.fused = False
    torch.manual_seed(9)
    with torch.no_grad():
        a = block(PDData(pos=pos, batch=batch, x=x))
    torch.manual_seed(9)  # same random subsampling in both
    with torch.no_grad():
        b = twin(PDData(pos=pos, batch=batch, x=x))
    assert a.x.shape == (n, 32) and torch.equal(a.idx, b.idx)
    torch.testing.assert_close(a.x, b.x, rtol=1e-4, atol=1e-4)
    conv = RandlaConv(0.25, 16, point_pos_nn=[10, 8, F], attention_nn=[2 * F, 8, 2 * F], down_conv_nn=[2 * F, 8, 16],
                      edge_list=True).to(DEV).eval()
    with torch.no_grad():
        out = conv(PDData(pos=pos, batch=batch, x=x))
    ei = out.edge_index
    assert ei.shape == (2, out.pos.shape[0] * 16)
    assert torch.equal(ei[1], torch.arange(out.pos.shape[0], device=DEV).repeat_interleave(16))  # query-major
    assert torch.equal(ei[0].view(-1, 16), out.neighbors)
    d = (pos[ei[0]] - out.pos[ei[1]]).pow(2).sum(1).view(-1, 16)
    assert bool((d[:, 1:] >= d[:, :-1] - 1e-7).all())  # closest first
