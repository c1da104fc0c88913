"""Channel-last fused kernels of the grouped-MLP aggregation (csrc/rows.hip via torch_points3d_amd.fused)
against a plain PyTorch fp32 evaluation of the reference's graph (Conv2d 1x1 -> BatchNorm2d(train) ->
LeakyReLU -> max_pool2d, modules/pointnet2/dense.py:36-75), forward and backward."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("B,N,npnt,ns,C,normalize", [(2, 300, 40, 16, 5, False), (3, 1000, 64, 64, 128, True),
                                                     (1, 50, 7, 3, 0, False), (2, 64, 9, 5, 1, True)])
def test_group_concat_fwd_bwd(B, N, npnt, ns, C, normalize):
    from torch_points3d_amd import fused
    g = torch.Generator().manual_seed(N + C)
    pos = torch.rand(B, N, 3, generator=g).to(DEV)
    new_pos = torch.rand(B, npnt, 3, generator=g).to(DEV)
    idx = torch.randint(0, N, (B, npnt, ns), generator=g).to(DEV)
    x = torch.randn(B, C, N, generator=g).to(DEV) if C else None
    r = 0.37
    # reference graph (channel-first), as PointNetMSGDown._prepare_features builds it
    xr = x.clone().requires_grad_(True) if C else None
    gp = pos.transpose(1, 2).contiguous().gather(2, idx.view(B, 1, -1).repeat(1, 3, 1)).view(B, 3, npnt, ns)
    gp = gp - new_pos.transpose(1, 2).unsqueeze(-1)
    if normalize:
        gp = gp / r
    ref = gp if not C else torch.cat([gp, xr.gather(2, idx.view(B, 1, -1).repeat(1, C, 1)).view(B, C, npnt, ns)], 1)
    xf = x.clone().requires_grad_(True) if C else None
    rows = fused.group_concat(pos, new_pos, None if not C else xf.transpose(1, 2), idx, r, normalize)
    ld = (C + 3 + 3) // 4 * 4
    assert rows.shape == (B * npnt * ns, ld)
    assert ld == C + 3 or bool((rows[:, C + 3:] == 0).all())  # zero padding columns
    got = rows.view(B, npnt, ns, ld)[..., : C + 3].permute(0, 3, 1, 2)
    if normalize:
        # the kernel performs the IEEE division the reference's CPU path performs; torch's GPU kernel multiplies
        # by a reciprocal for a scalar divisor, so this comparison is within one ulp rather than bitwise
        torch.testing.assert_close(got, ref.detach(), rtol=3e-7, atol=0)
        cpu = (pos.cpu().transpose(1, 2).contiguous().gather(2, idx.cpu().view(B, 1, -1).repeat(1, 3, 1))
               .view(B, 3, npnt, ns) - new_pos.cpu().transpose(1, 2).unsqueeze(-1)) / r
        assert torch.equal(got[:, :3].cpu(), cpu)
    else:
        assert torch.equal(got, ref.detach())  # copies and one subtraction: exact
    if C:
        cot = torch.randn(B, C + 3, npnt, ns, generator=g).to(DEV)
        ref.backward(cot)
        got.backward(cot)
        torch.testing.assert_close(xf.grad, xr.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M_groups,ns,Cin,Cout,slope,pool", [(70, 16, 7, 24, 0.01, True), (512, 64, 131, 128, 0.01, True),
                                                              (900, 1, 28, 24, 0.01, False), (33, 5, 6, 64, 0.0, True),
                                                              (4096, 1, 259, 256, 0.01, False), (8, 128, 35, 33, 1.0, True)])
@pytest.mark.parametrize("training", [True, False])
def test_linear_bn_act_matches_torch(M_groups, ns, Cin, Cout, slope, pool, training):
    from torch_points3d_amd import fused
    g = torch.Generator().manual_seed(M_groups + Cin)
    M = M_groups * ns
    A = torch.randn(M, Cin, generator=g).to(DEV)
    conv_a, conv_b = nn.Conv2d(Cin, Cout, 1, bias=False).to(DEV), nn.Conv2d(Cin, Cout, 1, bias=False).to(DEV)
    conv_b.load_state_dict(conv_a.state_dict())
    bn_a, bn_b = nn.BatchNorm2d(Cout).to(DEV), nn.BatchNorm2d(Cout).to(DEV)
    with torch.no_grad():
        bn_a.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
        bn_a.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
        bn_a.weight[0] = -0.7  # a negative gamma must not break the fused max-pool
        bn_a.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
        bn_a.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    bn_b.load_state_dict(bn_a.state_dict())
    bn_a.train(training)
    bn_b.train(training)
    act = nn.LeakyReLU(slope) if slope not in (0.0,) else nn.ReLU()
    # reference: (1, Cin, groups, ns) image
    Ar = A.clone().requires_grad_(True)
    img = Ar.view(1, M_groups, ns, Cin).permute(0, 3, 1, 2)
    ref = act(bn_a(conv_a(img)))
    if pool:
        ref = F.max_pool2d(ref, kernel_size=[1, ns])
    ref_rows = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    Af = A.clone().requires_grad_(True)
    got = fused.linear_bn_act(Af, conv_b, bn_b, float(slope), ns if pool else 0)
    scale = max(1.0, float(ref_rows.detach().abs().max()))
    torch.testing.assert_close(got, ref_rows.detach(), rtol=1e-4, atol=1e-5 * scale)
    if training:
        torch.testing.assert_close(bn_b.running_mean, bn_a.running_mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(bn_b.running_var, bn_a.running_var, rtol=1e-5, atol=1e-6)
        assert int(bn_b.num_batches_tracked) == int(bn_a.num_batches_tracked) == 1
    cot = torch.randn(ref_rows.shape, generator=g).to(DEV)
    ref_rows.backward(cot)
    got.backward(cot)
    # a LeakyReLU / max-pool decision can flip on a last-bit difference, so gradients are bounded in L2
    tol = 1e-5 if slope == 1.0 and not pool else 2e-2
    assert _rel(Af.grad, Ar.grad) < tol
    assert _rel(conv_b.weight.grad, conv_a.weight.grad) < tol
    assert _rel(bn_b.weight.grad, bn_a.weight.grad) < tol
    assert _rel(bn_b.bias.grad, bn_a.bias.grad) < tol


@pytest.mark.parametrize("B,m,n,C1,C2", [(2, 16, 50, 8, 5), (2, 128, 512, 256, 128), (1, 3, 9, 4, 0), (3, 40, 100, 1, 3)])
def test_interp_concat_fwd_bwd(B, m, n, C1, C2, hip):
    from torch_points3d_amd import fused
    g = torch.Generator().manual_seed(m * n)
    known = torch.rand(B, m, 3, generator=g).to(DEV)
    unknown = torch.rand(B, n, 3, generator=g).to(DEV)
    feat = torch.randn(B, C1, m, generator=g).to(DEV)
    skip = torch.randn(B, C2, n, generator=g).to(DEV) if C2 else None
    dist, idx = hip.three_nn(unknown, known)
    # reference graph: DenseFPModule.conv + BaseDenseConvolutionUp.forward (core/base_conv/dense.py:102-144)
    dist_recip = 1.0 / (dist + 1e-8)
    w_ref = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
    w = fused.idw_weights(dist)
    torch.testing.assert_close(w, w_ref, rtol=1e-6, atol=1e-7)
    fr = feat.clone().requires_grad_(True)
    sr = skip.clone().requires_grad_(True) if C2 else None
    ref = hip.three_interpolate(fr, idx, w)
    if C2:
        ref = torch.cat([ref, sr], dim=1)
    ff = feat.clone().requires_grad_(True)
    sf = skip.clone().requires_grad_(True) if C2 else None
    rows = fused.interp_concat(ff.transpose(1, 2), idx, w, None if not C2 else sf.transpose(1, 2))
    ld = (C1 + C2 + 3) // 4 * 4
    assert rows.shape == (B * n, ld)
    got = rows.view(B, n, ld)[..., : C1 + C2].transpose(1, 2)
    assert torch.equal(got, ref.detach())
    cot = torch.randn(B, C1 + C2, n, generator=g).to(DEV)
    ref.backward(cot)
    got.backward(cot)
    torch.testing.assert_close(ff.grad, fr.grad, rtol=1e-5, atol=1e-5)
    if C2:
        assert torch.equal(sf.grad, sr.grad)


def test_fused_model_equals_unfused_model():
    """Same weights, same input: channel-last fused path vs the reference's (B,C,np,ns) PyTorch graph, on GPU."""
    from torch_points3d_amd.dense import Data
    from torch_points3d_amd.pointnet2 import PointNet2Unet
    for cfg in ("unet_3_ss", "unet_3_ms", "unet_4_ss"):
        torch.manual_seed(0)
        a = PointNet2Unet(3, output_nc=7, config=cfg, fused=True).to(DEV).train()
        torch.manual_seed(0)
        b = PointNet2Unet(3, output_nc=7, config=cfg, fused=False).to(DEV).train()
        g = torch.Generator().manual_seed(3)
        pos = (torch.rand(2, 4096, 3, generator=g) * 2 - 1).to(DEV)
        x = torch.randn(2, 4096, 3, generator=g).to(DEV)
        oa = a(Data(pos=pos, x=x)).x
        ob = b(Data(pos=pos, x=x)).x
        assert oa.shape == ob.shape == (2, 7, 4096)
        torch.testing.assert_close(oa, ob, rtol=1e-3, atol=1e-3)
        oa.square().mean().backward()
        ob.square().mean().backward()
        gmax = max(float(p.grad.norm()) for p in b.parameters())
        for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
            assert na == nb
            # (a BatchNorm bias that feeds another BatchNorm has a mathematically zero gradient: absolute floor)
            assert float((pa.grad - pb.grad).norm()) < 5e-2 * float(pb.grad.norm()) + 1e-5 * gmax, na
        for (na, ba), (nb, bb) in zip(a.named_buffers(), b.named_buffers()):
            torch.testing.assert_close(ba.float(), bb.float(), rtol=1e-3, atol=1e-4, msg=lambda m, na=na: na + m)


@pytest.mark.parametrize("M,N,K", [(1000, 64, 6), (4096, 128, 131), (70000, 10, 128), (33, 7, 5), (20000, 256, 259),
                                   (5000, 1024, 512), (262144, 128, 128), (17, 130, 70), (3000, 64, 150), (9000, 128, 384),
                                   (40000, 32, 128), (6000, 7, 260), (524288, 10, 128), (3000, 12, 131)])
def test_gemm_tn_matches_fp64(M, N, K):
    """split-K MFMA weight-gradient kernel vs an fp64 evaluation (and vs torch.mm's fp32 for scale)."""
    from torch_points3d_amd import fused
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g).to(DEV)
    A = torch.randn(M, K, generator=g).to(DEV)
    got = fused.gemm_tn(dY, A)
    ref = torch.mm(dY.double().t(), A.double())
    err = float((got.double() - ref).abs().max())
    lib = float((torch.mm(dY.t(), A).double() - ref).abs().max())
    assert got.shape == (N, K)
    assert err <= max(2.0 * lib, 1e-5 * float(ref.abs().max())), (err, lib)
    assert torch.equal(got, fused.gemm_tn(dY, A))  # fixed-order split reduction: reproducible
    # exact integer data: catches any operand / accumulator layout mix-up
    dYi = torch.randint(-3, 4, (M, N), generator=g).float().to(DEV)
    Ai = torch.randint(-3, 4, (M, K), generator=g).float().to(DEV)
    if M <= 70000:
        assert torch.equal(fused.gemm_tn(dYi, Ai), torch.mm(dYi.double().t(), Ai.double()).float())


def test_sharded_step_same_trajectory():
    """one graph replay and one eager step from identical states give identical parameters and BN buffers"""
    import copy
    import torch.nn.functional as F
    from torch_points3d_amd.dense import Data
    from torch_points3d_amd.dp import ShardedStep
    from torch_points3d_amd.pointnet2 import PointNet2Unet

    g = torch.Generator().manual_seed(6)
    pos = (torch.rand(2, 2048, 3, generator=g) * 2 - 1).to(DEV)
    x = torch.randn(2, 2048, 3, generator=g).to(DEV)
    y = torch.randint(0, 7, (2, 2048), generator=g).to(DEV)
    torch.manual_seed(0)
    net = PointNet2Unet(3, output_nc=7, config="unet_3_ss").to(DEV).train()
    tr = ShardedStep(net, lambda ps: torch.optim.SGD(ps, lr=0.05), lambda: F.cross_entropy(net(Data(pos=pos, x=x)).x, y),
                     world_size=1, use_graph=True)
    assert tr.warmup_and_capture(1)
    state = copy.deepcopy(net.state_dict())
    tr.step()  # graph replay
    torch.cuda.synchronize()
    after_graph = copy.deepcopy(net.state_dict())
    net.load_state_dict(state)
    tr.eager_step()
    torch.cuda.synchronize()
    for k, v in net.state_dict().items():
        torch.testing.assert_close(v.float(), after_graph[k].float(), rtol=1e-6, atol=1e-7, msg=lambda m, k=k: k + m)


def test_pipelined_step_same_trajectory_as_sharded_step():
    """geometry of step i+1 on a second stream during step i (dp.PipelinedStep, HIP graphs) follows exactly the
    trajectory of the plain stepper: same parameters and BatchNorm buffers after several steps, and the precomputed
    geometry is bit-identical to what the forward pass computes itself"""
    import torch.nn.functional as F
    from torch_points3d_amd.dense import Data
    from torch_points3d_amd.dp import PipelinedStep, ShardedStep
    from torch_points3d_amd.pointnet2 import PointNet2Unet

    g = torch.Generator().manual_seed(6)
    pos = (torch.rand(2, 4096, 3, generator=g) * 2 - 1).to(DEV)
    x = torch.randn(2, 4096, 3, generator=g).to(DEV)
    y = torch.randint(0, 7, (2, 4096), generator=g).to(DEV)

    def make():
        torch.manual_seed(0)
        return PointNet2Unet(3, output_nc=7, config="unet_3_ss").to(DEV).train()

    a, b = make(), make()
    geo = b.precompute_geometry(pos)
    assert torch.equal(geo.down[0].idx, a.down_modules[0].sampler(pos).long())
    with torch.no_grad():
        assert torch.equal(a(Data(pos=pos, x=x)).x, b(Data(pos=pos, x=x), geometry=geo).x)
    a, b = make(), make()
    ta = ShardedStep(a, lambda ps: torch.optim.SGD(ps, lr=0.05), lambda: F.cross_entropy(a(Data(pos=pos, x=x)).x, y),
                     world_size=1, use_graph=False)
    tb = PipelinedStep(b, lambda ps: torch.optim.SGD(ps, lr=0.05),
                       lambda slot: b.precompute_geometry(pos, backward_tables=True),  # inverted tables built ahead as well
                       lambda geo_: F.cross_entropy(b(Data(pos=pos, x=x), geometry=geo_).x, y), world_size=1, use_graph=True)
    assert tb.warmup_and_capture(2)
    for _ in range(2 + 2):  # the eager warm-up steps of the pipelined stepper (2) and its side-stream warm-up (2 passes)
        ta.step()
    for _ in range(5):  # crosses both geometry slots several times
        ta.step()
        tb.step()
    torch.cuda.synchronize()
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        torch.testing.assert_close(v.float(), w.float(), rtol=1e-5, atol=1e-6, msg=lambda m, k=k: k + m)


@pytest.mark.parametrize("M,N,K", [(1000, 64, 8), (4096, 128, 132), (70000, 128, 128), (33, 8, 4), (20000, 256, 260),
                                   (5000, 1024, 512), (262144, 128, 64), (129, 132, 68), (4096, 256, 1280),
                                   (140000, 256, 64), (50000, 384, 32), (130100, 128, 8)])
def test_gemm_rows_matches_fp64_and_stats(M, N, K):
    """fp32 MFMA rows GEMM (forward / input-gradient contraction) + fused BatchNorm statistics epilogue"""
    from torch_points3d_amd import fused
    K_ = K
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    Bm = torch.randn(K, N, generator=g).to(DEV)
    C, part = fused.gemm_rows(A, Bm.t().contiguous(), want_stats=True)
    ref = torch.mm(A.double(), Bm.double())
    err = float((C.double() - ref).abs().max())
    lib = float((torch.mm(A, Bm).double() - ref).abs().max())
    assert err <= max(2.0 * lib, 1e-5 * float(ref.abs().max())), (err, lib)
    # without statistics a long contraction with few tiles runs K-split (two-level summation): at least as accurate
    Cs, _ = fused.gemm_rows(A, Bm.t().contiguous())
    assert float((Cs.double() - ref).abs().max()) <= max(1.5 * err, 1e-6 * float(ref.abs().max()))
    from torch_points3d_amd import _lib
    # statistics chunks: one per (128-row block, wave row) or per (persistent workgroup, wave row); each holds four rows
    # of N: sum d, sum d^2 (d = value - shift), the shift, and the number of matrix rows that went into it
    chunks = _lib.load().tp3d_gemm_rows_stat_chunks(M, N)
    wave_rows = 4 if 0 < N % 128 <= 64 else 2  # 128 x 64 tiles (4 wave rows) for such widths, else 128 x 128 (2)
    tiles_n = -(-N // (64 if wave_rows == 4 else 128))
    assert chunks == wave_rows * ((M + 127) // 128) or chunks == wave_rows * (1024 // tiles_n)
    assert part.numel() >= chunks * 4 * N * 4
    p = part[: chunks * 4 * N * 4].view(torch.float32).view(chunks, 4, N).double()
    S, Q, Ks, n = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
    assert float(n[:, 0].sum()) == M and bool((n == n[:, :1]).all())
    mean = (n * Ks + S).sum(0) / M
    torch.testing.assert_close(mean, ref.mean(0), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    live = n > 0
    mk = torch.where(live, Ks + S / n.clamp(min=1), torch.zeros_like(Ks))
    m2 = (torch.where(live, Q - S * S / n.clamp(min=1), torch.zeros_like(Q)) + n * (mk - mean) ** 2).sum(0)
    torch.testing.assert_close(m2 / M, ref.var(0, unbiased=False), rtol=2e-5, atol=1e-6)
    # the statistics rows end exactly at chunks*4*N floats: a guard band behind them must stay untouched
    guard = torch.full((chunks * 4 * N + 8 * 2 * N,), 7.0, device=DEV)
    C2 = torch.empty_like(C)
    _lib.call("tp3d_gemm_rows_f32", A.data_ptr(), Bm.t().contiguous().data_ptr(), M, N, K_, C2.data_ptr(), guard.data_ptr(),
              None, _lib.stream_ptr(A.device))
    assert torch.equal(C2, C) and bool((guard[chunks * 4 * N:] == 7.0).all())
    # exact integer data: any operand / accumulator layout mix-up shows up as a wrong integer
    Ai = torch.randint(-3, 4, (M, K), generator=g).float().to(DEV)
    Bi = torch.randint(-3, 4, (K, N), generator=g).float().to(DEV)
    Ci, _ = fused.gemm_rows(Ai, Bi.t().contiguous())
    assert torch.equal(Ci, torch.mm(Ai.double(), Bi.double()).float())


@pytest.mark.parametrize("G,ns,C", [(700, 128, 128), (64, 32, 64), (33, 8, 1024), (5, 64, 2052), (300, 16, 100), (40, 7, 128),
                                    (90, 64, 7), (2048, 128, 128)])
def test_bn_act_maxpool_is_the_first_maximum_of_every_group(G, ns, C):
    """tp3d_bn_act_maxpool_f32 against the plain formulas, exactly: value and arg-max row, on data with many exact ties
    inside a group (max_pool2d's first-maximum rule) and negative scales."""
    from torch_points3d_amd import _lib
    g = torch.Generator().manual_seed(G + ns + C)
    Y = torch.randint(-3, 4, (G * ns, C), generator=g).float().to(DEV)  # small integers: ties everywhere
    mean = torch.randint(-1, 2, (C,), generator=g).float().to(DEV)
    scale = (torch.randint(0, 2, (C,), generator=g).float() * 2 - 1).to(DEV) * 0.5
    shift = torch.randint(-1, 2, (C,), generator=g).float().to(DEV) * 0.25
    slope = 0.25
    out = torch.empty(G, C, device=DEV)
    arg = torch.empty(G, C, dtype=torch.int32, device=DEV)
    _lib.call("tp3d_bn_act_maxpool_f32", Y.data_ptr(), mean.data_ptr(), scale.data_ptr(), shift.data_ptr(), slope, G, ns, C,
              out.data_ptr(), arg.data_ptr(), _lib.stream_ptr(Y.device))
    z = (Y - mean) * scale + shift
    a = torch.where(z > 0, z, z * slope).view(G, ns, C)
    want = a.max(dim=1).values
    first = (a == want.unsqueeze(1)).float().argmax(dim=1)  # the first row holding the maximum
    assert torch.equal(out, want)
    assert torch.equal(arg.long(), first)
    # a NaN in a window is the window's result, as with max_pool2d (the first one)
    Y[3 * ns + 2, 0] = float("nan")
    Y[3 * ns + 5, 0] = float("nan")
    _lib.call("tp3d_bn_act_maxpool_f32", Y.data_ptr(), mean.data_ptr(), scale.data_ptr(), shift.data_ptr(), slope, G, ns, C,
              out.data_ptr(), arg.data_ptr(), _lib.stream_ptr(Y.device))
    assert bool(torch.isnan(out[3, 0])) and int(arg[3, 0]) == 2
    assert not bool(torch.isnan(out[2])[0]) and torch.equal(out[4:], want[4:])


def test_scatter_tables_built_ahead_give_the_same_gradients():
    """group_concat / interp_concat with a table from fused.scatter_table (inverted before the pass, as the geometry
    prefetch does) against the same ops inverting inside their backward: bit-identical gradients."""
    from torch_points3d_amd import fused
    g = torch.Generator().manual_seed(5)
    B, N, npnt, ns, C = 3, 700, 128, 32, 20
    pos = torch.rand(B, N, 3, generator=g).to(DEV)
    new_pos = pos[:, :npnt].contiguous()
    idx = torch.randint(0, N, (B, npnt, ns), generator=g).to(DEV)
    x = torch.randn(B, N, C, generator=g).to(DEV)
    grads = []
    for table in (None, fused.scatter_table(idx, None, N, 1)):
        xr = x.clone().requires_grad_(True)
        rows = fused.group_concat(pos, new_pos, xr, idx, 0.3, True, table)
        rows.backward(torch.ones_like(rows) * torch.arange(rows.shape[1], device=DEV))
        grads.append(xr.grad)
    assert torch.equal(grads[0], grads[1])
    n, m, C1, C2 = 900, 128, 24, 10
    idx3 = torch.randint(0, m, (B, n, 3), generator=g).to(DEV)
    w3 = torch.rand(B, n, 3, generator=g).to(DEV)
    feat = torch.randn(B, m, C1, generator=g).to(DEV)
    skip = torch.randn(B, n, C2, generator=g).to(DEV)
    grads = []
    for table in (None, fused.scatter_table(idx3, w3, m, 3)):
        fr, sr = feat.clone().requires_grad_(True), skip.clone().requires_grad_(True)
        rows = fused.interp_concat(fr, idx3, w3, sr, table)
        rows.backward(torch.cos(torch.arange(rows.numel(), device=DEV, dtype=torch.float32)).view_as(rows))
        grads.append((fr.grad, sr.grad))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


@pytest.mark.parametrize("C", [20, 100, 320])
def test_scatter_backward_with_hub_points(C):
    """Neighbour tables in which a few support points collect hundreds or thousands of slots (the first hit of a padded
    ball query does): runs of more than 128 slots are summed by a whole workgroup in 16 pieces.  Against an fp64
    index_add, and identical from call to call."""
    from torch_points3d_amd import fused
    g = torch.Generator().manual_seed(C)
    B, N, npnt, ns = 3, 600, 128, 64
    idx = torch.randint(0, N, (B, npnt, ns), generator=g)
    idx[:, :, 40:] = idx[:, :, :1]                      # padding with the first hit
    idx[0, :100] = 7                                    # one point holding 6400 slots
    idx[1, :, :] = torch.randint(0, 3, (npnt, ns), generator=g)  # every slot on three points
    idx = idx.to(DEV)
    pos = torch.rand(B, N, 3, generator=g).to(DEV)
    new_pos = pos[:, :npnt].contiguous()
    x = torch.randn(B, N, C, generator=g).to(DEV)
    cot = None
    outs = []
    for table in (None, fused.scatter_table(idx, None, N, 1), None):
        xr = x.clone().requires_grad_(True)
        rows = fused.group_concat(pos, new_pos, xr, idx, 0.3, False, table)
        if cot is None:
            cot = torch.randn(rows.shape, generator=g).to(DEV)
        rows.backward(cot)
        outs.append(xr.grad)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    want = torch.zeros(B, N, C, dtype=torch.float64, device=DEV)
    gr = cot.view(B, npnt * ns, -1)[:, :, 3:3 + C].double()
    for b in range(B):
        want[b].index_add_(0, idx[b].reshape(-1), gr[b])
    torch.testing.assert_close(outs[0].double(), want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("M,N,offset", [(262144, 128, 0.0), (300000, 64, 40.0), (5000, 256, 0.0), (140000, 132, 3.0)])
def test_bn_finalize_from_gemm_partials(M, N, offset):
    """tp3d_bn_finalize_f32 on the partial sums a rows GEMM leaves: few chunks are folded by one workgroup per channel,
    many (2048 for a launch that fills the persistent grid) first into 32 slices, coalesced and in place.  Mean and
    variance against fp64, also where |mean| >> std; running statistics and the batch counter updated on the device."""
    from torch_points3d_amd import _lib, fused
    g = torch.Generator().manual_seed(M + N)
    K = 64
    A = (torch.randn(M, K, generator=g) * 0.1 + offset / K).to(DEV)
    Bm = (torch.rand(K, N, generator=g) + 0.5).to(DEV)
    C, part = fused.gemm_rows(A, Bm.t().contiguous(), want_stats=True)
    ref = C.double()
    chunks = _lib.load().tp3d_gemm_rows_stat_chunks(M, N)
    stats = torch.empty(4, N, device=DEV)
    gamma, beta = torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV)
    rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    _lib.call("tp3d_bn_finalize_f32", part.data_ptr(), chunks, M, N, 1e-5, 0.1, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(),
              rv.data_ptr(), nbt.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(),
              _lib.stream_ptr(A.device))
    mean, var = ref.mean(0), ref.var(0, unbiased=False)
    torch.testing.assert_close(stats[0].double(), mean, rtol=2e-7, atol=2e-7 * float(var.sqrt().max()))  # float resolution
    torch.testing.assert_close(stats[1].double(), 1.0 / torch.sqrt(var + 1e-5), rtol=2e-6, atol=0)
    torch.testing.assert_close(stats[2], gamma * stats[1])
    assert torch.equal(stats[3], beta) and int(nbt) == 1
    torch.testing.assert_close(rm.double(), 0.1 * mean, rtol=1e-6, atol=1e-7 * float(var.sqrt().max()))
    torch.testing.assert_close(rv.double(), 0.9 + 0.1 * var * M / (M - 1), rtol=1e-5, atol=0)


@pytest.mark.parametrize("train", [True, False])
def test_rows_mlp_with_linear_bias_matches_modules(train):
    """The reference's partial-dense MLP keeps the Linear bias in front of BatchNorm (base_modules.py:29-43): the fused
    row kernels run the GEMM without it and account for it in the running mean (training) or the affine shift (eval)."""
    import copy
    from torch_points3d_amd import fused
    from torch_points3d_amd.partial_dense import MLP
    torch.manual_seed(0)
    mlp = MLP([12, 128, 20], bn_momentum=0.1, bias=True).to(DEV)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_()
                m.running_var.uniform_(0.5, 2.0)
            if isinstance(m, torch.nn.Linear):
                m.bias.normal_()
    ref = copy.deepcopy(mlp)
    mlp.train(train)
    ref.train(train)
    x = torch.randn(5000, 12, device=DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    g = torch.randn(5000, 20, device=DEV)
    out = fused.rows_mlp(mlp, xa)
    want = ref(xb)
    out.backward(g)
    want.backward(g)
    torch.testing.assert_close(out, want, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(xa.grad, xb.grad, rtol=1e-3, atol=1e-4 * float(xb.grad.abs().max()))
    for (k, a), (_, b) in zip(mlp.state_dict().items(), ref.state_dict().items()):
        torch.testing.assert_close(a.float(), b.float(), rtol=1e-4, atol=1e-5, msg=lambda m, k=k: k + ": " + m)
    for (k, a), (_, b) in zip(mlp.named_parameters(), ref.named_parameters()):
        if train and k.endswith("0.bias"):
            # a bias in front of batch-statistics BatchNorm has an exactly zero gradient; autograd through the plain
            # modules leaves the rounding residue of a sum that cancels
            assert float(a.grad.abs().max()) == 0.0 and float(b.grad.abs().max()) < 1e-3, k
            continue
        scale = float(b.grad.abs().max()) + 1e-6
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-3, atol=1e-4 * scale + 1e-5, msg=lambda m, k=k: k + ": " + m)


@pytest.mark.parametrize("pool_ns", [0, 16])
@pytest.mark.parametrize("widths,M", [([8, 64, 64, 128], 4000), ([132, 128, 128, 256], 4000), ([12, 32, 96, 196], 4000),
                                      ([260, 64], 4000), ([12, 64, 128, 256, 132, 128], 66000)])
@pytest.mark.parametrize("train", [True, False])
def test_mlp_chain_matches_layerwise_path(widths, M, pool_ns, train):
    """The fused layer chain (BatchNorm / activation folded into the GEMM prologues, statistics in the epilogues,
    dY and the activated inputs never written) against the layer-by-layer kernels it replaces: same outputs, same
    gradients for the input rows and every parameter, same running statistics."""
    import copy
    from torch_points3d_amd import fused
    from torch_points3d_amd.dense import MLP2D
    torch.manual_seed(3)
    mlp = MLP2D(widths).to(DEV)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
                m.running_mean.normal_()
                m.running_var.uniform_(0.5, 2.0)
    twin = copy.deepcopy(mlp)
    mlp.train(train)
    twin.train(train)
    # (the 66000-row case is large enough for the split-role kernels: 128-, 256- and 128-column outputs on them, the 64- and
    # 132-column ones on the separate pass; the 4000-row cases exercise the chain's fall-back passes)
    rows = torch.randn(M, widths[0], generator=torch.Generator().manual_seed(5)).to(DEV)
    ra, rb = rows.clone().requires_grad_(True), rows.clone().requires_grad_(True)
    cot = torch.randn((M // pool_ns) if pool_ns else M, widths[-1], generator=torch.Generator().manual_seed(6)).to(DEV)
    old = fused.CHAIN_MIN_ROWS, fused.USE_MLP_CHAIN
    try:
        fused.CHAIN_MIN_ROWS, fused.USE_MLP_CHAIN = 0, True
        assert fused._chain_ok(ra, fused.mlp_parts(mlp))
        out = fused.run_mlp(ra, fused.mlp_parts(mlp), pool_ns)
        fused.USE_MLP_CHAIN = False
        want = fused.run_mlp(rb, fused.mlp_parts(twin), pool_ns)
    finally:
        fused.CHAIN_MIN_ROWS, fused.USE_MLP_CHAIN = old
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    out.backward(cot)
    want.backward(cot)
    # a LeakyReLU mask can flip on a last-bit forward difference between the two paths (different GEMM kernels for the
    # narrow layers, another partition of the statistics chunks): bound the bulk tightly, allow isolated outliers through
    # the L2 norm (the five-layer 66000-row case has 16 times the elements that can flip)
    bound = 1e-3 if M <= 4000 else 3e-3
    assert float((ra.grad - rb.grad).norm() / rb.grad.norm()) < bound
    for (k, a), (_, b) in zip(mlp.named_parameters(), twin.named_parameters()):
        assert float((a.grad - b.grad).norm() / (b.grad.norm() + 1e-12)) < bound, k
    for (k, a), (_, b) in zip(mlp.state_dict().items(), twin.state_dict().items()):
        torch.testing.assert_close(a.float(), b.float(), rtol=1e-5, atol=1e-6, msg=lambda m, k=k: k + ": " + m)


@pytest.mark.parametrize("M,N,K", [(66000, 128, 128), (65537, 256, 132), (131072, 128, 64), (70000, 512, 36), (66000, 224, 8), (131077, 64, 64),
                                   (66000, 48, 132)])
def test_split_role_gemm_applies_the_previous_layers_batchnorm(M, N, K):
    """tp3d_gemm_rows_bnact_sp_f32 against the two kernels it stands for (tp3d_bn_act_f32, then tp3d_gemm_rows_f32): the
    same output, the activated rows as side output bit for bit, statistics chunks that finalize to the statistics of the
    output; ragged row counts, a contraction tail, one to four column tiles, the 64-column tile form."""
    from torch_points3d_amd import _lib, fused
    h = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    Y = (torch.randn(M, K, generator=g) * 2 + 0.3).to(DEV)
    Bt = (torch.randn(N, K, generator=g) * 0.2).to(DEV)
    mean, scale, beta = (torch.randn(K, generator=g) * 0.2).to(DEV), (torch.rand(K, generator=g) + 0.5).to(DEV), \
        (torch.randn(K, generator=g) * 0.3).to(DEV)
    chunks = h.tp3d_gemm_rows_sp_chunks(M, N, K, 1)
    assert chunks in (2 * 512 // ((N + 127) // 128), 2 * 1024 // ((N + 127) // 128))
    st = _lib.stream_ptr(Y.device)
    act_ref = torch.empty_like(Y)
    _lib.call("tp3d_bn_act_f32", _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 0.01, M, K, _lib.ptr(act_ref), st)
    ref = fused.gemm_rows(act_ref, Bt)[0]
    out = torch.full((M, N), float("nan"), device=DEV)
    act = torch.full((M, K), float("nan"), device=DEV)
    part = torch.full((chunks * 4 * N + 64,), float("nan"), device=DEV)
    _lib.call("tp3d_gemm_rows_bnact_sp_f32", _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 0.01, _lib.ptr(Bt), M, N, K,
              _lib.ptr(out), _lib.ptr(part), _lib.ptr(act), 0, st)
    assert torch.equal(act, act_ref)
    # (close, not equal: the plain kernel is free to pick another tile shape, i.e. another summation grouping)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    assert bool(torch.isnan(part[chunks * 4 * N:]).all()) and not bool(torch.isnan(part[:chunks * 4 * N]).any())
    bn = torch.nn.BatchNorm1d(N).to(DEV)
    stats = fused._finalize_stats(part, M, N, bn.weight.detach(), bn.bias.detach(), bn, Y.device, st, chunks)
    torch.cuda.synchronize()
    o64 = out.double()
    std = o64.std(0, unbiased=False)
    assert float(((stats[0].double() - o64.mean(0)).abs() / std).max()) < 1e-5
    torch.testing.assert_close(stats[1].double(), 1.0 / torch.sqrt(o64.var(0, unbiased=False) + bn.eps), rtol=2e-5, atol=0)
    # without statistics and without the side output: the same product
    out2 = torch.empty_like(out)
    _lib.call("tp3d_gemm_rows_bnact_sp_f32", _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 0.01, _lib.ptr(Bt), M, N, K,
              _lib.ptr(out2), None, None, 0, st)
    assert torch.equal(out2, out)
    # statistics without the side output (another number of workgroups, another partition into chunks)
    chunks0 = h.tp3d_gemm_rows_sp_chunks(M, N, K, 0)
    part0 = torch.full((chunks0 * 4 * N,), float("nan"), device=DEV)
    _lib.call("tp3d_gemm_rows_bnact_sp_f32", _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 0.01, _lib.ptr(Bt), M, N, K,
              _lib.ptr(out2), _lib.ptr(part0), None, 0, st)
    stats0 = fused._finalize_stats(part0, M, N, bn.weight.detach(), bn.bias.detach(), bn, Y.device, st, chunks0)
    assert torch.equal(out2, out)
    assert float(((stats0[0] - stats[0]).double().abs() / std).max()) < 1e-5
    torch.testing.assert_close(stats0[1], stats[1], rtol=2e-5, atol=0)


@pytest.mark.parametrize("M,N,K", [(66000, 128, 128), (131077, 64, 64), (140001, 256, 132)])
def test_reverse_row_order_is_the_same_computation(M, N, K):
    """`reverse` = 1 walks the row blocks last to first (cache reuse between consecutive kernels): per-row outputs are bit
    for bit the forward-order ones, BatchNorm statistics / reductions / weight gradients are the same sums in another
    order -- ragged M (a partial last row block and padding blocks lead the reversed walk)."""
    from torch_points3d_amd import _lib, fused
    h = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    Y = (torch.randn(M, K, generator=g) * 2 + 0.3).to(DEV)
    Bt = (torch.randn(N, K, generator=g) * 0.2).to(DEV)
    mean, scale, beta = (torch.randn(K, generator=g) * 0.2).to(DEV), (torch.rand(K, generator=g) + 0.5).to(DEV), \
        (torch.randn(K, generator=g) * 0.3).to(DEV)
    st = _lib.stream_ptr(Y.device)
    bn = torch.nn.BatchNorm1d(N).to(DEV)
    for entry, chunk_fn in (("tp3d_gemm_rows_bnact_sp_f32", h.tp3d_gemm_rows_sp_chunks), ("tp3d_gemm_rows_bnact_x3_f32", h.tp3d_gemm_rows_x3_chunks)):
        chunks = chunk_fn(M, N, K, 1)
        if not chunks:
            continue
        res = []
        for rev in (0, 1):
            out = torch.full((M, N), float("nan"), device=DEV)
            act = torch.full((M, K), float("nan"), device=DEV)
            part = torch.full((chunks * 4 * N,), float("nan"), device=DEV)
            _lib.call(entry, _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 0.01, _lib.ptr(Bt), M, N, K, _lib.ptr(out),
                      _lib.ptr(part), _lib.ptr(act), rev, st)
            stats = fused._finalize_stats(part, M, N, bn.weight.detach(), bn.bias.detach(), bn, Y.device, st, chunks)
            res.append((out, act, stats))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), entry
        std = res[0][0].double().std(0, unbiased=False)
        assert float(((res[0][2][0] - res[1][2][0]).double().abs() / std).max()) < 1e-5, entry
        torch.testing.assert_close(res[0][2][1], res[1][2][1], rtol=2e-5, atol=0)
    # backward: reductions (bit for bit: a chunk keeps its slot), the input-gradient GEMM with its dY side output (bit for bit)
    dA = torch.randn(M, K, generator=g).to(DEV)
    invstd = (torch.rand(K, generator=g) + 0.5).to(DEV)
    ws = _lib.bn_workspace(M, K, Y.device)
    reds = []
    for rev in (0, 1):
        red = torch.empty(4, K, device=DEV)
        _lib.call("tp3d_bn_bwd_reduce_f32", _lib.ptr(dA), None, _lib.ptr(Y), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(mean),
                  _lib.ptr(invstd), 0.01, M, 1, K, 1, _lib.ptr(red[0]), _lib.ptr(red[1]), _lib.ptr(red[2]), _lib.ptr(red[3]), _lib.ptr(ws),
                  rev, st)
        reds.append(red)
    assert torch.equal(reds[0], reds[1])
    if h.tp3d_gemm_rows_bnbwd_sp_serves(M, N, K):
        outs = []
        for rev in (0, 1):
            out = torch.full((M, N), float("nan"), device=DEV)
            dY = torch.full((M, K), float("nan"), device=DEV)
            _lib.call("tp3d_gemm_rows_bnbwd_sp_f32", _lib.ptr(Y), _lib.ptr(dA), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta),
                      _lib.ptr(reds[0][2]), _lib.ptr(reds[0][3]), 0.01, _lib.ptr(Bt), M, N, K, _lib.ptr(out), N, 0, 0, _lib.ptr(dY), None, 1,
                      rev, st)
            outs.append((out, dY))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # weight gradient on the bf16 pipe: the same products summed in another order
    if h.tp3d_gemm_tn_x3_serves(M, N, K):
        dYn = torch.randn(M, N, generator=g).to(DEV)
        a = fused.gemm_tn(dYn, Y, reverse=0)
        b = fused.gemm_tn(dYn, Y, reverse=1)
        c = fused.gemm_tn(dYn, Y, act=(mean, scale, beta, 0.01), reverse=1)
        d = fused.gemm_tn(dYn, Y, act=(mean, scale, beta, 0.01), reverse=0)
        want = dYn.double().t() @ Y.double()
        mag = dYn.double().abs().t() @ Y.double().abs()
        for got in (a, b):
            assert float(((got.double() - want).abs() / mag).max()) < 1e-6
        assert float(((c - d).double().abs() / mag).max()) < 1e-6
    torch.cuda.synchronize()


def test_split_role_gemm_declines_shapes_it_does_not_serve():
    from torch_points3d_amd import _lib
    h = _lib.load()
    for M, N, K in [(4096, 128, 128), (66000, 192, 64), (66000, 320, 64), (66000, 128, 516), (66000, 128, 6), (66000, 384, 64)]:
        assert h.tp3d_gemm_rows_sp_chunks(M, N, K, 0) == 0 and h.tp3d_gemm_rows_sp_chunks(M, N, K, 1) == 0, (M, N, K)
    with pytest.raises(_lib.Tp3dError):
        t = torch.zeros(4096, 128, device=DEV)
        _lib.call("tp3d_gemm_rows_bnact_sp_f32", _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), 0.01, _lib.ptr(t), 4096, 128, 128,
                  _lib.ptr(t), None, None, 0, _lib.stream_ptr(t.device))


def test_split_role_gemm_constants_table_is_ready_before_the_first_tile():
    """The loader waves read the per-channel constants from LDS for the very first tile they stage.  Alternating launches
    with different constants leave the other launch's table behind in LDS: every launch must still use its own."""
    from torch_points3d_amd import _lib
    M, N, K = 131072, 128, 64
    g = torch.Generator().manual_seed(9)
    Y = torch.randn(M, K, generator=g).to(DEV)
    Bt = (torch.randn(N, K, generator=g) * 0.2).to(DEV)
    st = _lib.stream_ptr(Y.device)
    sets = []
    for i in range(2):
        mean, scale, beta = (torch.randn(K, generator=g) + 3 * i).to(DEV), (torch.rand(K, generator=g) + 0.5 + i).to(DEV), \
            (torch.randn(K, generator=g) - 2 * i).to(DEV)
        ref = torch.empty_like(Y)
        _lib.call("tp3d_bn_act_f32", _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 0.01, M, K, _lib.ptr(ref), st)
        sets.append((mean, scale, beta, ref))
    out = torch.empty(M, N, device=DEV)
    acts = [torch.empty_like(Y) for _ in range(8)]
    for i in range(8):
        mean, scale, beta, _ = sets[i & 1]
        _lib.call("tp3d_gemm_rows_bnact_sp_f32", _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 0.01, _lib.ptr(Bt), M, N, K,
                  _lib.ptr(out), None, _lib.ptr(acts[i]), 0, st)
    for i in range(8):
        assert torch.equal(acts[i], sets[i & 1][3]), i


@pytest.mark.parametrize("M,N,K", [(66000, 128, 128), (65537, 256, 132), (131072, 64, 64), (70000, 12, 64), (66000, 128, 256)])
def test_split_role_input_gradient_gemm_forms_dy_in_its_loaders(M, N, K):
    """tp3d_gemm_rows_bnbwd_sp_f32: dY side output against the BatchNorm + LeakyReLU backward formula in float64, the
    product against dY @ W in float64; the reduction constants come from tp3d_bn_bwd_reduce_f32 as in the chain."""
    from torch_points3d_amd import _lib
    h = _lib.load()
    assert h.tp3d_gemm_rows_bnbwd_sp_serves(M, N, K) == 1
    g = torch.Generator().manual_seed(M + N + K)
    Y = (torch.randn(M, K, generator=g) * 1.5 + 0.2).to(DEV)
    dA = torch.randn(M, K, generator=g).to(DEV)
    Wt = (torch.randn(N, K, generator=g) * 0.2).to(DEV)  # (input width, output width) = W^T
    gamma, beta = (torch.rand(K, generator=g) + 0.5).to(DEV), (torch.randn(K, generator=g) * 0.3).to(DEV)
    mean = Y.double().mean(0)
    invstd = 1.0 / torch.sqrt(Y.double().var(0, unbiased=False) + 1e-5)
    mean32, invstd32 = mean.float(), invstd.float()
    scale = (gamma.double() * invstd).float()
    st = _lib.stream_ptr(Y.device)
    red = torch.empty(4, K, device=DEV)
    ws = _lib.bn_workspace(M, K, Y.device)
    _lib.call("tp3d_bn_bwd_reduce_f32", _lib.ptr(dA), None, _lib.ptr(Y), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(mean32),
              _lib.ptr(invstd32), 0.01, M, 1, K, 1, _lib.ptr(red[0]), _lib.ptr(red[1]), _lib.ptr(red[2]), _lib.ptr(red[3]), _lib.ptr(ws), 0, st)
    out = torch.full((M, N), float("nan"), device=DEV)
    dY = torch.full((M, K), float("nan"), device=DEV)
    _lib.call("tp3d_gemm_rows_bnbwd_sp_f32", _lib.ptr(Y), _lib.ptr(dA), _lib.ptr(mean32), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(red[2]),
              _lib.ptr(red[3]), 0.01, _lib.ptr(Wt), M, N, K, _lib.ptr(out), N, 0, 0, _lib.ptr(dY), None, 1, 0, st)
    torch.cuda.synchronize()
    yc = Y.double() - mean32.double()
    z = yc * scale.double() + beta.double()
    dz = dA.double() * torch.where(z > 0, 1.0, 0.01)
    xhat = yc * invstd32.double()
    want = scale.double() * (dz - dz.mean(0) - xhat * (dz * xhat).mean(0))
    near_kink = z.abs() < 1e-5
    err = (dY.double() - want).abs()
    err[near_kink] = 0
    assert float(err.max()) < 2e-5 * float(want.abs().max())
    ref = dY.double() @ Wt.double().t()
    assert float((out.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max()) + 1e-6
    # without the side output: the same product
    out2 = torch.empty_like(out)
    _lib.call("tp3d_gemm_rows_bnbwd_sp_f32", _lib.ptr(Y), _lib.ptr(dA), _lib.ptr(mean32), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(red[2]),
              _lib.ptr(red[3]), 0.01, _lib.ptr(Wt), M, N, K, _lib.ptr(out2), N, 0, 0, None, None, 1, 0, st)
    assert torch.equal(out2, out)
    # into a column range of wider rows (the feature columns of grouped rows): nothing outside it is touched
    wide = torch.full((M, N + 8), 7.0, device=DEV)
    _lib.call("tp3d_gemm_rows_bnbwd_sp_f32", _lib.ptr(Y), _lib.ptr(dA), _lib.ptr(mean32), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(red[2]),
              _lib.ptr(red[3]), 0.01, _lib.ptr(Wt), M, N, K, wide.data_ptr() + 12, N + 8, 0, 0, None, None, 1, 0, st)
    assert torch.equal(wide[:, 3:3 + N], out) and bool((wide[:, :3] == 7).all()) and bool((wide[:, 3 + N:] == 7).all())
    # the three columns in front of the range and four of the five behind it written as zeros by the kernel
    _lib.call("tp3d_gemm_rows_bnbwd_sp_f32", _lib.ptr(Y), _lib.ptr(dA), _lib.ptr(mean32), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(red[2]),
              _lib.ptr(red[3]), 0.01, _lib.ptr(Wt), M, N, K, wide.data_ptr() + 12, N + 8, 3, 4, None, None, 1, 0, st)
    assert torch.equal(wide[:, 3:3 + N], out) and bool((wide[:, :3] == 0).all()) and bool((wide[:, 3 + N:3 + N + 4] == 0).all())
    assert bool((wide[:, 3 + N + 4:] == 7).all())
    assert h.tp3d_gemm_rows_bnbwd_sp_serves(M, N, 260) == 0 and h.tp3d_gemm_rows_bnbwd_sp_serves(4096, N, K) == 0


@pytest.mark.parametrize("M,N,K", [(1048576, 64, 8), (70001, 64, 8), (33000, 128, 16), (40000, 16, 4), (300, 64, 12), (5000, 256, 8), (777, 4, 4)])
def test_narrow_first_layer_weight_gradient_from_y_and_da(M, N, K):
    """tp3d_gemm_tn_bn_narrow_f32: dW = dY^T A with dY formed from (Y, dA), against float64 and against the two-pass route
    (tp3d_bn_act_bwd_f32 + tp3d_gemm_tn_f32); repeated launches bit-identical (fixed summation order)."""
    from torch_points3d_amd import _lib
    h = _lib.load()
    assert h.tp3d_gemm_tn_bn_narrow_serves(M, N, K) == 1 and h.tp3d_gemm_tn_bn_narrow_serves(M, N, 20) == 0
    assert h.tp3d_gemm_tn_bn_narrow_serves(M, N, 6) == 0 and h.tp3d_gemm_tn_bn_narrow_serves(M, 72, K) == 0
    g = torch.Generator().manual_seed(M + N + K)
    Y = (torch.randn(M, N, generator=g) * 1.5 + 0.2).to(DEV)
    dA = torch.randn(M, N, generator=g).to(DEV)
    A = torch.randn(M, K, generator=g).to(DEV)
    gamma, beta = (torch.rand(N, generator=g) + 0.5).to(DEV), (torch.randn(N, generator=g) * 0.3).to(DEV)
    mean = Y.mean(0)
    invstd = 1.0 / torch.sqrt(Y.var(0, unbiased=False) + 1e-5)
    scale = gamma * invstd
    st = _lib.stream_ptr(Y.device)
    red = torch.empty(4, N, device=DEV)
    ws = _lib.bn_workspace(M, N, Y.device)
    _lib.call("tp3d_bn_bwd_reduce_f32", _lib.ptr(dA), None, _lib.ptr(Y), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(mean),
              _lib.ptr(invstd), 0.01, M, 1, N, 1, _lib.ptr(red[0]), _lib.ptr(red[1]), _lib.ptr(red[2]), _lib.ptr(red[3]), _lib.ptr(ws), 0, st)
    nws = torch.empty(h.tp3d_gemm_tn_bn_narrow_workspace_floats(M, N, K), device=DEV)
    outs = []
    for _ in range(2):
        dW = torch.full((N, K), float("nan"), device=DEV)
        _lib.call("tp3d_gemm_tn_bn_narrow_f32", _lib.ptr(Y), _lib.ptr(dA), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta),
                  _lib.ptr(red[2]), _lib.ptr(red[3]), 0.01, _lib.ptr(A), M, N, K, _lib.ptr(dW), _lib.ptr(nws), 0, st)
        outs.append(dW)
    rev = torch.full((N, K), float("nan"), device=DEV)  # last rows first: the same row splits, each in its slot -- bit for bit
    _lib.call("tp3d_gemm_tn_bn_narrow_f32", _lib.ptr(Y), _lib.ptr(dA), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta),
              _lib.ptr(red[2]), _lib.ptr(red[3]), 0.01, _lib.ptr(A), M, N, K, _lib.ptr(rev), _lib.ptr(nws), 1, st)
    outs.append(rev)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    yc = Y.double() - mean.double()
    dz = dA.double() * torch.where(yc * scale.double() + beta.double() > 0, 1.0, 0.01)
    dY64 = scale.double() * ((dz - red[2].double()) - yc * red[3].double())
    want = dY64.t() @ A.double()
    mag = dY64.abs().t() @ A.double().abs()
    assert float(((outs[0].double() - want).abs() / mag).max()) < 1e-6
    # the two-pass route of the layer-wise backward
    dY = torch.empty_like(Y)
    dgb = torch.empty(2, N, device=DEV)
    _lib.call("tp3d_bn_act_bwd_f32", _lib.ptr(dA), None, _lib.ptr(Y), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(mean), _lib.ptr(invstd),
              0.01, M, 1, N, 1, _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), _lib.ptr(dY), _lib.ptr(ws), st)
    two = torch.empty(N, K, device=DEV)
    tws = _lib.gemm_tn_workspace(M, N, K, Y.device)
    _lib.call("tp3d_gemm_tn_f32", _lib.ptr(dY), _lib.ptr(A), M, N, K, _lib.ptr(two), _lib.ptr(tws), st)
    assert float(((two.double() - want).abs() / mag).max()) < 1e-6
    assert float(((outs[0] - two).abs().double() / mag).max()) < 1e-6


@pytest.mark.parametrize("M,N,K", [(1048576, 64, 8), (70001, 64, 8), (33000, 128, 16), (40000, 16, 4), (300, 64, 12), (5000, 256, 8), (777, 4, 4)])
def test_narrow_first_layer_forward_contraction_and_its_statistics(M, N, K):
    """tp3d_gemm_rows_narrow_f32: Y = A W^T against float64 (1e-5 of the scale: k ascending, fp32 FMA) and against the MFMA
    rows kernel; the statistics chunks finalize to the mean / variance of the output; both directions bit for bit."""
    from torch_points3d_amd import _lib, fused
    h = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, K, generator=g) + 0.3).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.4).to(DEV)
    st = _lib.stream_ptr(A.device)
    chunks = h.tp3d_gemm_rows_narrow_chunks(M)
    assert 1 <= chunks <= 1024
    outs = []
    for rev in (0, 1):
        Y = torch.full((M, N), float("nan"), device=DEV)
        part = torch.full((chunks * 4 * N,), float("nan"), device=DEV)
        _lib.call("tp3d_gemm_rows_narrow_f32", _lib.ptr(A), _lib.ptr(W), M, N, K, _lib.ptr(Y), _lib.ptr(part), rev, st)
        outs.append((Y, part.clone()))
    plain = torch.full((M, N), float("nan"), device=DEV)
    _lib.call("tp3d_gemm_rows_narrow_f32", _lib.ptr(A), _lib.ptr(W), M, N, K, _lib.ptr(plain), None, 0, st)
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(plain, outs[0][0])
    want = A.double() @ W.double().t()
    mag = A.double().abs() @ W.double().abs().t()
    assert float(((outs[0][0].double() - want).abs() / mag).max()) < 1e-6
    ref = fused.gemm_rows(A, W)[0]
    torch.testing.assert_close(outs[0][0], ref, rtol=1e-5, atol=1e-5 * float(want.abs().max()))
    bn = torch.nn.BatchNorm1d(N).to(DEV)
    stats = fused._finalize_stats(outs[0][1], M, N, bn.weight.detach(), bn.bias.detach(), bn, A.device, st, chunks)
    torch.cuda.synchronize()
    o64 = outs[0][0].double()
    std = o64.std(0, unbiased=False)
    assert float(((stats[0].double() - o64.mean(0)).abs() / std).max()) < 1e-5
    torch.testing.assert_close(stats[1].double(), 1.0 / torch.sqrt(o64.var(0, unbiased=False) + bn.eps), rtol=2e-5, atol=0)


@pytest.mark.parametrize("M,N,K", [(131072, 128, 128), (140001, 128, 64), (262144, 256, 128), (150016, 128, 112), (140033, 64, 64)])
@pytest.mark.parametrize("reverse", [0, 1])
def test_weight_gradient_kernel_also_reduces_the_layer_below(M, N, K, reverse):
    """tp3d_gemm_tn_x3_act_red_f32: dW bit for bit the plain activated-operand form's; dbeta / dgamma / c1 / c2 of the
    layer below against float64 and against tp3d_bn_bwd_reduce_f32 on the same (dA, Yp); eval-mode: c1 = c2 = 0."""
    from torch_points3d_amd import _lib, fused
    h = _lib.load()
    chunks = h.tp3d_gemm_tn_x3_red_chunks(M, N, K)
    assert chunks > 0 and h.tp3d_gemm_tn_x3_red_chunks(M, 128, 160) == 0 and h.tp3d_gemm_tn_x3_red_chunks(M, 64, 128) == 0
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g).to(DEV)
    Yp = (torch.randn(M, K, generator=g) * 1.5 + 0.2).to(DEV)
    dA = torch.randn(M, K, generator=g).to(DEV)
    gamma, beta = (torch.rand(K, generator=g) + 0.5).to(DEV), (torch.randn(K, generator=g) * 0.3).to(DEV)
    mean = Yp.mean(0)
    invstd = 1.0 / torch.sqrt(Yp.var(0, unbiased=False) + 1e-5)
    scale = gamma * invstd
    st = _lib.stream_ptr(dY.device)
    plain = fused.gemm_tn(dY, Yp, act=(mean, scale, beta, 0.01), reverse=reverse)
    ws = _lib.gemm_tn_workspace(M, N, K, dY.device, x3=True)
    rws = torch.full((chunks * 2 * K,), float("nan"), device=DEV)
    out = torch.full((N, K), float("nan"), device=DEV)
    red = torch.full((4, K), float("nan"), device=DEV)
    _lib.call("tp3d_gemm_tn_x3_act_red_f32", _lib.ptr(dY), _lib.ptr(Yp), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(invstd),
              0.01, _lib.ptr(dA), 1, M, N, K, 6, _lib.ptr(out), _lib.ptr(ws), _lib.ptr(red), _lib.ptr(rws), reverse, st)
    torch.cuda.synchronize()
    assert torch.equal(out, plain)
    assert bool(torch.isfinite(rws).all())
    yc = Yp.double() - mean.double()
    # (the side of the activation's kink decided in fp32, in the kernels' operation order: with 10^7 elements a few z
    #  round to the other side of zero in float64, and one flipped row moves a column sum by ~|dA|)
    positive = ((Yp - mean) * scale + beta) > 0
    dz = dA.double() * torch.where(positive, 1.0, 0.01)
    xh = yc * invstd.double()
    want = torch.stack([dz.sum(0), (dz * xh).sum(0)])
    mag = torch.stack([dz.abs().sum(0), (dz * xh).abs().sum(0)])
    assert float(((red[:2].double() - want).abs() / mag).max()) < 1e-6
    sep = torch.empty(4, K, device=DEV)
    bws = _lib.bn_workspace(M, K, dY.device)
    _lib.call("tp3d_bn_bwd_reduce_f32", _lib.ptr(dA), None, _lib.ptr(Yp), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(mean), _lib.ptr(invstd),
              0.01, M, 1, K, 1, _lib.ptr(sep[0]), _lib.ptr(sep[1]), _lib.ptr(sep[2]), _lib.ptr(sep[3]), _lib.ptr(bws), 0, st)
    assert float(((red[:2] - sep[:2]).double().abs() / mag).max()) < 1e-6
    torch.testing.assert_close(red[2], red[0] / M, rtol=1e-6, atol=0)
    torch.testing.assert_close(red[3], invstd * (red[1] / M), rtol=1e-6, atol=0)
    _lib.call("tp3d_gemm_tn_x3_act_red_f32", _lib.ptr(dY), _lib.ptr(Yp), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(invstd),
              0.01, _lib.ptr(dA), 0, M, N, K, 6, _lib.ptr(out), _lib.ptr(ws), _lib.ptr(red), _lib.ptr(rws), reverse, st)
    torch.cuda.synchronize()
    assert float(red[2:].abs().max()) == 0.0 and torch.equal(out, plain)


@pytest.mark.parametrize("M,N,K", [(65536, 64, 256), (1331, 256, 1024), (27, 2048, 1024), (216, 512, 3072), (9261, 64, 256), (5000, 36, 64),
                                   (70000, 128, 132)])
def test_rows_gemm_epilogue_applies_eval_batchnorm_and_activation(M, N, K):
    """tp3d_gemm_rows_epi_f32 == tp3d_gemm_rows_f32 followed by tp3d_bn_act_f32 (plain tiles: bit for bit; K-split launches: the
    epilogue runs in the slab sum), wide / narrow tiles, ragged rows, widths that are not a multiple of the tile."""
    from torch_points3d_amd import _lib, fused
    h = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    mean, scale, beta = (torch.randn(N, generator=g) * 0.2).to(DEV), (torch.rand(N, generator=g) + 0.5).to(DEV), \
        (torch.randn(N, generator=g) * 0.3).to(DEV)
    st = _lib.stream_ptr(A.device)
    Y = fused.gemm_rows(A, W)[0]
    want = torch.empty_like(Y)
    _lib.call("tp3d_bn_act_f32", _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 0.1, M, N, _lib.ptr(want), st)
    n = h.tp3d_gemm_rows_workspace_floats(M, N, K)
    ws = torch.empty(max(n, 1), device=DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    _lib.call("tp3d_gemm_rows_epi_f32", _lib.ptr(A), _lib.ptr(W), M, N, K, _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 0.1,
              _lib.ptr(out), _lib.ptr(ws) if n else None, st)
    torch.cuda.synchronize()
    assert torch.equal(out, want)


@pytest.mark.parametrize("widths,M,pool_ns", [([132, 128, 128, 256], 140032, 0), ([8, 64, 64, 128], 262144, 64)])
@pytest.mark.parametrize("train", [True, False])
def test_large_chain_gradients_with_and_without_the_fused_reductions(widths, M, pool_ns, train):
    """Row counts at which the bf16-pipe weight-gradient kernel serves (>= 131072): the chain with the BatchNorm-backward
    reductions riding on the weight-gradient kernels (fused.WGRAD_RED), the narrow first-layer kernels and alternating row
    directions, against the same chain with those three off -- every gradient within rounding, in train and in eval mode (eval:
    c1 = c2 = 0 in the reductions the weight-gradient kernel finalizes)."""
    import copy
    from torch_points3d_amd import fused
    from torch_points3d_amd.dense import MLP2D
    torch.manual_seed(13)
    mlp = MLP2D(widths).to(DEV)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
                m.running_mean.normal_()
                m.running_var.uniform_(0.5, 2.0)
    twin = copy.deepcopy(mlp)
    mlp.train(train)
    twin.train(train)
    rows = torch.randn(M, widths[0], generator=torch.Generator().manual_seed(5)).to(DEV)
    ra, rb = rows.clone().requires_grad_(widths[0] > 16), rows.clone().requires_grad_(widths[0] > 16)
    cot = torch.randn((M // pool_ns) if pool_ns else M, widths[-1], generator=torch.Generator().manual_seed(6)).to(DEV)
    names = ("WGRAD_RED", "WGRAD_NARROW", "FWD_NARROW", "ROW_ORDER_ALTERNATE")
    old = {n: getattr(fused, n) for n in names}
    seen = []
    real_call = fused._lib.call
    try:
        fused._lib.call = lambda name, *a: (seen.append(name), real_call(name, *a))[1]
        out = fused.run_mlp(ra, fused.mlp_parts(mlp), pool_ns)
        out.backward(cot)
        fused._lib.call = real_call
        for n in names:
            setattr(fused, n, False)
        want = fused.run_mlp(rb, fused.mlp_parts(twin), pool_ns)
        want.backward(cot)
    finally:
        fused._lib.call = real_call
        for n, v in old.items():
            setattr(fused, n, v)
    assert "tp3d_gemm_tn_x3_act_red_f32" in seen
    if widths[0] <= 16:
        assert {"tp3d_gemm_rows_narrow_f32", "tp3d_gemm_tn_bn_narrow_f32"} <= set(seen)
    torch.testing.assert_close(out.detach(), want.detach(), rtol=1e-5, atol=1e-5 * float(want.detach().abs().max()))
    # a LeakyReLU mask flips where a last-bit difference of the statistics (another order of the chunks) moves z across
    # zero -- a handful of the 5e7 hidden activations -- and every flip changes a whole row's contribution: bounded
    # through the L2 norm as in test_mlp_chain_matches_layerwise_path (a dropped 32-row block would be 1.5e-2)
    for (k, a), (_, b) in zip(mlp.named_parameters(), twin.named_parameters()):
        assert float((a.grad - b.grad).norm()) <= 3e-3 * float(b.grad.norm()) + 1e-10, k
    if ra.requires_grad:
        assert float((ra.grad - rb.grad).norm()) <= 3e-3 * float(rb.grad.norm())


def test_chain_contracts_only_the_feature_columns_of_grouped_rows():
    """Grouped rows are [relative position (3), features (C), padding]; their producer reads the gradient of the feature
    columns only, and the chain's first input-gradient GEMM computes just those (the rest stays zero): the gradient that
    reaches the features must not change."""
    import copy
    from torch_points3d_amd import fused
    from torch_points3d_amd import torchpoints as tp
    from torch_points3d_amd.dense import MLP2D
    torch.manual_seed(2)
    B, N, npnt, ns, C = 8, 4096, 256, 32, 128
    pos = (torch.rand(B, N, 3, device=DEV) * 2 - 1)
    new_pos = pos[:, :npnt].contiguous()
    idx = tp.ball_query(0.3, ns, pos, new_pos)[0]
    mlp = MLP2D([C + 3, 128, 128]).to(DEV).train()
    twin = copy.deepcopy(mlp)
    xa = torch.randn(B, N, C, device=DEV).requires_grad_(True)
    xb = xa.detach().clone().requires_grad_(True)
    old = fused.CHAIN_MIN_ROWS, fused.USE_MLP_CHAIN, fused.CHAIN_BWD_LOADER
    try:
        fused.CHAIN_MIN_ROWS, fused.USE_MLP_CHAIN, fused.CHAIN_BWD_LOADER = 0, True, True
        rows = fused.group_concat(pos, new_pos, xa, idx, 0.3, True)
        assert rows._tp3d_grad_cols == (3, C) and rows.shape[1] == 132
        out = fused.run_mlp(rows, fused.mlp_parts(mlp), ns)
        fused.CHAIN_BWD_LOADER = False
        want = fused.run_mlp(fused.group_concat(pos, new_pos, xb, idx, 0.3, True), fused.mlp_parts(twin), ns)
    finally:
        fused.CHAIN_MIN_ROWS, fused.USE_MLP_CHAIN, fused.CHAIN_BWD_LOADER = old
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    cot = torch.randn_like(out)
    out.backward(cot)
    want.backward(cot)
    assert float((xa.grad - xb.grad).norm() / xb.grad.norm()) < 1e-3
    for (k, a), (_, b) in zip(mlp.named_parameters(), twin.named_parameters()):
        assert float((a.grad - b.grad).norm() / (b.grad.norm() + 1e-12)) < 1e-3, k


@pytest.mark.parametrize("M,N,K,ns", [(65536 * 2, 128, 128, 64), (66048, 64, 256, 64), (131072, 128, 64, 128), (67072, 256, 128, 256)])
def test_split_role_input_gradient_gemm_with_pooled_gradient(M, N, K, ns):
    """The same kernel fed with the gradient of the max-pooled output and the winning rows: dY side output against the
    dense form of the same kernel on the scattered gradient (bit for bit), and the product likewise."""
    from torch_points3d_amd import _lib
    g = torch.Generator().manual_seed(M + ns)
    G = M // ns
    Y = (torch.randn(M, K, generator=g) * 1.5 + 0.2).to(DEV)
    dP = torch.randn(G, K, generator=g).to(DEV)
    arg = torch.randint(0, ns, (G, K), generator=g, dtype=torch.int32).to(DEV)
    dense = torch.zeros(G, ns, K, device=DEV)
    dense.scatter_(1, arg.long().unsqueeze(1), dP.unsqueeze(1))
    dense = dense.view(M, K).contiguous()
    Wt = (torch.randn(N, K, generator=g) * 0.2).to(DEV)
    mean, scale, beta = (torch.randn(K, generator=g) * 0.2).to(DEV), (torch.rand(K, generator=g) + 0.5).to(DEV), \
        (torch.randn(K, generator=g) * 0.3).to(DEV)
    c1, c2 = (torch.randn(K, generator=g) * 0.01).to(DEV), (torch.randn(K, generator=g) * 0.01).to(DEV)
    st = _lib.stream_ptr(Y.device)
    outs = []
    for dA, a_ptr, n_ in ((dense, None, 1), (dP, _lib.ptr(arg), ns)):
        out = torch.full((M, N), float("nan"), device=DEV)
        dY = torch.full((M, K), float("nan"), device=DEV)
        _lib.call("tp3d_gemm_rows_bnbwd_sp_f32", _lib.ptr(Y), _lib.ptr(dA), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), _lib.ptr(c1),
                  _lib.ptr(c2), 0.01, _lib.ptr(Wt), M, N, K, _lib.ptr(out), N, 0, 0, _lib.ptr(dY), a_ptr, n_, 0, st)
        outs.append((out, dY))
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0], outs[1][0])


def test_headline_network_with_and_without_the_loader_wave_chain():
    """The BASELINE SSG network at its full cloud size (N=16384, 8 clouds: every grouped MLP large enough for the
    split-role kernels) with the layer chain on and off: the same scores, the same loss gradient on every parameter within
    the tolerance a changed summation grouping of the BatchNorm statistics allows, the same running statistics."""
    from torch_points3d_amd import fused
    from torch_points3d_amd.dense import Data
    from torch_points3d_amd.pointnet2 import PointNet2Unet
    torch.manual_seed(0)
    B, N = 8, 16384
    g = torch.Generator().manual_seed(4)
    pos = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(DEV)
    x = torch.randn(B, N, 3, generator=g).to(DEV)
    y = torch.randint(0, 10, (B * N,), generator=g).to(DEV)
    net = PointNet2Unet(3, output_nc=10, config="unet_3_ss").to(DEV).train()
    twin = PointNet2Unet(3, output_nc=10, config="unet_3_ss").to(DEV).train()
    twin.load_state_dict(net.state_dict())
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    seen = []
    real_call = fused._lib.call

    def spy(name, *a):
        seen.append(name)
        return real_call(name, *a)
    old, old_bwd = fused.USE_MLP_CHAIN, fused.CHAIN_BWD_LOADER
    try:
        fused._lib.call = spy
        fused.USE_MLP_CHAIN = True
        out_a = net(Data(pos=pos, x=x)).x
        la = torch.nn.functional.cross_entropy(out_a.transpose(1, 2).reshape(-1, 10) if out_a.dim() == 3 else out_a, y)
        la.backward()
        used = set(seen)
        fused.USE_MLP_CHAIN = False
        out_b = twin(Data(pos=pos, x=x)).x
        lb = torch.nn.functional.cross_entropy(out_b.transpose(1, 2).reshape(-1, 10) if out_b.dim() == 3 else out_b, y)
        lb.backward()
        # third pass: the chain's forward (the same pre-activations, hence the same LeakyReLU masks as the first pass) with
        # its backward on the apply pass + library GEMM instead of the loader-wave form
        third = PointNet2Unet(3, output_nc=10, config="unet_3_ss").to(DEV).train()
        third.load_state_dict(state0)
        fused.USE_MLP_CHAIN, fused.CHAIN_BWD_LOADER = True, False
        out_c = third(Data(pos=pos, x=x)).x
        torch.nn.functional.cross_entropy(out_c.transpose(1, 2).reshape(-1, 10), y).backward()
    finally:
        fused._lib.call = real_call
        fused.USE_MLP_CHAIN, fused.CHAIN_BWD_LOADER = old, old_bwd
    assert torch.equal(out_c, out_a)
    for (k, a), (_, c) in zip(net.named_parameters(), third.named_parameters()):
        ref = float(c.grad.norm())
        if k.endswith(".bias") and k[:-4] + "weight" in dict(third.named_parameters()):
            ref = max(ref, float(dict(third.named_parameters())[k[:-4] + "weight"].grad.norm()))
        assert float((a.grad - c.grad).norm()) < 1e-4 * ref + 1e-12, k  # same masks: only rounding is left
    assert {"tp3d_gemm_rows_bnact_sp_f32", "tp3d_gemm_rows_bnbwd_sp_f32"} <= used
    scale = float(out_b.detach().abs().max())
    # two fp32 evaluations of a 17-layer train-mode network (different contraction kernels -- the chain's hidden layers run
    # on the bf16 matrix pipe --, another partition of the BatchNorm statistics): measured 1.35e-5 * scale on 4 of 1.3 M
    # scores, the rest inside 1e-5 * scale; each path's own distance to float64 is pinned per chain in
    # test_chain_on_the_bf16_pipe_is_as_close_to_float64_as_the_fp32_chain
    torch.testing.assert_close(out_a, out_b, rtol=1e-4, atol=2e-5 * scale)
    assert abs(float(la) - float(lb)) < 1e-5 * abs(float(lb))
    # Two fp32 evaluations of the same network differ in the last bit of some pre-activations (here: another partition
    # of the BatchNorm statistics into chunks), and an element that crosses the LeakyReLU kink changes its gradient by
    # 99 %: a fraction f of such elements moves a layer's gradient by ~sqrt(f) in the L2 norm.  Measured on this network:
    # 7e-4 at the head, growing to 6e-3 at the first layer.  A BatchNorm bias gradient is a sum with cancellation (the
    # two in front of a max-pool especially): it is compared on the scale of its layer's BatchNorm weight gradient.
    grads_b = dict((k, p.grad) for k, p in twin.named_parameters())
    for k, a in net.named_parameters():
        b = grads_b[k]
        ref = float(b.norm())
        if k.endswith(".bias") and k[:-4] + "weight" in grads_b:
            ref = max(ref, float(grads_b[k[:-4] + "weight"].norm()))
        assert float((a.grad - b).norm()) < 2e-2 * ref + 1e-12, k
    for (k, a), (_, b) in zip(net.state_dict().items(), twin.state_dict().items()):
        torch.testing.assert_close(a.float(), b.float(), rtol=1e-5, atol=1e-6, msg=lambda m, k=k: k + ": " + m)


def test_eval_statistics_follow_a_replayed_training_graph():
    """eval-mode BatchNorm statistics are cached per module; a HIP-graph replay of the training step runs no
    Python and moves no version counter, so the cache must be invalidated by the stepper -- validate, replay, validate."""
    from torch_points3d_amd.dense import Data
    from torch_points3d_amd.dp import ShardedStep
    from torch_points3d_amd.pointnet2 import PointNet2Unet
    torch.manual_seed(3)
    net = PointNet2Unet(3, output_nc=6, config="unet_3_ss").to(DEV).train()
    g = torch.Generator().manual_seed(5)
    pos = (torch.rand(2, 1024, 3, generator=g) * 2 - 1).to(DEV)
    x = torch.randn(2, 1024, 3, generator=g).to(DEV)
    y = torch.randint(0, 6, (2, 1024), generator=g).to(DEV)
    stepper = ShardedStep(net, lambda ps: torch.optim.Adam(ps, lr=1e-2, capturable=True),
                          lambda: F.cross_entropy(net(Data(pos=pos, x=x)).x, y), world_size=1, use_graph=True)
    graphed = stepper.warmup_and_capture(2)

    def validate():
        net.eval()
        with torch.no_grad():
            out = net(Data(pos=pos, x=x)).x.clone()
        net.train()
        return out

    def validate_uncached():
        for m in net.modules():
            if hasattr(m, "_tp3d_eval_stats"):
                del m._tp3d_eval_stats
        return validate()

    first = validate()
    for _ in range(3):
        stepper.step()
    torch.cuda.synchronize()
    second = validate()
    fresh = validate_uncached()
    assert graphed, "the test is about graph replay"
    assert torch.equal(second, fresh), "validation after replayed training used stale BatchNorm statistics"
    assert not torch.equal(first, second)  # three Adam steps at lr 1e-2 move the scores


def test_no_grad_forward_keeps_no_side_outputs():
    """under torch.no_grad() the parameters still report requires_grad inside Function.forward; the chain
    must not allocate / write the activated side outputs then, and must give the same scores as the grad-mode pass."""
    from torch_points3d_amd import _lib, fused
    from torch_points3d_amd.dense import Data
    from torch_points3d_amd.pointnet2 import PointNet2Unet
    torch.manual_seed(4)
    net = PointNet2Unet(3, output_nc=6, config="unet_3_ss").to(DEV).train()
    g = torch.Generator().manual_seed(6)
    pos = (torch.rand(4, 4096, 3, generator=g) * 2 - 1).to(DEV)
    x = torch.randn(4, 4096, 3, generator=g).to(DEV)
    seen = []
    prev = _lib.set_post_call_hook(lambda name, args: seen.append((name, args)))
    try:
        with torch.no_grad():
            a = net(Data(pos=pos, x=x)).x
        # argument 12 of the split-role forward GEMM is the side output (include/tp3d_hip.h)
        sp = [args for name, args in seen if name == "tp3d_gemm_rows_bnact_sp_f32"]
        assert sp and all(args[11] is None for args in sp)
        del seen[:]
        keep = fused.WGRAD_X3_ACT
        fused.WGRAD_X3_ACT = False  # (with it, layers whose weight gradient the bf16-pipe contraction serves keep none either)
        try:
            b = net(Data(pos=pos, x=x)).x
        finally:
            fused.WGRAD_X3_ACT = keep
        sp = [args for name, args in seen if name == "tp3d_gemm_rows_bnact_sp_f32"]
        assert sp and all(args[11] is not None for args in sp)
        del seen[:]
        c = net(Data(pos=pos, x=x)).x
        sp = [args for name, args in seen if name == "tp3d_gemm_rows_bnact_sp_f32"]
        h = _lib.load()
        assert sp and all((args[11] is None) == bool(fused.WGRAD_X3 and fused.WGRAD_X3_ACT and h.tp3d_gemm_tn_x3_serves(args[6], args[7], args[8]))
                          for args in sp)
        torch.testing.assert_close(c.detach(), b.detach(), rtol=1e-5, atol=1e-5)
    finally:
        _lib.set_post_call_hook(prev)
    assert fused._outer_grad is True
    torch.testing.assert_close(a, b.detach(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("M,N,K", [(131072, 128, 128), (140001, 128, 132), (262144, 256, 128), (131072, 128, 64),
                                   (150000, 64, 128), (131072, 128, 112), (131073, 128, 160), (200000, 256, 132), (140033, 64, 64)])
@pytest.mark.parametrize("terms", [9, 6])
def test_gemm_tn_x3_matches_fp64(M, N, K, terms):
    """The weight-gradient contraction on the bf16 matrix pipe (csrc/gemm_tn_x3.hip: every fp32 value split exactly into
    three bf16 terms, the product as 9 or 6 term pairs, fp32 accumulation) against float64 -- the bar of
    test_gemm_tn_matches_fp64 (within 2x the library GEMM's error or 1e-5 of the scale), and against the fp32 MFMA kernel's
    own error on the same inputs (measured 1.4-2.4x: the products are exact on both pipes, the bf16 MFMA's accumulator
    rounds differently); exact on small integers; reproducible; non-finite operands poison the same outputs."""
    from torch_points3d_amd import _lib, fused
    assert _lib.load().tp3d_gemm_tn_x3_serves(M, N, K)
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g).to(DEV)
    A = torch.randn(M, K, generator=g).to(DEV)
    got = fused.gemm_tn(dY, A, x3=terms)
    base = fused.gemm_tn(dY, A, x3=0)
    ref = torch.mm(dY.double().t(), A.double())
    scale = float(ref.abs().max())
    err, err32 = float((got.double() - ref).abs().max()), float((base.double() - ref).abs().max())
    lib = float((torch.mm(dY.t(), A).double() - ref).abs().max())
    assert got.shape == (N, K)
    assert err <= max(2.0 * lib, 1e-5 * scale), (err, lib, scale)
    assert err <= 3.0 * err32 + 1e-7 * scale, (err, err32)
    assert torch.equal(got, fused.gemm_tn(dY, A, x3=terms))
    dYi = torch.randint(-3, 4, (M, N), generator=g).float().to(DEV)
    Ai = torch.randint(-3, 4, (M, K), generator=g).float().to(DEV)
    assert torch.equal(fused.gemm_tn(dYi, Ai, x3=terms), torch.mm(dYi.double().t(), Ai.double()).float())
    # rows holding +-inf, NaN, a denormal-range value and FLT_MAX-sized values: the finite / non-finite pattern of the result
    # is the fp32 kernel's (an infinity may read NaN: x - hi(x) is NaN for an infinite x)
    dY[5, 3], dY[M - 2, 9], dY[77, 1] = float("inf"), float("nan"), 3.0e38
    A[11, 2], A[M // 2, 4], A[77, 0] = 1e-38, -float("inf"), 0.5
    a, b = fused.gemm_tn(dY, A, x3=terms), fused.gemm_tn(dY, A, x3=0)
    assert torch.equal(torch.isfinite(a), torch.isfinite(b))
    fin = torch.isfinite(b)
    assert float((a[fin].double() - b[fin].double()).abs().max()) <= 1e-5 * float(b[fin].abs().max())


@pytest.mark.parametrize("M,N,K", [(131072, 128, 128), (140000, 128, 132), (131072, 128, 64), (150001, 256, 160), (140033, 64, 64)])
def test_gemm_tn_x3_forms_the_activated_operand_bit_exactly(M, N, K):
    """tp3d_gemm_tn_x3_act_f32: A = LeakyReLU((Yp - mean) * scale + beta) formed by the loader waves == the plain contraction on
    the rows tp3d_bn_act_f32 writes (the forward kernels' expression and order), so dropping the forward pass's activated side
    output changes no bit of the weight gradient."""
    from torch_points3d_amd import _lib, fused
    g = torch.Generator().manual_seed(M + K)
    dY = torch.randn(M, N, generator=g).to(DEV)
    Yp = (torch.randn(M, K, generator=g) * 2 + 0.3).to(DEV)
    mean, scale, beta = torch.randn(K, generator=g).to(DEV), (torch.rand(K, generator=g) + 0.5).to(DEV), torch.randn(K, generator=g).to(DEV)
    scale[0] = -0.7
    slope = 0.01
    act = torch.empty_like(Yp)
    _lib.call("tp3d_bn_act_f32", _lib.ptr(Yp), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), slope, M, K, _lib.ptr(act),
              _lib.stream_ptr(Yp.device))
    want = fused.gemm_tn(dY, act, x3=6)
    got = fused.gemm_tn(dY, Yp, x3=6, act=(mean, scale, beta, slope))
    assert torch.equal(got, want)
    ref = torch.mm(dY.double().t(), torch.nn.functional.leaky_relu((Yp.double() - mean.double()) * scale.double() + beta.double(), slope))
    assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("M,N,K", [(66000, 128, 128), (65537, 256, 132), (131072, 128, 64), (70000, 512, 36)])
def test_forward_x3_gemm_is_as_accurate_as_the_fp32_mfma_form(M, N, K):
    """tp3d_gemm_rows_bnact_x3_f32 (fp32 contraction as six bf16 term pairs on the matrix pipe, exact three-term split) against
    float64 and the fp32 MFMA split-role kernel on the same inputs: output error no larger than 1.5x the fp32 form's, the
    activated side output bit-identical, statistics chunks that finalize to the output's statistics."""
    from torch_points3d_amd import _lib, fused
    h = _lib.load()
    g = torch.Generator().manual_seed(M + N)
    Y = (torch.randn(M, K, generator=g) * 1.7 + 0.4).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    mean, scale = torch.randn(K, generator=g).to(DEV), (torch.rand(K, generator=g) + 0.5).to(DEV)
    beta = torch.randn(K, generator=g).to(DEV)
    slope = 0.01
    act64 = torch.nn.functional.leaky_relu((Y.double() - mean.double()) * scale.double() + beta.double(), slope)
    ref = act64 @ W.double().t()
    res = {}
    for name, cq in (("tp3d_gemm_rows_bnact_sp_f32", h.tp3d_gemm_rows_sp_chunks), ("tp3d_gemm_rows_bnact_x3_f32", h.tp3d_gemm_rows_x3_chunks)):
        chunks = cq(M, N, K, 1)
        assert chunks > 0, name
        C = torch.empty(M, N, device=DEV)
        part = torch.empty(chunks * 4 * N, device=DEV)
        act = torch.empty(M, K, device=DEV)
        _lib.call(name, _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), slope, _lib.ptr(W), M, N, K, _lib.ptr(C),
                  _lib.ptr(part), _lib.ptr(act), 0, _lib.stream_ptr(Y.device))
        stats = torch.empty(4, N, device=DEV)
        rm, rv, nb = torch.zeros(N, device=DEV), torch.ones(N, device=DEV), torch.zeros(1, dtype=torch.long, device=DEV)
        ones, zeros = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)
        _lib.call("tp3d_bn_finalize_f32", _lib.ptr(part), chunks, M, N, 1e-5, 0.1, _lib.ptr(ones), _lib.ptr(zeros), _lib.ptr(rm),
                  _lib.ptr(rv), _lib.ptr(nb), _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(stats[2]), _lib.ptr(stats[3]),
                  _lib.stream_ptr(Y.device))
        res[name] = (C, act, stats)
    (o_sp, a_sp, s_sp), (o_x3, a_x3, s_x3) = res["tp3d_gemm_rows_bnact_sp_f32"], res["tp3d_gemm_rows_bnact_x3_f32"]
    assert torch.equal(a_sp, a_x3)
    scale_c = float(ref.abs().max())
    e_sp, e_x3 = float((o_sp.double() - ref).abs().max()) / scale_c, float((o_x3.double() - ref).abs().max()) / scale_c
    assert e_x3 < 1e-6 and e_x3 < 1.5 * e_sp + 1e-7, (e_sp, e_x3)
    std = ref.std(0, unbiased=False)
    assert float(((s_x3[0].double() - ref.mean(0)).abs() / std).max()) < 1e-5
    torch.testing.assert_close(s_x3[1], s_sp[1], rtol=2e-5, atol=0)
    assert h.tp3d_gemm_rows_x3_chunks(M, 64, K, 1) == 0 and h.tp3d_gemm_rows_x3_chunks(4096, N, K, 1) == 0


@pytest.mark.parametrize("widths,M,pool_ns", [([132, 128, 128, 256], 140000, 0), ([68, 128, 128], 262144, 64)])
def test_chain_on_the_bf16_pipe_is_as_close_to_float64_as_the_fp32_chain(widths, M, pool_ns):
    """A shared MLP (train-mode BatchNorm, LeakyReLU) large enough for the split-role kernels, evaluated by the fused chain
    with its hidden contractions on the bf16 matrix pipe (fused.FWD_X3) and on the fp32 MFMA, against a float64 evaluation of
    the same layers: both within 1e-5 of the scale, the bf16-pipe form no further from float64 than 1.5x the fp32 form."""
    import copy
    from torch_points3d_amd import fused
    from torch_points3d_amd.dense import MLP2D
    torch.manual_seed(11)
    mlp = MLP2D(widths).to(DEV).train()
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
    rows = (torch.randn(M, widths[0], generator=torch.Generator().manual_seed(7)) * 1.5 + 0.2).to(DEV)
    ref = rows.double()
    for conv, bn, slope in fused.mlp_parts(mlp):
        y = ref @ conv.weight.detach().double().reshape(conv.weight.shape[0], -1).t()
        mu, var = y.mean(0), y.var(0, unbiased=False)
        z = (y - mu) / torch.sqrt(var + bn.eps) * bn.weight.detach().double() + bn.bias.detach().double()
        ref = torch.nn.functional.leaky_relu(z, slope)
    if pool_ns:
        ref = ref.view(M // pool_ns, pool_ns, -1).max(1)[0]
    outs, used = {}, {}
    old = fused.FWD_X3, fused.CHAIN_MIN_ROWS
    real_call = fused._lib.call
    try:
        fused.CHAIN_MIN_ROWS = 0
        for flag in (True, False):
            seen = []
            fused._lib.call = lambda name, *a, seen=seen: (seen.append(name), real_call(name, *a))[1]
            fused.FWD_X3 = flag
            twin = copy.deepcopy(mlp)
            with torch.no_grad():
                outs[flag] = fused.run_mlp(rows, fused.mlp_parts(twin), pool_ns).double()
            used[flag] = set(seen)
    finally:
        fused._lib.call = real_call
        fused.FWD_X3, fused.CHAIN_MIN_ROWS = old
    assert "tp3d_gemm_rows_bnact_x3_f32" in used[True] and "tp3d_gemm_rows_bnact_x3_f32" not in used[False]
    scale = float(ref.abs().max())
    e_x3, e_32 = float((outs[True] - ref).abs().max()), float((outs[False] - ref).abs().max())
    assert e_x3 <= 1e-5 * scale and e_32 <= 1e-5 * scale, (e_x3, e_32, scale)
    assert e_x3 <= 1.5 * e_32 + 1e-7 * scale, (e_x3, e_32)
