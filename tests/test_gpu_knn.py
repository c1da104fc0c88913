"""Exact kNN on the uniform grid (csrc/knn.hip), the sort-based grid build for clouds beyond one workgroup's LDS
(csrc/grid.hip) and knn_interpolate / FPModule_PD (torch_points3d_amd/partial_dense.py).

Indices are compared bit-exact with the brute-force oracle (oracle/tpk_ref_cpu.c: tpk_ref_knn_partial_dense_f32; same
fp32 distance expression, ties by lower index).  Parity against torch_cluster's `knn` itself is unpinned (absent)."""
import pytest
import torch

from oracle.kpconv_cpu import torch_knn_interpolate

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def clouds(M, nclouds, seed, kind="uniform"):
    g = torch.Generator().manual_seed(seed)
    if kind == "uniform":
        x = torch.rand(M, 3, generator=g)
    elif kind == "plane":  # a thin slab: cells along z are almost all empty
        x = torch.rand(M, 3, generator=g) * torch.tensor([1.0, 1.0, 1e-3])
    elif kind == "clustered":  # most points inside one tiny blob: the candidate list overflows, fallback scan runs
        x = torch.rand(M, 3, generator=g)
        x[: M * 3 // 4] = 0.5 + 1e-3 * torch.rand(M * 3 // 4, 3, generator=g)
    elif kind == "lattice":  # exact ties everywhere
        x = torch.randint(0, 12, (M, 3), generator=g).float() * 0.125
    bx = torch.sort(torch.randint(0, nclouds, (M,), generator=g))[0] if nclouds > 1 else torch.zeros(M, dtype=torch.long)
    return x, bx


@pytest.mark.parametrize("M,Nq,nclouds,k,kind", [
    (1000, 400, 1, 1, "uniform"), (1000, 400, 3, 3, "uniform"), (5000, 3000, 2, 16, "uniform"),
    (20000, 9000, 4, 16, "plane"), (30000, 2000, 1, 5, "clustered"), (4000, 4000, 2, 4, "lattice"),
    (70, 50, 1, 100, "uniform"), (65536, 20000, 1, 1, "uniform"), (65536, 8000, 2, 16, "uniform"),
    (200000, 30000, 1, 16, "uniform"), (150000, 20000, 3, 1, "plane"), (3, 10, 1, 2, "uniform")])
def test_knn_partial_dense_matches_oracle(hip, oracle, M, Nq, nclouds, k, kind):
    x, bx = clouds(M, nclouds, M + k, kind)
    g = torch.Generator().manual_seed(Nq)
    # queries: support points, jittered copies, and points well outside the bounding box
    pick = torch.randint(0, M, (Nq,), generator=g)
    y = x[pick] + 0.01 * torch.randn(Nq, 3, generator=g)
    y[: Nq // 3] = x[pick[: Nq // 3]]
    y[-max(Nq // 20, 1):] += 3.0
    by, order = torch.sort(bx[pick])
    y = y[order].contiguous()
    if nclouds > 1:
        by[-1] = nclouds + 2  # a query whose cloud does not exist in the support
    idx, d2 = hip.knn(k, x.to(DEV), y.to(DEV), bx.to(DEV), by.to(DEV))
    ref_idx, ref_d2 = oracle.knn(k, x, y, bx, by)
    assert torch.equal(idx.cpu(), ref_idx)
    assert torch.equal(d2.cpu(), ref_d2)


def test_knn_dense_and_cell_hint(hip, oracle):
    g = torch.Generator().manual_seed(5)
    B, N, npq, k = 3, 3000, 700, 8
    x = torch.rand(B, N, 3, generator=g)
    y = torch.rand(B, npq, 3, generator=g)
    bx = torch.arange(B).repeat_interleave(N)
    by = torch.arange(B).repeat_interleave(npq)
    ref_idx, ref_d2 = oracle.knn(k, x.reshape(-1, 3), y.reshape(-1, 3), bx, by)
    ref_local = (ref_idx - (torch.arange(B) * N).repeat_interleave(npq)[:, None]).reshape(B, npq, k)
    for cell in (0.0, 0.01, 0.3, 50.0):
        idx, d2 = hip.knn(k, x.to(DEV), y.to(DEV), cell=cell)
        assert torch.equal(idx.cpu(), ref_local), cell
        assert torch.equal(d2.cpu().reshape(-1, k), ref_d2), cell


@pytest.mark.parametrize("M,Nq,nsample,sort", [(100000, 20000, 25, False), (100000, 10000, 16, True)])
def test_ball_query_large_cloud_uses_sorted_grid_build(hip, oracle, M, Nq, nsample, sort):
    # one cloud of more than 65536 points: the sort-based grid build (csrc/grid.hip gridg_*) serves it
    x, _ = clouds(M, 1, 11)
    bx = torch.cat([torch.zeros(M - 5000, dtype=torch.long), torch.ones(5000, dtype=torch.long)])
    g = torch.Generator().manual_seed(2)
    pick = torch.sort(torch.randint(0, M, (Nq,), generator=g))[0]
    y = (x[pick] + 0.002 * torch.randn(Nq, 3, generator=g)).contiguous()
    by = bx[pick]
    r = 0.03
    idx, d2 = hip.ball_query(r, nsample, x.to(DEV), y.to(DEV), mode="partial_dense", batch_x=bx.to(DEV),
                             batch_y=by.to(DEV), sort=sort)
    ref_idx, ref_d2 = oracle.ball_query(r, nsample, x, y, mode="partial_dense", batch_x=bx, batch_y=by, sort=sort)
    assert torch.equal(idx.cpu(), ref_idx)
    assert torch.equal(d2.cpu(), ref_d2)


@pytest.mark.parametrize("k,C,C2", [(1, 64, 32), (3, 17, 0), (3, 128, 128), (5, 4, 3)])
def test_knn_interpolate_forward_backward(oracle, k, C, C2):
    from torch_points3d_amd.partial_dense import knn_interpolate
    g = torch.Generator().manual_seed(k * 100 + C)
    M, Nq = 3000, 7000
    pos_x, bx = clouds(M, 2, 3)
    pos_y = torch.rand(Nq, 3, generator=g)
    by = torch.sort(torch.randint(0, 2, (Nq,), generator=g))[0]
    x = torch.randn(M, C, generator=g)
    skip = torch.randn(Nq, C2, generator=g) if C2 else None
    gout = torch.randn(Nq, C + C2, generator=g)
    # reference: plain torch fp32 on the oracle's neighbours
    idx, d2 = oracle.knn(k, pos_x, pos_y, bx, by)
    xr = x.clone().requires_grad_(True)
    ref = torch_knn_interpolate(xr, idx, d2)
    sr = None
    if C2:
        sr = skip.clone().requires_grad_(True)
        ref = torch.cat([ref, sr], dim=1)
    ref.backward(gout)
    xd = x.to(DEV).requires_grad_(True)
    sd = skip.to(DEV).requires_grad_(True) if C2 else None
    out = knn_interpolate(xd, pos_x.to(DEV), pos_y.to(DEV), bx.to(DEV), by.to(DEV), k=k, skip=sd)
    out.backward(gout.to(DEV))
    # tolerance: fp32, 1e-5 relative to the feature scale (north_star)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-5, atol=1e-5 * float(xr.grad.abs().max()))
    if C2:
        assert torch.equal(sd.grad.cpu(), sr.grad)


def test_fp_module_pd_and_neighbour_finder(oracle):
    from torch_points3d_amd.grid_sampling import GridSampling3D
    from torch_points3d_amd.kpconv_blocks import PDData
    from torch_points3d_amd.partial_dense import FPModule_PD, KNNNeighbourFinder
    torch.manual_seed(0)
    N = 6000
    pos = torch.rand(N, 3)
    batch = torch.sort(torch.randint(0, 2, (N,)))[0]
    fine = PDData(pos=pos.to(DEV), x=torch.randn(N, 16).to(DEV), batch=batch.to(DEV))
    coarse = GridSampling3D(0.1)(fine.clone())
    coarse.x = torch.randn(coarse.pos.shape[0], 32, device=DEV)
    fp = FPModule_PD(1, [32 + 16, 24], bn_momentum=0.2).to(DEV)
    out = fp((coarse, fine))
    assert out.x.shape == (N, 24) and out.pos is fine.pos
    # same module on the CPU with the oracle's neighbours
    idx, d2 = oracle.knn(1, coarse.pos.cpu(), pos, coarse.batch.cpu(), batch)
    feats = torch.cat([torch_knn_interpolate(coarse.x.cpu(), idx, d2), fine.x.cpu()], dim=1)
    ref = fp.cpu().nn(feats)
    torch.testing.assert_close(out.x.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    # every fine point's nearest coarse point is the mean of (one of) the voxels around it: within a voxel diagonal
    assert float(d2.max()) <= 3 * 0.1 ** 2 * 1.01
    # precomputed path gives the same features
    fp = fp.to(DEV)
    pre = fp.upsample_op.precompute(coarse, fine)
    out2 = fp((coarse, fine), precomputed=[pre])
    torch.testing.assert_close(out2.x, out.x, rtol=1e-5, atol=1e-5)
    edges = KNNNeighbourFinder(4)(coarse.pos, fine.pos, coarse.batch, fine.batch)
    ref_idx, _ = oracle.knn(4, coarse.pos.cpu(), pos, coarse.batch.cpu(), batch)
    assert edges.shape == (2, N * 4)
    assert torch.equal(edges[1].cpu().reshape(N, 4), ref_idx) and torch.equal(edges[0].cpu(), torch.arange(N).repeat_interleave(4))


def test_randla_conv_against_edge_list_formulation(oracle):
    """RandlaConv (fixed-k neighbour table) against the reference's message-passing formulation written with an
    explicit edge list and scatter-add in plain torch (modules/RandLANet/modules.py:25-54)."""
    from torch_points3d_amd.kpconv_blocks import PDData
    from torch_points3d_amd.randla import RandlaConv
    torch.manual_seed(0)
    N, C, k = 4000, 8, 16
    pos = torch.rand(N, 3)
    batch = torch.sort(torch.randint(0, 2, (N,)))[0]
    x = torch.randn(N, C)
    conv = RandlaConv(ratio=0.25, k=k, point_pos_nn=[10, 8, C], attention_nn=[2 * C, 8, 2 * C],
                      down_conv_nn=[2 * C, 8, 16]).to(DEV)
    out = conv(PDData(pos=pos.to(DEV), batch=batch.to(DEV), x=x.to(DEV)))
    idx = out.idx.cpu()
    assert idx.shape[0] == N // 4 and out.x.shape == (N // 4, 16)
    assert torch.equal(out.pos.cpu(), pos[idx]) and torch.equal(out.batch.cpu(), batch[idx])
    ref_nbr, _ = oracle.knn(k, pos, pos[idx], batch, batch[idx])
    assert torch.equal(out.neighbors.cpu(), ref_nbr)
    # edge-list formulation on the CPU with the same weights
    kern = conv._conv.cpu()
    row = torch.arange(idx.shape[0]).repeat_interleave(k)  # target (query) of each edge
    col = ref_nbr.reshape(-1)                                 # source (support) of each edge
    pos_i, pos_j, x_j = pos[idx][row], pos[col], x[col]
    vij = pos_i - pos_j
    rij = kern.point_pos_nn(torch.cat([pos_i, pos_j, vij, vij.norm(dim=1, keepdim=True)], 1))
    f = torch.cat([x_j, rij], 1)
    msg = torch.softmax(kern.attention_nn(f), -1) * f
    aggr = torch.zeros(idx.shape[0], f.shape[1]).index_add_(0, row, msg)
    ref = kern.global_nn(aggr)
    torch.testing.assert_close(out.x.detach().cpu(), ref.detach(), rtol=1e-3, atol=1e-4)


def test_grid_is_reused_only_for_the_same_support_and_radius(hip, oracle):
    """Consecutive partial-dense searches over the same support tensor with the same radius skip the grid build
    (KPConv: last block of a level, strided block of the next); anything else rebuilds.  Results always equal the
    oracle."""
    from torch_points3d_amd import _lib
    x, bx = clouds(30000, 2, 21)
    xd, bxd = x.to(DEV), bx.to(DEV)
    g = torch.Generator().manual_seed(4)

    def queries(n):
        pick = torch.sort(torch.randint(0, x.shape[0], (n,), generator=g))[0]
        return (x[pick] + 0.01 * torch.randn(n, 3, generator=g)).contiguous(), bx[pick]

    def check(r, xs, xsd):
        y, by = queries(5000)
        idx, d2 = hip.ball_query(r, 20, xsd, y.to(DEV), mode="partial_dense", batch_x=bxd, batch_y=by.to(DEV))
        ref, refd = oracle.ball_query(r, 20, xs, y, mode="partial_dense", batch_x=bx, batch_y=by)
        assert torch.equal(idx.cpu(), ref) and torch.equal(d2.cpu(), refd)

    timer = _lib.KernelTimer()
    _lib.set_timer(timer)
    try:
        check(0.05, x, xd)          # builds
        check(0.05, x, xd)          # same support, same radius: reuse
        check(0.08, x, xd)          # other radius: rebuild
        hip.knn(3, xd, xd[:100], bxd, bxd[:100])   # another user of the grid workspace
        check(0.08, x, xd)          # must rebuild
        xd.mul_(1.5)                # in-place change of the support (version counter moves)
        check(0.08, x * 1.5, xd)    # must rebuild
    finally:
        _lib.set_timer(None)
    # recorded integer arguments of the entry point: (M, Nq, nsample, sort, clouds, largest, workspace bytes, reuse, stream)
    reuse = [a[-2] for (name, a), _, _ in timer.records if name == "tp3d_ball_query_partial_dense_f32"]
    assert reuse == [0, 1, 0, 0, 0]


def test_knn_and_neighbour_ops_on_empty_inputs(hip):
    from torch_points3d_amd.fused import nbr_maxpool
    from torch_points3d_amd.partial_dense import knn_interpolate
    x = torch.rand(50, 3, device=DEV)
    empty = torch.empty(0, 3, device=DEV)
    bx = torch.zeros(50, dtype=torch.long, device=DEV)
    be = torch.zeros(0, dtype=torch.long, device=DEV)
    idx, d2 = hip.knn(4, x, empty, bx, be)                  # no queries
    assert idx.shape == (0, 4) and d2.shape == (0, 4)
    idx, d2 = hip.knn(4, empty, x, be, bx)                  # no support at all
    assert idx.shape == (50, 4) and bool((idx == -1).all()) and bool((d2 == -1).all())
    idx, d2 = hip.knn(3, x, x, bx, bx + 1)                  # the queries' cloud holds no support point
    assert bool((idx == -1).all())
    idx, d2 = hip.knn(1, x, x, bx, bx)
    assert torch.equal(idx[:, 0].cpu(), torch.arange(50)) and float(d2.abs().max()) == 0.0
    out = nbr_maxpool(torch.rand(50, 8, device=DEV), torch.empty(0, 5, dtype=torch.long, device=DEV))
    assert out.shape == (0, 8)
    # interpolation without a skip tensor and with a single support point
    feats = torch.randn(1, 6, device=DEV)
    y = knn_interpolate(feats, x[:1], x, bx[:1], bx, k=3)
    torch.testing.assert_close(y, feats.expand(50, 6), rtol=1e-6, atol=1e-6)


def test_fp_module_pd_global_innermost_on_device():
    from torch_points3d_amd.kpconv_blocks import PDData
    from torch_points3d_amd.partial_dense import FPModule_PD
    torch.manual_seed(0)
    N = 300
    batch = torch.sort(torch.randint(0, 3, (N,)))[0].to(DEV)
    skip = PDData(pos=torch.rand(N, 3, device=DEV), x=torch.randn(N, 4, device=DEV), batch=batch)
    pooled = PDData(pos=torch.zeros(3, 3, device=DEV), x=torch.randn(3, 6, device=DEV), batch=torch.arange(3, device=DEV))
    fp = FPModule_PD(1, [10, 5], bn_momentum=0.1).to(DEV).eval()
    out = fp((pooled, skip))
    want = fp.nn(torch.cat([pooled.x[batch], skip.x], dim=1))
    torch.testing.assert_close(out.x, want, rtol=1e-5, atol=1e-5)
