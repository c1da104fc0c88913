"""Parity of the HIP kernels (through the C-ABI / torch_points_kernels API) against the CPU oracle, the
reference-generated goldens, and -- at BASELINE.json's full sizes -- size-independent properties.

Bar: indices bit-exact; fp32 features within 1e-5 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = dict(rtol=1e-5, atol=1e-5)


def cloud(B, N, seed, kind="uniform"):
    g = torch.Generator().manual_seed(seed)
    if kind == "uniform":
        return torch.rand(B, N, 3, generator=g) * 2 - 1
    if kind == "randn":
        return torch.randn(B, N, 3, generator=g)
    if kind == "clustered":  # heavy ties / duplicates and dense balls
        p = torch.rand(B, N, 3, generator=g) * 2 - 1
        p[:, N // 2:] = p[:, : N - N // 2].clone()
        return p
    if kind == "lattice":  # exact distance ties everywhere
        return torch.randint(0, 5, (B, N, 3), generator=g).float() * 0.25
    raise ValueError(kind)


# --------------------------------------------------------------------------------------------- FPS

@pytest.mark.parametrize("B,N,npoint,kind", [
    (2, 1024, 512, "randn"), (3, 512, 128, "uniform"), (1, 5, 3, "uniform"), (2, 64, 64, "uniform"),
    (2, 65, 17, "uniform"), (4, 2048, 512, "uniform"), (2, 3000, 700, "clustered"), (2, 4096, 64, "lattice"),
    (2, 16384, 512, "uniform"), (1, 20000, 300, "uniform"), (1, 40000, 64, "uniform"), (2, 1, 1, "uniform"),
    (2, 300, 300, "lattice"),
])
def test_fps_bit_exact(hip, oracle, B, N, npoint, kind):
    pos = cloud(B, N, 10 + N, kind)
    got = hip.furthest_point_sample(pos.to(DEV), npoint)
    assert got.dtype == torch.int64 and got.shape == (B, npoint)
    assert torch.equal(got.cpu(), oracle.furthest_point_sample(pos, npoint))


def test_fps_kat_and_errors(hip):
    pos = torch.tensor([[[0, 0, 0], [0.5, 0.5, 0], [0.4, 0.2, 0], [2, 2, 2], [-1, -2, -0.01]]]).float().to(DEV)
    assert hip.furthest_point_sample(pos, 3)[0].tolist() == [0, 3, 4]  # reference test/test_fps.py:35-42
    with pytest.raises(ValueError):
        hip.furthest_point_sample(pos, 6)
    assert hip.furthest_point_sample(pos, 0).shape == (1, 0)
    assert hip.furthest_point_sample(pos[:0], 3).shape == (0, 3)


# --------------------------------------------------------------------------------------- ball query

@pytest.mark.parametrize("B,N,npnt,r,ns,kind", [
    (2, 1024, 512, 0.2, 64, "randn"), (2, 512, 128, 0.4, 64, "randn"), (3, 2048, 300, 0.2, 32, "uniform"),
    (2, 1000, 77, 0.1, 16, "uniform"), (2, 1500, 64, 0.45, 8, "clustered"), (2, 900, 50, 0.26, 128, "lattice"),
    (1, 3, 3, 1.01, 32, "uniform"), (2, 5000, 33, 2.5, 64, "uniform"), (1, 70, 70, 0.5, 1, "uniform"),
    (4, 16384, 512, 0.2, 64, "uniform"), (2, 2048, 100, 3.0, 16, "uniform"), (2, 4096, 256, 0.3, 48, "clustered"),
    (2, 3000, 128, 0.26, 200, "lattice"), (1, 30000, 700, 0.05, 32, "randn"), (2, 8192, 64, 0.01, 8, "uniform"),
])
@pytest.mark.parametrize("sort", [False, True])
def test_ball_query_dense_bit_exact(hip, oracle, B, N, npnt, r, ns, kind, sort):
    if sort and N > 4096 and r > 1.0:
        pytest.skip("every point in every ball, sorted: quadratic selection fallback, covered at N=2048")
    x = cloud(B, N, 20 + N, kind)
    y = x[:, torch.randperm(N, generator=torch.Generator().manual_seed(N))[:npnt]].contiguous()
    y[:, -1] = 50.0  # one query with an empty ball
    gi, gd = hip.ball_query(r, ns, x.to(DEV), y.to(DEV), sort=sort)
    ri, rd = oracle.ball_query(r, ns, x, y, sort=sort)
    assert gi.dtype == torch.int64 and gi.shape == (B, npnt, ns)
    assert torch.equal(gi.cpu(), ri)
    assert torch.equal(gd.cpu(), rd)  # same fp32 evaluation order: exact


@pytest.mark.parametrize("case", ["chunk_tail", "one_query", "same_place", "planar", "dense_ball", "zero_radius", "far_apart",
                                  "one_cell", "slot_limit"])
def test_ball_query_dense_edge_geometries(hip, oracle, case):
    """Unsorted dense queries at set-abstraction sizes on the geometries that stress a grid search: a cloud one point past
    a 1024 boundary, a single centre, coincident centres, a planar cloud, saturated and empty balls, centres far outside
    the cloud's box."""
    g = torch.Generator().manual_seed(11)
    B, N, npnt, r, ns = 2, 5000, 512, 0.2, 64
    x = torch.rand(B, N, 3, generator=g) * 2 - 1
    y = x[:, torch.randperm(N, generator=g)[:npnt]].contiguous()
    if case == "chunk_tail":
        N = 2049
        x = x[:, :N].contiguous()
        y = x[:, -npnt:].contiguous()  # includes the single point of the last chunk
    elif case == "one_query":
        y = x[:, 4321:4322].contiguous()
    elif case == "same_place":  # every centre on the same point: one cell, 512 lists hit by the same support points
        y = x[:, 7:8].expand(B, npnt, 3).contiguous()
    elif case == "planar":
        x[..., 1] = 0.5
        y = x[:, :npnt].contiguous()
    elif case == "dense_ball":  # far more hits than slots in the first chunk already
        x = x * 0.05
        y = x[:, :npnt].contiguous()
        ns = 16
    elif case == "zero_radius":
        r = 0.0
    elif case == "one_cell":  # the whole cloud on one point: a grid of a single cell, most cell slabs of the build empty
        x = x[:, :1].expand(B, N, 3).contiguous()
        y = x[:, :npnt].contiguous()
    elif case == "slot_limit":  # balls of 150..1000 hits, 45 % above the per-query hit slots of the grid search (640)
        x = x * 0.5
        r = 0.36
        y = x[:, :npnt].contiguous()
    else:  # centres far outside the cloud's box and from each other
        y = (torch.rand(B, npnt, 3, generator=g) * 2 - 1) * 40
        y[:, :8] = x[:, :8]
    gi, gd = hip.ball_query(r, ns, x.to(DEV), y.to(DEV))
    ri, rd = oracle.ball_query(r, ns, x, y)
    assert torch.equal(gi.cpu(), ri)
    assert torch.equal(gd.cpu(), rd)


def test_ball_query_dirichlet_kat(hip):
    # reference test/test_losses.py:16-24
    pos = torch.tensor([[[0, 0, 0], [1, 0, 0], [1.1, 0, 0]]], dtype=torch.float, device=DEV)
    f = torch.tensor([[1, 1, 3]], dtype=torch.float, device=DEV)
    nei = hip.ball_query(1.01, 32, pos, pos, sort=True)[0].reshape(1, -1).long()
    fn = f.gather(1, nei).reshape(1, 3, -1)
    var = ((f.unsqueeze(-1).repeat(1, 1, fn.shape[-1]) - fn) ** 2).sum(-1)
    assert var.cpu().tolist() == [[0.0, 4.0, 4.0]]


@pytest.mark.parametrize("sort", [False, True])
def test_ball_query_partial_dense_bit_exact(hip, oracle, sort):
    g = torch.Generator().manual_seed(3)
    sizes = [700, 1, 0, 1300, 64]  # ragged clouds incl. an empty and a single-point one
    x = torch.cat([torch.rand(n, 3, generator=g) for n in sizes])
    bx = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(sizes)])
    qsizes = [100, 3, 2, 200, 10]  # cloud 2 has queries but no support points
    y = torch.cat([torch.rand(n, 3, generator=g) for n in qsizes])
    by = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(qsizes)])
    for r, ns in [(0.15, 25), (0.3, 8), (0.05, 16)]:
        gi, gd = hip.ball_query(r, ns, x.to(DEV), y.to(DEV), mode="partial_dense", batch_x=bx.to(DEV),
                                batch_y=by.to(DEV), sort=sort)
        ri, rd = oracle.ball_query(r, ns, x, y, mode="partial_dense", batch_x=bx, batch_y=by, sort=sort)
        assert torch.equal(gi.cpu(), ri) and torch.equal(gd.cpu(), rd)
        assert (ri == -1).any()


@pytest.mark.parametrize("sort", [False, True])
def test_ball_query_partial_dense_grid_bit_exact(hip, oracle, sort):
    """clouds large enough for the uniform-grid path (>= 2048 points), ragged, with queries outside the boxes"""
    g = torch.Generator().manual_seed(11)
    sizes = [5000, 300, 0, 9000, 2048]
    x = torch.cat([torch.rand(n, 3, generator=g) * (1 + i) for i, n in enumerate(sizes)])
    bx = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(sizes)])
    qsizes = [700, 50, 4, 900, 333]
    y = torch.cat([torch.rand(n, 3, generator=g) * (1 + i) * 1.2 - 0.1 for i, n in enumerate(qsizes)])
    by = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(qsizes)])
    for r, ns in [(0.12, 25), (0.4, 16), (0.03, 40)]:
        gi, gd = hip.ball_query(r, ns, x.to(DEV), y.to(DEV), mode="partial_dense", batch_x=bx.to(DEV),
                                batch_y=by.to(DEV), sort=sort)
        ri, rd = oracle.ball_query(r, ns, x, y, mode="partial_dense", batch_x=bx, batch_y=by, sort=sort)
        assert torch.equal(gi.cpu(), ri) and torch.equal(gd.cpu(), rd)


# ------------------------------------------------------------------------------------------ three_nn

@pytest.mark.parametrize("B,n,m,kind", [
    (2, 1024, 512, "randn"), (2, 512, 128, "uniform"), (3, 777, 3, "uniform"), (2, 2000, 1500, "lattice"),
    (1, 1, 5, "uniform"), (2, 16384, 512, "uniform"),
])
def test_three_nn(hip, oracle, B, n, m, kind):
    unknown = cloud(B, n, 30 + n, kind)
    known = cloud(B, m, 31 + m, kind)
    gd, gi = hip.three_nn(unknown.to(DEV), known.to(DEV))
    rd, ri = oracle.three_nn(unknown, known)
    assert torch.equal(gi.cpu(), ri)
    assert torch.equal(gd.cpu(), rd)  # correctly rounded sqrt of a bit-identical squared distance
    with pytest.raises(ValueError):
        hip.three_nn(unknown.to(DEV), known[:, :2].to(DEV))


@pytest.mark.parametrize("case", ["subset", "lattice", "duplicates", "planar", "one_place", "far_queries", "m64", "m896",
                                  "m897", "line", "two_clusters", "sparse_shell"])
def test_three_nn_grid_path_is_exact(hip, oracle, case):
    """The in-LDS grid walk (64 <= m <= 896, n >= 4 m) must give the scan's answer on the geometries that stress its
    stop rule and its (distance, index) ranking: exact ties, duplicates, degenerate boxes, queries outside the box of the
    known cloud, lanes that need several shells."""
    g = torch.Generator().manual_seed(7)
    B, n, m = 2, 5000, 512
    unknown = torch.rand(B, n, 3, generator=g) * 2 - 1
    if case == "subset":  # the decoder's case: the known points are a subset of the unknown ones (distance 0 hits)
        known = unknown[:, torch.randperm(n, generator=g)[:m]].contiguous()
    elif case == "lattice":
        unknown = torch.randint(0, 9, (B, n, 3), generator=g).float() * 0.125
        known = torch.randint(0, 9, (B, m, 3), generator=g).float() * 0.125
    elif case == "duplicates":
        known = (torch.rand(B, m // 4, 3, generator=g) * 2 - 1).repeat(1, 4, 1)
    elif case == "planar":
        known = torch.rand(B, m, 3, generator=g) * 2 - 1
        known[..., 2] = 0.25
    elif case == "one_place":
        known = torch.full((B, m, 3), 0.5)
    elif case == "far_queries":
        known = torch.rand(B, m, 3, generator=g) * 0.2
        unknown = unknown * 5
    elif case == "m64":
        m = 64
        known = torch.rand(B, m, 3, generator=g) * 2 - 1
    elif case in ("m896", "m897"):  # the largest known cloud the grid form takes / the first the scan takes again
        n, m = 8192, int(case[1:])
        unknown = torch.rand(B, n, 3, generator=g) * 2 - 1
        known = torch.rand(B, m, 3, generator=g) * 2 - 1
    elif case == "sparse_shell":  # known points on a sphere, unknown ones inside: empty cells around most queries
        known = torch.nn.functional.normalize(torch.randn(B, m, 3, generator=g), dim=-1)
        unknown = unknown * 0.5
    elif case == "line":
        known = torch.zeros(B, m, 3)
        known[..., 0] = torch.rand(B, m, generator=g)
    else:  # two far-apart clusters: most cells empty, queries between them need many shells
        known = torch.rand(B, m, 3, generator=g) * 0.05
        known[:, m // 2:] += 3.0
    gd, gi = hip.three_nn(unknown.to(DEV), known.to(DEV))
    rd, ri = oracle.three_nn(unknown, known)
    assert torch.equal(gi.cpu(), ri)
    assert torch.equal(gd.cpu(), rd)


# ------------------------------------------------------------------------- interpolate / grouping

def _assert_scatter_grad(got, want, flat_idx, nbins):
    """(B, C, nbins) scatter-add gradients: bit-identical to the oracle's sequential sum wherever a destination collects
    at most 128 slots (GS_HUB_MIN of csrc/csr.hip); longer runs are summed in a fixed parallel order, i.e. to round-off."""
    for b in range(got.shape[0]):
        runs = torch.bincount(flat_idx[b], minlength=nbins)
        short = runs <= 128
        assert torch.equal(got[b][:, short], want[b][:, short])
        if not bool(short.all()):
            torch.testing.assert_close(got[b][:, ~short], want[b][:, ~short], rtol=1e-5,
                                       atol=1e-5 * float(want[b].abs().max()))


@pytest.mark.parametrize("B,C,m,n", [(2, 256, 128, 512), (2, 128, 512, 1024), (3, 5, 7, 130), (1, 33, 3, 1),
                                     (2, 3, 512, 16384), (1, 2, 64, 40000), (2, 1, 9, 300)])
def test_three_interpolate_fwd_bwd(hip, oracle, B, C, m, n):
    g = torch.Generator().manual_seed(B * 1000 + n)
    feat = torch.randn(B, C, m, generator=g)
    idx = torch.randint(0, m, (B, n, 3), generator=g)
    w = torch.rand(B, n, 3, generator=g)
    w = w / w.sum(-1, keepdim=True)
    cot = torch.randn(B, C, n, generator=g)
    fa = feat.clone().to(DEV).requires_grad_(True)
    fb = feat.clone().requires_grad_(True)
    oa = hip.three_interpolate(fa, idx.to(DEV), w.to(DEV))
    ob = oracle.three_interpolate(fb, idx, w)
    assert torch.equal(oa.cpu(), ob.detach())  # fixed 3-term order, no fma: exact
    oa.backward(cot.to(DEV))
    ob.backward(cot)
    # transpose + gather-sum accumulates in the oracle's order (ascending (i,t), mul then add): exact for every known
    # point that collects at most 128 slots; longer runs are summed by a whole workgroup in a fixed parallel order
    _assert_scatter_grad(fa.grad.cpu(), fb.grad, idx.reshape(B, -1), m)
    (g2,) = torch.autograd.grad(hip.three_interpolate(fa, idx.to(DEV), w.to(DEV)), fa, cot.to(DEV))
    assert torch.equal(g2, fa.grad)  # no atomics: bitwise reproducible run to run


@pytest.mark.parametrize("B,C,N,npnt,ns", [(2, 8, 1024, 512, 64), (2, 131, 512, 128, 64), (3, 3, 50, 7, 5),
                                          (1, 20, 9, 1, 1), (2, 3, 16384, 512, 64), (1, 2, 50000, 1100, 64),
                                          (2, 1, 40, 30, 3), (1, 6, 70000, 64, 16)])
def test_grouping_fwd_bwd(hip, oracle, B, C, N, npnt, ns):
    g = torch.Generator().manual_seed(B * 77 + N)
    feat = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, npnt, ns), generator=g)
    idx[:, :, 1:] = torch.where(torch.rand(B, npnt, ns - 1, generator=g) < 0.5, idx[:, :, :1], idx[:, :, 1:]) \
        if ns > 1 else idx[:, :, 1:]  # padded slots repeat slot 0, as ball_query emits them
    cot = torch.randn(B, C, npnt, ns, generator=g)
    fa = feat.clone().to(DEV).requires_grad_(True)
    fb = feat.clone().requires_grad_(True)
    oa = hip.grouping_operation(fa, idx.to(DEV))
    ob = oracle.grouping_operation(fb, idx)
    assert torch.equal(oa.cpu(), ob.detach())
    oa.backward(cot.to(DEV))
    ob.backward(cot)
    _assert_scatter_grad(fa.grad.cpu(), fb.grad, idx.reshape(B, -1), N)  # ascending-slot accumulation, as the oracle
    (g2,) = torch.autograd.grad(hip.grouping_operation(fa, idx.to(DEV)), fa, cot.to(DEV))
    assert torch.equal(g2, fa.grad)  # no atomics: bitwise reproducible run to run


@pytest.mark.parametrize("npnt,ns,N", [(64, 128, 300), (600, 128, 300), (40, 64, 2000)])
def test_grouping_bwd_giant_bins(hip, oracle, npnt, ns, N):
    """Dense ball queries pad with their first hit: when that is the same point for many queries, one destination
    collects thousands of slots (here npnt * (ns - 40)).  The inverse-index build sorts such a bin window segment by
    window segment; the short runs keep the oracle's accumulation order bit for bit, the giant one is summed in a fixed
    parallel order.  (600 x 128 slots do not fit the in-LDS tables: second code path.)"""
    B, C = 2, 5
    g = torch.Generator().manual_seed(npnt + ns)
    feat = torch.randn(B, C, N, generator=g)
    idx = torch.randint(0, N, (B, npnt, ns), generator=g)
    idx[:, :, 40:] = 0  # every query: 40 hits, then padding with point 0
    idx[1, :, 40:] = idx[1, :1, :1]  # second cloud: another shared first hit
    cot = torch.randn(B, C, npnt, ns, generator=g)
    fa = feat.clone().to(DEV).requires_grad_(True)
    fb = feat.clone().requires_grad_(True)
    hip.grouping_operation(fa, idx.to(DEV)).backward(cot.to(DEV))
    oracle.grouping_operation(fb, idx).backward(cot)
    # runs of more than 128 slots are summed by a whole workgroup (pieces per wave, strides per lane, fixed order): the
    # oracle's sequential sum to fp32 round-off, and the same bits from call to call (the reference accumulates with
    # atomics in no fixed order at all)
    torch.testing.assert_close(fa.grad.cpu(), fb.grad, rtol=1e-5, atol=1e-5 * float(fb.grad.abs().max()))
    (g2,) = torch.autograd.grad(hip.grouping_operation(fa, idx.to(DEV)), fa, cot.to(DEV))
    assert torch.equal(g2, fa.grad)
    # the channel-last rows path (fused set abstraction) goes through the same tables
    from torch_points3d_amd import fused
    pos = torch.rand(B, N, 3, generator=g).to(DEV)
    new_pos = pos[:, :npnt].contiguous() if npnt <= N else pos.repeat(1, 3, 1)[:, :npnt].contiguous()
    xa = feat.transpose(1, 2).contiguous().to(DEV).requires_grad_(True)
    rows = fused.group_concat(pos, new_pos, xa, idx.to(DEV), 1.0, False)
    cot_rows = torch.zeros_like(rows)
    cot_rows[:, 3:3 + C] = cot.permute(0, 2, 3, 1).reshape(-1, C).to(DEV)
    rows.backward(cot_rows)
    # runs of more than 128 slots are summed there by a whole workgroup in 16 pieces (one wave walking thousands of rows
    # held the launch): a fixed association, reproducible, but not the oracle's sequential one -- so close, not equal
    ga = xa.grad.transpose(1, 2).cpu()
    torch.testing.assert_close(ga, fb.grad, rtol=1e-5, atol=1e-5 * float(fb.grad.abs().max()))
    xb = feat.transpose(1, 2).contiguous().to(DEV).requires_grad_(True)
    fused.group_concat(pos, new_pos, xb, idx.to(DEV), 1.0, False).backward(cot_rows)
    assert torch.equal(xb.grad, xa.grad)


def test_strided_and_int32_inputs_are_normalised(hip, oracle):
    x = cloud(2, 400, 5)
    xt = x.transpose(1, 2).contiguous().transpose(1, 2)  # non-contiguous view
    assert torch.equal(hip.furthest_point_sample(xt.to(DEV), 50).cpu(), oracle.furthest_point_sample(x, 50))
    feat = torch.randn(2, 6, 400)
    idx = torch.randint(0, 400, (2, 10, 4))
    assert torch.equal(hip.grouping_operation(feat.to(DEV), idx.int().to(DEV)).cpu(),
                       oracle.grouping_operation(feat, idx))


# ------------------------------------------------------- goldens written by the reference's modules
# BASELINE.json north_star: "fp32 features within 1e-5".  What is asserted, and why in this form:
#   * every stage on the fixture's own inputs (teacher forcing), train-mode BatchNorm:      |GPU - golden| <= 1e-5 + 1e-5 |golden|
#   * the whole network in eval mode (running statistics), stage outputs:                    same bound
#   * the whole network in train mode, stage outputs, against the fp64 evaluation of the pass: |GPU - fp64| <= 2 |golden - fp64|
#     (max norm per stage): the error a stack of train-mode BatchNorms amplifies is MEASURED against what the reference's
#     own CPU pass loses, not argued
#   * gradients: element-wise on the fixtures whose forward pass stays >= 1e-3 away from every LeakyReLU kink and
#     clear of every arg-max switch (tests/golden/make_golden.py: move_off_the_kink); on the others a flipped mask makes an
#     element-wise comparison of two correct fp32 implementations ill-posed, so only their L2 error is bounded.

TIGHT = dict(rtol=1e-5, atol=1e-5)


def _cases():
    from golden_util import CASES
    return CASES


@pytest.mark.parametrize("name", ["c1_example", "small_ssg", "small_msg", "c3_charlesmsg"])
def test_kernel_goldens(hip, name):
    g = load_golden(name)
    cfg = _cases()[name]()
    cur = g["pos"].to(DEV)
    for i in range(len(cfg["npoint"])):
        fps = hip.furthest_point_sample(cur, cfg["npoint"][i])
        assert torch.equal(fps.cpu(), g["fps%d" % i])
        new = cur.gather(1, fps.unsqueeze(-1).repeat(1, 1, 3))
        for s, (r, ns) in enumerate(zip(cfg["radii"][i], cfg["nsample"][i])):
            idx, d2 = hip.ball_query(r, ns, cur, new)
            assert torch.equal(idx.cpu(), g["ball%d_%d_idx" % (i, s)])
            assert torch.equal(d2.cpu(), g["ball%d_%d_d2" % (i, s)])
        cur = new


def _maxerr(a, b):
    return float((a.double() - b.double()).abs().max())


def _golden_fp64_distance(g, k):
    """max |golden - fp64 evaluation| of a stage on the subset the fixture stores the fp64 tensor for (None: not stored)"""
    if "f64/" + k not in g:
        return None
    s_sub, first = [int(v) for v in g["meta_sub/f64/" + k]]
    s_gold = int(g["meta_sub_out"][0]) if (k in ("fc0_x", "out_x") and "meta_sub_out" in g) else 1
    gold = g[k][:1] if first else g[k]
    gold = gold[..., ::max(1, s_sub // s_gold)] if s_sub > 1 else gold
    return _maxerr(gold, torch.as_tensor(g["f64/" + k]))


# (small_ssg_tanh is left out here: Tanh is not an activation of the fused row kernels, so both of its paths are the
#  library graph; it serves the element-wise gradient test below)
ALL_MODELS = ["c1_example", "small_ssg", "small_msg", "small_ssg_slope1", "small_ssg_kinkfree", "small_msg_kinkfree",
              "c3_charlesmsg"]
# fused=False runs the reference's (B,C,np,ns) graph on MIOpen / rocBLAS around the HIP spatial kernels: the 1x1
# convolutions are then vendor-library arithmetic (measured up to 7e-5 at the K = 1280 layer), not this build's kernels,
# and are held to 1e-4 instead
LIBRARY_GRAPH = dict(rtol=1e-4, atol=1e-4)
WITH_VARIANTS = ["c1_example", "small_ssg", "small_msg", "small_ssg_kinkfree", "small_msg_kinkfree", "c3_charlesmsg"]


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", ALL_MODELS)
def test_model_goldens_teacher_forced(hip, name, fused):
    """Every stage of the network (set abstraction x2, global module, feature propagation x3, head) on the HIP path,
    fed the REFERENCE's tensors for that stage's inputs, train-mode BatchNorm: features within 1e-5.
    fused=True: channel-last HIP kernels for the grouped-MLP aggregation; fused=False: the reference's (B,C,np,ns)
    PyTorch graph around the HIP spatial kernels."""
    from golden_util import build_from_golden, head_subsample, report, run_teacher_forced
    g = load_golden(name)
    net = build_from_golden(g, name, None, device=DEV, fused=fused)  # kernels=None -> the HIP product path
    out = run_teacher_forced(net, g, DEV)
    worst = {}
    for k, v in out.items():
        got = head_subsample(g, k, v.detach()).cpu()
        worst[k] = _maxerr(got, g[k])
        tol = dict(TIGHT if fused else LIBRARY_GRAPH)
        own = _golden_fp64_distance(g, k)
        if own is not None and 2.0 * own > tol["atol"]:
            # the reference's OWN fp32 pass is further than 5e-6 from the exact (fp64) result at this stage (long
            # contractions feeding a BatchNorm over a few hundred rows): 1e-5 is then inside fp32 round-off for ANY
            # implementation, and the bound becomes twice that measured distance
            tol["atol"] = 2.0 * own
            worst[k + " (bound 2x golden-fp64 distance)"] = tol["atol"]
        torch.testing.assert_close(got, g[k], msg=lambda m, k=k: k + ": " + m, **tol)
    report("teacher_forced_max_abs_err", "%s/%s" % (name, "fused" if fused else "reference-graph"), worst)
    bn = __import__("golden_util").stage_lists(net)[0][0].mlps[0][0][1]
    torch.testing.assert_close(bn.running_mean.cpu(), g["bn_after/first_running_mean"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(bn.running_var.cpu(), g["bn_after/first_running_var"], rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", WITH_VARIANTS)
def test_model_goldens_eval_mode(hip, name, fused):
    """The whole network in eval mode (running statistics as the fixture's training pass left them), every stage
    output within 1e-5 of the reference modules' -- no teacher forcing: errors may accumulate through the stack."""
    from golden_util import build_from_golden, load_after_state, report, run_stages, variant
    g = load_golden(name)
    net = load_after_state(build_from_golden(g, name, None, device=DEV, fused=fused), g).eval()
    with torch.no_grad():
        rec = run_stages(net, g, DEV)
    worst, checked = {}, 0
    for k, v in rec.items():
        if "eval/" + k not in g:
            continue
        ref, got = variant(g, "eval/", k, v.cpu())
        worst[k] = _maxerr(got, ref)
        torch.testing.assert_close(got, ref, msg=lambda m, k=k: k + ": " + m, **TIGHT)
        checked += 1
    assert checked >= 7
    report("eval_mode_max_abs_err", "%s/%s" % (name, "fused" if fused else "reference-graph"), worst)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", WITH_VARIANTS)
def test_model_goldens_train_mode_fp64_bound(hip, name, fused):
    """The whole network in train mode, un-forced: per stage, the HIP path's distance to the fp64 evaluation of the
    same pass is at most twice the distance of the reference's own fp32 CPU pass (the golden) to it.  The distance is
    the RMS error over the stage's (subsampled) tensor; the max-norm of two fp32 passes' round-off fluctuates by more
    than 2x on the tiny nets (a stage of 120 rows), so it is recorded and bounded at 4x."""
    from golden_util import build_from_golden, report, run_stages, variant
    g = load_golden(name)
    net = build_from_golden(g, name, None, device=DEV, fused=fused)
    rec = run_stages(net, g, DEV)
    ratios = {}
    for k, v in rec.items():
        if "f64/" + k not in g:
            continue
        ref64, got = variant(g, "f64/", k, v.detach().cpu())
        s_sub = int(g["meta_sub/f64/" + k][0])
        s_gold = int(g["meta_sub_out"][0]) if (k in ("fc0_x", "out_x") and "meta_sub_out" in g) else 1
        gold = g[k][:1] if int(g["meta_sub/f64/" + k][1]) else g[k]
        gold = gold[..., ::max(1, s_sub // s_gold)] if s_sub > 1 else gold
        ref64 = torch.as_tensor(ref64)
        eg, ec = (got.double() - ref64), (gold.double() - ref64)
        rms_g, rms_c = float(eg.pow(2).mean().sqrt()), float(ec.pow(2).mean().sqrt())
        max_g, max_c = float(eg.abs().max()), float(ec.abs().max())
        ratios[k] = {"rms": [rms_g, rms_c], "max": [max_g, max_c]}
        if not fused:
            # vendor-library arithmetic (MIOpen / rocBLAS 1x1 convolutions) around the HIP kernels: recorded, not bounded
            # by the reference's round-off (measured 7x at the 1280-channel layer); sanity bound only
            assert max_g <= 1e-3 * max(1.0, float(ref64.abs().max())), k
            continue
        assert rms_g <= 2.0 * rms_c + 1e-8, "%s: rms |GPU-fp64| = %.3g vs rms |golden-fp64| = %.3g" % (k, rms_g, rms_c)
        assert max_g <= 4.0 * max_c + 1e-7, "%s: max |GPU-fp64| = %.3g vs max |golden-fp64| = %.3g" % (k, max_g, max_c)
    assert len(ratios) >= 7
    report("train_mode_err_vs_fp64_[gpu,golden]", "%s/%s" % (name, "fused" if fused else "reference-graph"), ratios)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", ["small_ssg_kinkfree", "small_msg_kinkfree", "small_ssg_tanh", "small_ssg_slope1"])
def test_model_gradients_elementwise(hip, name, fused):
    """Whole-network backward on the fixtures that are clear of LeakyReLU kinks and arg-max switches: the gradient of
    the input and of every stored parameter, element-wise."""
    from golden_util import build_from_golden, cotangent, report, run_stages, stage_lists
    g = load_golden(name)
    net = build_from_golden(g, name, None, device=DEV, fused=fused)
    x_in = g["x"].to(DEV).requires_grad_(True)
    rec = run_stages(net, g, DEV, x_in=x_in)
    (rec["out_x"] * cotangent(g).to(DEV)).sum().backward()
    worst = {}

    def check(tag, got, want):
        scale = max(1.0, float(want.abs().max()))
        worst[tag] = _maxerr(got, want) / scale
        torch.testing.assert_close(got, want, rtol=1e-4, atol=2e-5 * scale, msg=lambda m: tag + ": " + m)

    check("grad_x_in", x_in.grad.cpu(), g["grad_x_in"])
    check("grad_first_conv", stage_lists(net)[0][0].mlps[0][0][0].weight.grad.cpu(), g["grad_first_conv"])
    check("grad_last_fp_conv", stage_lists(net)[2][-1].nn[0][0].weight.grad.cpu(), g["grad_last_fp_conv"])
    for k, p in net.named_parameters():
        if "pgrad/" + k in g:
            check("pgrad/" + k, p.grad.cpu(), g["pgrad/" + k])
    report("gradient_max_err_over_scale", "%s/%s" % (name, "fused" if fused else "reference-graph"),
           {"worst": max(worst.values()), "n": len(worst)})


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", ["small_ssg_kinkfree", "small_msg_kinkfree"])
def test_stage_gradients_teacher_forced(hip, name, fused):
    """Per stage: the fixture's gradient of the stage output goes in, the gradients towards the stage's inputs come
    out -- element-wise against what the reference's stage produced on the same tensors."""
    from golden_util import build_from_golden, run_teacher_forced
    g = load_golden(name)
    net = build_from_golden(g, name, None, device=DEV, fused=fused)
    out, ins = run_teacher_forced(net, g, DEV, grads=True)
    checked = 0
    for key, inputs in ins.items():
        if "gout/" + key not in g:
            continue
        grads = torch.autograd.grad(out[key], inputs, grad_outputs=g["gout/" + key].to(DEV), allow_unused=True,
                                    retain_graph=True)
        for j, got in enumerate(grads):
            want = g.get("gin/%s/%d" % (key, j))
            if want is None:
                continue
            scale = max(1.0, float(want.abs().max()))
            torch.testing.assert_close(got.cpu(), want, rtol=1e-4, atol=2e-5 * scale,
                                       msg=lambda m, k=key, j=j: "%s/%d: %s" % (k, j, m))
            checked += 1
    assert checked >= 8


@pytest.mark.parametrize("name", ["c1_example", "small_ssg", "small_msg", "c3_charlesmsg"])
def test_model_gradients_l2_on_kinked_fixtures(hip, name):
    """LeakyReLU fixtures with pre-activations down to 1e-8 of the kink (meta_min_preact): one mask flipped by a last-bit
    GEMM difference moves, through train-mode BatchNorm backward, every gradient of its layer (measured: 1 flip in
    50 400 activations -> ~1 %), so for these only the relative L2 error is bounded; the element-wise gradient checks
    run on the conditioned twins of the same networks above."""
    from golden_util import build_from_golden, cotangent, run_stages, stage_lists
    g = load_golden(name)
    net = build_from_golden(g, name, None, device=DEV)
    x_in = g["x"].to(DEV).requires_grad_(True)
    rec = run_stages(net, g, DEV, x_in=x_in)
    target = rec["fc0_x"] if "fc0_x" in rec else rec["out_x"]
    (target * cotangent(g).to(DEV)).sum().backward()
    ga, gb = x_in.grad.cpu(), g["grad_x_in"]
    wa, wb = stage_lists(net)[2][-1].nn[0][0].weight.grad.cpu(), g["grad_last_fp_conv"]
    assert float((ga - gb).norm() / gb.norm()) < 0.25
    assert float((wa - wb).norm() / wb.norm()) < 0.25


def test_c3_full_batch_properties(hip):
    """BASELINE config 3 at its full size (pointnet2_charlesmsg, B=32, N=2048, 16 categories, 50 classes): properties
    that need no CPU reference.  Eval-mode scores of a cloud do not depend on which other clouds share its batch -- the
    property data-parallel sharding by cloud rests on -- and a second run is bit-identical."""
    from torch_points3d_amd.dense import Data
    from torch_points3d_amd.pointnet2 import PointNet2_D
    B, N = 32, 2048
    gen = torch.Generator().manual_seed(5)
    pos = (torch.rand(B, N, 3, generator=gen) * 2 - 1).to(DEV)
    x = torch.randn(B, N, 3, generator=gen).to(DEV)
    cat = torch.randint(0, 16, (B, 1), generator=gen).repeat(1, N).to(DEV)
    torch.manual_seed(1)
    net = PointNet2_D(3, 50, num_categories=16).to(DEV)
    net.train()
    scores = net(Data(pos=pos, x=x), cat)  # one training pass: running statistics move off their initial values
    assert scores.shape == (B * N, 50) and bool(torch.isfinite(scores).all())
    loss = torch.nn.functional.cross_entropy(scores, torch.randint(0, 50, (B * N,), generator=gen).to(DEV))
    loss.backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    net.eval()
    with torch.no_grad():
        full = net(Data(pos=pos, x=x), cat).view(B, N, 50)
        again = net(Data(pos=pos, x=x), cat).view(B, N, 50)
        assert torch.equal(full, again)
        for lo, hi in ((0, 4), (4, 16), (16, 32)):
            part = net(Data(pos=pos[lo:hi], x=x[lo:hi]), cat[lo:hi]).view(hi - lo, N, 50)
            torch.testing.assert_close(part, full[lo:hi], rtol=1e-5, atol=1e-5)


# ---------------------------------------------------- full BASELINE size: size-independent properties

def test_full_size_properties(hip):
    """B=32, N=16384 (BASELINE config 2 shapes): properties that need no CPU reference."""
    B, N, npnt, ns, r = 32, 16384, 512, 64, 0.2
    pos = cloud(B, N, 1234).to(DEV)
    fps = hip.furthest_point_sample(pos, npnt)
    assert fps.shape == (B, npnt) and int(fps.min()) >= 0 and int(fps.max()) < N
    assert bool((fps[:, 0] == 0).all())
    srt = fps.sort(dim=1)[0]
    assert bool((srt[:, 1:] != srt[:, :-1]).all()), "FPS repeated a point on distinct coordinates"
    new = pos.gather(1, fps.unsqueeze(-1).repeat(1, 1, 3))
    # greedy property: the distance of sample i to the earlier samples is non-increasing in i
    d = torch.cdist(new[:2, :64], new[:2, :64])
    mins = torch.stack([d[:, i, :i].min(dim=1)[0] for i in range(1, 64)], 1)
    assert bool((mins[:, 1:] <= mins[:, :-1] + 1e-6).all())

    idx, d2 = hip.ball_query(r, ns, pos, new)
    assert int(idx.min()) >= 0 and int(idx.max()) < N
    valid = d2 >= 0
    assert bool(valid[:, :, 0].all()), "every centroid is a cloud point, so its ball holds at least itself"
    nb = pos.gather(1, idx.reshape(B, -1, 1).repeat(1, 1, 3)).reshape(B, npnt, ns, 3)
    dd = ((nb - new.unsqueeze(2)) ** 2).sum(-1)
    assert bool((dd[valid] < r * r + 1e-6).all())
    torch.testing.assert_close(dd[valid], d2[valid], rtol=1e-4, atol=1e-6)
    # ascending index order among real hits; padded slots repeat slot 0
    inc = (idx[:, :, 1:] > idx[:, :, :-1]) | ~valid[:, :, 1:]
    assert bool(inc.all())
    assert bool((idx[~valid] == idx[:, :, :1].expand_as(idx)[~valid]).all())
    # idempotence / determinism: a second launch gives identical bits
    idx2, _ = hip.ball_query(r, ns, pos, new)
    assert torch.equal(idx, idx2) and torch.equal(fps, hip.furthest_point_sample(pos, npnt))
    # completeness on a sample: count of hits equals a brute-force count (capped at nsample)
    cnt = (torch.cdist(new[:1, :128], pos[:1]) ** 2 < r * r).sum(-1).clamp(max=ns)
    assert int(((valid[:1, :128].sum(-1) - cnt).abs() > 1).sum()) == 0  # cdist rounding may flip a boundary point

    dist, i3 = hip.three_nn(pos, new)
    assert bool((dist[:, :, 0] <= dist[:, :, 1]).all()) and bool((dist[:, :, 1] <= dist[:, :, 2]).all())
    w = 1.0 / (dist + 1e-8)
    w = w / w.sum(-1, keepdim=True)
    ones = torch.ones(B, 4, npnt, device=DEV)
    torch.testing.assert_close(hip.three_interpolate(ones, i3, w), torch.ones(B, 4, N, device=DEV), rtol=1e-5,
                               atol=1e-5)  # weights sum to one: constants are reproduced
    # linearity of grouping: group(a+b) == group(a)+group(b), and gather/scatter adjointness <G f, g> = <f, G^T g>
    f = torch.randn(B, 3, N, device=DEV, requires_grad=True)
    gsum = hip.grouping_operation(f, idx)
    cot = torch.randn_like(gsum)
    (grad,) = torch.autograd.grad(gsum, f, cot)
    lhs = (gsum * cot).sum().item()
    rhs = (f * grad).sum().item()
    assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(lhs))
