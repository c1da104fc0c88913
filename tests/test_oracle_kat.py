"""Pins the CPU oracle (oracle/tpk_ref_cpu.c) on the reference's own known answers and on an independent
numpy restatement.  CPU only."""
import numpy as np
import pytest
import torch


def test_dirichlet_dense_kat(oracle):
    # reference test/test_losses.py:16-24 through core/losses/dirichlet_loss.py:51-55
    pos = torch.tensor([[[0, 0, 0], [1, 0, 0], [1.1, 0, 0]]], dtype=torch.float)
    f = torch.tensor([[1, 1, 3]], dtype=torch.float)
    nei = oracle.ball_query(1.01, 32, pos, pos, sort=True)[0].reshape(1, -1).long()
    fn = f.gather(1, nei).reshape(1, 3, -1)
    var = ((f.unsqueeze(-1).repeat(1, 1, fn.shape[-1]) - fn) ** 2).sum(-1)
    torch.testing.assert_close(var, torch.tensor([[0.0, 4.0, 4.0]]))
    assert abs(0.5 * var.mean().item() - 4 / 3.0) < 1e-6  # dirichlet_loss = 1/2 * mean(var) (test_losses.py:23-24)
    # closest first incl. self, strict inside (1.0 in, 1.1 out), padded with the closest
    idx, d2 = oracle.ball_query(1.01, 4, pos, pos, sort=True)
    assert idx[0].tolist() == [[0, 1, 0, 0], [1, 2, 0, 1], [2, 1, 2, 2]]
    assert d2[0, 0].tolist() == [0.0, 1.0, -1.0, -1.0]


def test_fps_kat(oracle):
    # reference test/test_fps.py:35-42: start at point 0, then farthest: {0, 3, 4}
    pos = torch.tensor([[[0, 0, 0], [0.5, 0.5, 0], [0.4, 0.2, 0], [2, 2, 2], [-1, -2, -0.01]]]).float()
    assert oracle.furthest_point_sample(pos, 3)[0].tolist() == [0, 3, 4]
    with pytest.raises(ValueError):
        oracle.furthest_point_sample(pos, 6)


def test_dense_padding_rule(oracle):
    # reference core/spatial_ops/neighbour_finder.py:166-172: unfilled slots repeat slot 0
    x = torch.tensor([[[0, 0, 0], [5, 5, 5], [0.1, 0, 0], [0.2, 0, 0], [9, 9, 9]]]).float()
    y = torch.tensor([[[0.05, 0, 0], [100, 100, 100]]]).float()
    idx, d2 = oracle.ball_query(0.5, 4, x, y)
    assert idx[0, 0].tolist() == [0, 2, 3, 0]
    assert d2[0, 0, 3].item() == -1.0
    assert idx[0, 1].tolist() == [0, 0, 0, 0] and d2[0, 1].tolist() == [-1.0] * 4  # empty ball
    idx, _ = oracle.ball_query(0.5, 2, x, y)  # more hits than slots: first nsample in index order
    assert idx[0, 0].tolist() == [0, 2]
    idx, _ = oracle.ball_query(0.5, 3, x, y)  # exactly nsample
    assert idx[0, 0].tolist() == [0, 2, 3]


def test_sorted_truncation_keeps_closest(oracle):
    x = torch.tensor([[[0.3, 0, 0], [0.2, 0, 0], [0.1, 0, 0], [0.1, 0, 0]]]).float()
    y = torch.zeros(1, 1, 3)
    idx, d2 = oracle.ball_query(1.0, 2, x, y, sort=True)
    assert idx[0, 0].tolist() == [2, 3]  # ties by index
    idx, _ = oracle.ball_query(1.0, 2, x, y, sort=False)
    assert idx[0, 0].tolist() == [0, 1]


def test_partial_dense_shadow(oracle):
    # -1 shadow padding: reference core/common_modules/gathering.py:10, datasets/multiscale_data.py:104-130
    x = torch.tensor([[0, 0, 0], [0.1, 0, 0], [0, 0, 0], [0.1, 0, 0], [0.2, 0, 0]]).float()
    bx = torch.tensor([0, 0, 1, 1, 1])
    y = torch.tensor([[0, 0, 0], [0, 0, 0]]).float()
    by = torch.tensor([0, 1])
    idx, d2 = oracle.ball_query(0.15, 3, x, y, mode="partial_dense", batch_x=bx, batch_y=by)
    assert idx.tolist() == [[0, 1, -1], [2, 3, -1]]
    assert d2[0, 2].item() == -1.0
    with pytest.raises(Exception):
        oracle.ball_query(0.15, 3, x, y, mode="partial_dense")
    with pytest.raises(Exception):
        oracle.ball_query(0.15, 3, x[None], y[None], mode="dense", batch_x=bx, batch_y=by)


def test_three_nn_ties_and_sqrt(oracle):
    known = torch.tensor([[[1, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]]]).float()
    unknown = torch.zeros(1, 1, 3)
    dist, idx = oracle.three_nn(unknown, known)
    assert idx[0, 0].tolist() == [0, 1, 2]
    assert dist[0, 0].tolist() == [1.0, 1.0, 2.0]
    with pytest.raises(ValueError):
        oracle.three_nn(unknown, known[:, :2])


# ---------------------------------------------------------------------------------------------------
# independent restatement in numpy (no shared code with the C file) on random small cases


def _np_sqdist(a, b):
    d = (a[:, None, :] - b[None, :, :]).astype(np.float32)
    return ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(np.float32)


def _np_fps(p, k):
    n = p.shape[0]
    mind = np.full(n, 1e10, np.float32)
    sel = [0]
    for _ in range(1, k):
        d = _np_sqdist(p, p[sel[-1]][None])[:, 0]
        mind = np.minimum(mind, d)
        sel.append(int(np.argmax(mind)))  # first max = lowest index
    return sel


def _np_ball(x, y, r, ns, sort):
    d = _np_sqdist(y, x)
    r2 = np.float32(r) * np.float32(r)
    out = np.zeros((y.shape[0], ns), np.int64)
    for j in range(y.shape[0]):
        hits = np.nonzero(d[j] < r2)[0]
        if sort:
            hits = hits[np.argsort(d[j][hits], kind="stable")]
        hits = hits[:ns]
        if len(hits):
            out[j, : len(hits)] = hits
            out[j, len(hits):] = hits[0]
    return out


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_against_numpy_restatement(oracle, seed):
    g = torch.Generator().manual_seed(seed)
    B, N, npnt, ns = 2, 257, 50, 9
    pos = torch.rand(B, N, 3, generator=g)
    pos[:, 5] = pos[:, 17]  # duplicate point: exact ties
    fps = oracle.furthest_point_sample(pos, npnt)
    for b in range(B):
        assert fps[b].tolist() == _np_fps(pos[b].numpy(), npnt)
    new = pos.gather(1, fps.unsqueeze(-1).repeat(1, 1, 3))
    for sort in (False, True):
        idx, d2 = oracle.ball_query(0.3, ns, pos, new, sort=sort)
        for b in range(B):
            assert np.array_equal(idx[b].numpy(), _np_ball(pos[b].numpy(), new[b].numpy(), 0.3, ns, sort))
    dist, idx3 = oracle.three_nn(pos, new)
    for b in range(B):
        d = _np_sqdist(pos[b].numpy(), new[b].numpy())
        order = np.argsort(d, axis=1, kind="stable")[:, :3]
        assert np.array_equal(idx3[b].numpy(), order)
        np.testing.assert_allclose(dist[b].numpy(), np.sqrt(np.take_along_axis(d, order, 1)), rtol=1e-6)


def test_interpolate_and_group_against_torch(oracle):
    g = torch.Generator().manual_seed(5)
    B, C, m, n = 2, 7, 33, 90
    feat = torch.randn(B, C, m, generator=g, requires_grad=True)
    idx = torch.randint(0, m, (B, n, 3), generator=g)
    w = torch.rand(B, n, 3, generator=g)
    out = oracle.three_interpolate(feat, idx, w)
    ref = sum(w[:, None, :, t] * feat.gather(2, idx[:, None, :, t].expand(B, C, n)) for t in range(3))
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    cot = torch.randn(B, C, n, generator=g)
    (ga,) = torch.autograd.grad(out, feat, cot, retain_graph=True)
    (gb,) = torch.autograd.grad(ref, feat, cot)
    torch.testing.assert_close(ga, gb, rtol=1e-5, atol=1e-5)

    gi = torch.randint(0, m, (B, 11, 5), generator=g)
    grouped = oracle.grouping_operation(feat, gi)
    ref = feat.gather(2, gi.view(B, 1, -1).repeat(1, C, 1)).view(B, C, 11, 5)  # tpk 0.7.0's own definition
    assert torch.equal(grouped, ref)
    cot = torch.randn(B, C, 11, 5, generator=g)
    (ga,) = torch.autograd.grad(grouped, feat, cot, retain_graph=True)
    (gb,) = torch.autograd.grad(ref, feat, cot)
    torch.testing.assert_close(ga, gb, rtol=1e-5, atol=1e-5)
