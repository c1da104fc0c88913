"""Dirichlet loss on the HIP kernels against the reference's own known answers (test/test_losses.py:16-38) and a
brute-force evaluation on random clouds."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_reference_known_answers():
    from torch_points3d_amd.losses import _variance_estimator_dense, _variance_estimator_sparse, dirichlet_loss
    pos = torch.tensor([[[0, 0, 0], [1, 0, 0], [1.1, 0, 0]]], dtype=torch.float, device=DEV)
    f = torch.tensor([[1, 1, 3]], dtype=torch.float, device=DEV)
    assert _variance_estimator_dense(1.01, pos, f).tolist() == [[0, 4, 4]]
    assert dirichlet_loss(1.01, pos, f).item() == pytest.approx(4 / 3.0)
    pos = torch.tensor([[0, 0, 0], [1, 0, 0], [1.1, 0, 0], [0, 0, 0], [1, 0, 0], [1.1, 0, 0]], dtype=torch.float, device=DEV)
    f = torch.tensor([1, 1, 3, 0, 1, 0], dtype=torch.float, device=DEV)
    batch_idx = torch.tensor([0, 0, 0, 1, 1, 1], device=DEV)
    assert _variance_estimator_sparse(1.01, pos, f, batch_idx).tolist() == [0, 4, 4, 1, 2, 1]
    assert dirichlet_loss(1.01, pos, f, batch_idx).item() == pytest.approx(sum([0, 4, 4, 1, 2, 1]) / (2 * 6))


def test_random_clouds_match_brute_force_and_autograd():
    from torch_points3d_amd.losses import DirichletLoss
    g = torch.Generator().manual_seed(0)
    B, N, r = 3, 400, 0.18  # every ball holds fewer than 32 points, so no truncation rule is involved
    pos = torch.rand(B, N, 3, generator=g)
    f = torch.randn(B, N, generator=g)
    d2 = ((pos[:, :, None] - pos[:, None]) ** 2).sum(-1)
    assert int((d2 < r * r).sum(-1).max()) <= 32
    fr = f.clone().requires_grad_(True)
    ref = 0.5 * ((((fr[:, :, None] - fr[:, None]) ** 2) * (d2 < r * r)).sum(-1)).mean()
    ref.backward()
    fd = f.to(DEV).requires_grad_(True)
    loss = DirichletLoss(r)(pos.to(DEV), fd)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(fd.grad.cpu(), fr.grad, rtol=1e-5, atol=1e-6)
    # same clouds in the sparse layout
    batch = torch.arange(B).repeat_interleave(N).to(DEV)
    sparse = DirichletLoss(r)(pos.reshape(-1, 3).to(DEV), f.reshape(-1).to(DEV), batch)
    torch.testing.assert_close(sparse.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
