"""torch_points_kernels.region_grow / instance_iou -- the two remaining names the reference imports from the kernels package
(models/panoptic/pointgroup.py:3, core/losses/panoptic_losses.py:3, metrics/panoptic_tracker.py:12)."""
import numpy as np
import pytest
import torch

import torch_points_kernels as tp

DEV = "cuda:0"


def _instance_iou_loss(clusters, scores, inst, batch, lo=0.25, hi=0.75):
    """core/losses/panoptic_losses.py:25-48 restated around the function under test"""
    ious = tp.instance_iou(clusters, inst, batch).max(1)[0]
    lower, higher = ious < lo, ious > hi
    middle = ~lower & ~higher
    shat = torch.zeros_like(ious)
    shat[higher] = 1
    shat[middle] = (ious[middle] - lo) / (hi - lo)
    return torch.nn.functional.binary_cross_entropy(scores, shat)


def test_instance_iou_reference_known_answers():
    # reference test/test_pointgroup.py:28-39
    clusters = [torch.tensor([0, 1, 2]), torch.tensor([3, 4])]
    scores = torch.tensor([1, 0]).float()
    batch = torch.tensor([0, 0, 0, 0, 0])
    assert _instance_iou_loss(clusters, scores, torch.tensor([1, 1, 1, 0, 0]), batch).item() == 0
    assert abs(_instance_iou_loss(clusters, scores, torch.tensor([1, 1, 1, 2, 2]), batch).item() - 50) < 1e-5


def test_instance_iou_against_set_arithmetic_over_several_clouds():
    g = torch.Generator().manual_seed(1)
    sizes = [40, 25, 60]
    batch = torch.cat([torch.full((n,), i) for i, n in enumerate(sizes)])
    inst = torch.cat([torch.randint(0, k + 1, (n,), generator=g) for n, k in zip(sizes, [3, 1, 5])])
    n = batch.numel()
    clusters = []
    for _ in range(9):
        cloud = int(torch.randint(0, 3, (1,), generator=g))
        members = torch.nonzero(batch == cloud).view(-1)
        clusters.append(members[torch.randperm(members.numel(), generator=g)[: int(torch.randint(1, 20, (1,), generator=g))]])
    got = tp.instance_iou(clusters, inst, batch)
    per_cloud = [int(inst[batch == s].max()) for s in range(3)]
    assert got.shape == (9, sum(per_cloud))
    col = 0
    for s in range(3):
        for k in range(1, per_cloud[s] + 1):
            gt = set(torch.nonzero((batch == s) & (inst == k)).view(-1).tolist())
            for c, cl in enumerate(clusters):
                a = set(cl.tolist())
                want = len(a & gt) / max(len(a | gt), 1)
                assert abs(float(got[c, col]) - want) < 1e-6, (c, col)
            col += 1
    assert n == sum(sizes)


def _grow_reference(pos, labels, batch, ignore, radius, nsample, min_size):
    """plain-Python restatement: per label and cloud, brute-force neighbour rows (first nsample within radius, ascending
    index), clusters grown from the lowest unvisited index"""
    out = []
    pos = pos.double().numpy()
    for label in sorted(set(labels.tolist()) - set(ignore)):
        sel = np.nonzero((labels == label).numpy())[0]
        p, b = pos[sel], batch.numpy()[sel]
        d2 = ((p[:, None, :].astype(np.float32) - p[None, :, :].astype(np.float32)) ** 2)
        d2 = (d2[..., 0] + d2[..., 1]) + d2[..., 2]
        ok = (d2 < np.float32(radius) * np.float32(radius)) & (b[:, None] == b[None, :])
        rows = [np.nonzero(ok[i])[0][:nsample] for i in range(len(sel))]
        seen = np.zeros(len(sel), bool)
        for i in range(len(sel)):
            if seen[i]:
                continue
            seen[i] = True
            stack, members = [i], [i]
            while stack:
                k = stack.pop()
                for nb in rows[k]:
                    if not seen[nb]:
                        seen[nb] = True
                        stack.append(nb)
                        members.append(nb)
            if len(members) >= min_size:
                out.append(frozenset(sel[members].tolist()))
    return out


@pytest.mark.parametrize("seed", [0, 1])
def test_region_grow_matches_plain_restatement_on_cpu(seed):
    g = torch.Generator().manual_seed(seed)
    centres = torch.rand(12, 3, generator=g) * 4
    pos = torch.cat([c + torch.randn(int(k), 3, generator=g) * 0.05 for c, k in zip(centres, torch.randint(3, 60, (12,), generator=g))])
    n = pos.shape[0]
    batch = (torch.arange(n) * 3 // n).long()  # three clouds, sorted
    labels = torch.randint(0, 4, (n,), generator=g)
    got = tp.region_grow(pos, labels, batch, ignore_labels=[2], radius=0.12, nsample=32, min_cluster_size=4)
    want = _grow_reference(pos, labels, batch, [2], 0.12, 32, 4)
    assert [frozenset(c.tolist()) for c in got] == want
    for c in got:  # one label, one cloud per cluster; never an ignored label
        assert labels[c].unique().numel() == 1 and int(labels[c][0]) != 2 and batch[c].unique().numel() == 1
    assert tp.region_grow(pos[:0], labels[:0], batch[:0]) == []


@pytest.mark.gpu
def test_region_grow_on_device_tensors_gives_the_same_clusters():
    g = torch.Generator().manual_seed(3)
    centres = torch.rand(30, 3, generator=g) * 6
    pos = torch.cat([c + torch.randn(int(k), 3, generator=g) * 0.04 for c, k in zip(centres, torch.randint(20, 200, (30,), generator=g))])
    n = pos.shape[0]
    batch = (torch.arange(n) * 2 // n).long()
    labels = torch.randint(0, 3, (n,), generator=g)
    cpu = tp.region_grow(pos, labels, batch, ignore_labels=torch.tensor([0]), radius=0.1, nsample=48, min_cluster_size=10)
    dev = tp.region_grow(pos.to(DEV), labels.to(DEV), batch.to(DEV), ignore_labels=torch.tensor([0]).to(DEV), radius=0.1,
                         nsample=48, min_cluster_size=10)
    assert len(cpu) > 5 and [frozenset(c.tolist()) for c in cpu] == [frozenset(c.tolist()) for c in dev]
    assert all(c.device.type == "cuda" for c in dev)
    iou = tp.instance_iou(dev, labels.to(DEV) + 1, batch.to(DEV))
    assert iou.shape[0] == len(dev) and float(iou.max()) <= 1.0 and float(iou.min()) >= 0.0
