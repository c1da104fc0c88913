"""``torch_points_kernels.instance_iou`` (reference core/losses/panoptic_losses.py:3,37, metrics/panoptic_tracker.py:12,194):
intersection over union of every predicted cluster with every ground-truth instance.

    instance_iou(clusters: list of LongTensors of point indices, instance_labels (N,), batch (N,)) -> (len(clusters), G)

Ground-truth instances are numbered 1..g_s inside every cloud s (0 = no instance); the columns run over the clouds'
instances one cloud after the other, G = sum_s g_s with g_s the largest label of cloud s -- the layout the reference's
tracker undoes with `gt_ids + 1 - instance_offsets[sample]` (panoptic_tracker.py:194-207).  Known answers:
test/test_pointgroup.py:28-39 through instance_iou_loss (tests/test_cluster_cpu.py).

A counting problem, not a hot path: one bincount over (cluster, instance) keys on whatever device the labels live on.
"""
import torch


def instance_iou(instance_idx, gt_instances, batch):
    dev = gt_instances.device
    gt = gt_instances.long()
    b = batch.long().to(dev)
    nb = int(b.max()) + 1 if b.numel() else 0
    per_cloud = torch.zeros(nb, dtype=torch.long, device=dev)
    if gt.numel():
        per_cloud.scatter_reduce_(0, b, gt, reduce="amax", include_self=True)  # largest instance label of every cloud
    offsets = torch.cumsum(per_cloud, 0) - per_cloud
    G = int(per_cloud.sum()) if nb else 0
    nc = len(instance_idx)
    if nc == 0 or G == 0:
        return torch.zeros((nc, G), dtype=torch.float32, device=dev)
    column = torch.where(gt > 0, offsets[b] + gt - 1, torch.full_like(gt, -1))  # global instance column of every point
    gt_size = torch.bincount(column[column >= 0], minlength=G).float()
    sizes = torch.tensor([c.numel() for c in instance_idx], dtype=torch.long, device=dev)
    flat = torch.cat([c.to(dev).long().view(-1) for c in instance_idx])
    owner = torch.repeat_interleave(torch.arange(nc, device=dev), sizes)
    col = column[flat]
    keep = col >= 0
    inter = torch.bincount(owner[keep] * G + col[keep], minlength=nc * G).view(nc, G).float()
    union = sizes.float().unsqueeze(1) + gt_size.unsqueeze(0) - inter
    return inter / union.clamp(min=1.0)
