"""BASELINE configs 4 and 5 at their FULL sizes inside the GPU suite (the other tests stop at 20 000 / 200 000 points):
size-independent properties checked against the brute-force oracle on samples.

* config 4: KPConv unet_4 forward on one cloud of 65 536 points (one point per 0.02 voxel): every level's neighbour
  table against the oracle's radius search on a query sample (indices bit-exact, -1 shadows trailing), the sampled
  clouds against the numpy grid-sampling restatement, run-to-run determinism of the scores.
* config 5: random subsample + exact 16-NN on a 10^6-point scene: (idx, dist2) of a 4096-query sample bit-exact
  against the brute-force oracle, distances ascending, every neighbour inside its own cloud."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_config4_kpconv_unet4_full_size(oracle):
    from bench_kpconv import synthetic_cloud
    from oracle import voxel_ref
    from torch_points3d_amd.kpconv_blocks import PDData, SimpleBlock
    from torch_points3d_amd.kpconv_unet import KPConv
    n = 65536
    torch.manual_seed(0)
    model = KPConv("unet", input_nc=3, in_feat=64, in_grid_size=0.02, num_layers=4, output_nc=13).to(DEV).eval()
    pos, batch = synthetic_cloud(n, 1, 0.02)
    x = torch.cat([torch.ones(n, 1), torch.randn(n, 3, generator=torch.Generator().manual_seed(1))], 1)
    levels = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: levels.append(
        (mod, inp[0].pos.cpu(), inp[0].batch.cpu(), out.pos.cpu(), out.batch.cpu(), out.idx_neighboors.cpu())))
        for m in model.modules() if isinstance(m, SimpleBlock)]
    data = lambda: PDData(pos=pos.to(DEV), batch=batch.to(DEV), x=x.to(DEV))  # noqa: E731
    with torch.no_grad():
        a = model(data()).x
        first = list(levels)
        del levels[:]
        b = model(data()).x
    for h in hooks:
        h.remove()
    assert a.shape == (n, 13) and bool(torch.isfinite(a).all())
    assert torch.equal(a, b), "two forward passes of the same cloud differ"  # no atomics anywhere on the path
    assert len(first) == 10
    g = torch.Generator().manual_seed(2)
    sizes = []
    for mod, s_pos, s_batch, q_pos, q_batch, idx in first:
        nq, width = idx.shape
        sizes.append(nq)
        r, max_num = mod.neighbour_finder._radius, mod.neighbour_finder._max_num_neighbors
        assert width == max_num
        sel = torch.randperm(nq, generator=g)[:512].sort()[0]
        want = oracle.ball_query(r, max_num, s_pos, q_pos[sel].contiguous(), mode="partial_dense", batch_x=s_batch,
                                 batch_y=q_batch[sel].contiguous())[0]
        assert torch.equal(idx[sel], want), "neighbour table differs from the brute-force search (radius %g)" % r
        valid = idx >= 0
        assert bool((valid[:, 1:] <= valid[:, :-1]).all())  # -1 shadows only behind the real neighbours
        assert int(idx.max()) < s_pos.shape[0]
        if mod.is_strided:  # the sampled cloud: GridSampling3D(mode="mean") of the block's input
            ref = voxel_ref.grid_sampling_mean(s_pos.numpy(), mod.sampler._grid_size, batch=s_batch.numpy())
            assert q_pos.shape[0] == ref["pos"].shape[0]
            assert torch.equal(q_pos, torch.from_numpy(ref["pos"])) and torch.equal(q_batch, torch.from_numpy(ref["batch"]))
    assert sizes[0] == n and sizes[-1] < n // 50 and all(p >= q for p, q in zip(sizes, sizes[1:]))


def test_config5_knn_at_one_million_points(oracle, hip):
    from bench_knn import room
    from torch_points3d_amd.randla import RandomSampler
    n, k = 1000000, 16
    pos_cpu = room(n)
    pos = pos_cpu.to(DEV)
    batch = torch.zeros(n, dtype=torch.long, device=DEV)
    torch.manual_seed(11)
    idx = RandomSampler(ratio=0.25)(pos, batch=batch)
    assert idx.shape == (n // 4,) and int(idx.min()) >= 0 and int(idx.max()) < n
    q = pos[idx]
    nbr, d2 = hip.knn(k, pos, q, batch, batch[idx])
    assert nbr.shape == (n // 4, k) and int(nbr.min()) >= 0 and int(nbr.max()) < n
    assert bool((d2[:, 1:] >= d2[:, :-1]).all())  # closest first
    assert bool((d2[:, 0] == 0).all())            # every query is a point of the cloud: itself (or a duplicate) first
    sel = torch.randperm(n // 4, generator=torch.Generator().manual_seed(12))[:4096]
    want_idx, want_d2 = oracle.knn(k, pos_cpu, q.cpu()[sel].contiguous())
    assert torch.equal(nbr.cpu()[sel], want_idx) and torch.equal(d2.cpu()[sel], want_d2)
    nbr2, d22 = hip.knn(k, pos, q, batch, batch[idx])
    assert torch.equal(nbr, nbr2) and torch.equal(d2, d22)  # deterministic
