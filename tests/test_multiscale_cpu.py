"""Host-side multi-scale precompute (torch_points3d_amd.multiscale_cpu) against the oracle restatements (no GPU):
voxel clustering vs oracle/voxel_ref.py, radius search and k-NN vs the brute-force oracle kernels."""
import numpy as np
import torch

from oracle import tpk_ref, voxel_ref
from torch_points3d_amd.multiscale_cpu import grid_sampling_cpu, knn_cpu, radius_search_cpu


def cloud(n, clouds, seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand(n, 3, generator=g) * torch.tensor([1.0, 0.8, 0.3])
    batch = torch.sort(torch.randint(0, clouds, (n,), generator=g))[0]
    return pos, batch


def test_grid_sampling_cpu_matches_the_voxel_restatement():
    pos, batch = cloud(6000, 3, 1)
    for size in (0.05, 0.11):
        got_pos, got_batch = grid_sampling_cpu(pos, batch, size)
        ref = voxel_ref.grid_sampling_mean(pos.numpy(), size, batch=batch.numpy())
        assert np.array_equal(got_batch.numpy(), ref["batch"])
        assert np.array_equal(got_pos.numpy(), ref["pos"])  # same sequential sums, same IEEE division: bit-exact


def test_radius_search_and_knn_cpu_match_the_oracle():
    pos, batch = cloud(5000, 3, 2)
    qpos, qbatch = grid_sampling_cpu(pos, batch, 0.06)
    for r, k in ((0.1, 20), (0.04, 8)):
        got = radius_search_cpu(pos, qpos, batch, qbatch, r, k)
        ref, _ = tpk_ref.ball_query(r, k, pos, qpos, mode="partial_dense", batch_x=batch, batch_y=qbatch)
        assert torch.equal(got, ref)
    idx, d2 = knn_cpu(qpos, pos, qbatch, batch, 3)
    ridx, rd2 = tpk_ref.knn(3, qpos, pos, qbatch, batch)
    assert torch.equal(idx, ridx) and torch.equal(d2, rd2)


def _worker(q, levels, up_k, pos, batch):
    from torch_points3d_amd.kpconv_blocks import PDData
    from torch_points3d_amd.multiscale_cpu import MultiScaleTransformCPU
    torch.set_num_threads(1)  # as torch's DataLoader does in its workers
    t = MultiScaleTransformCPU.from_parameters(levels, up_k)
    out = t(PDData(pos=pos, batch=batch))
    q.put(_digest(out))


def _digest(out):
    """plain-Python fingerprint of the tables (tensors are not sent through the queue)"""
    rows = [(e.pos.numpy().tobytes(), e.idx_neighboors.numpy().tobytes()) for e in out.multiscale]
    rows += [(u.knn_idx.numpy().tobytes(), u.knn_d2.numpy().tobytes()) for u in out.upsample]
    import hashlib
    return [hashlib.sha1(a + b).hexdigest() for a, b in rows]


def test_multiscale_cpu_runs_in_forked_workers():
    """the reference runs this transform inside forked DataLoader workers (datasets/base_dataset.py:251-263)"""
    import multiprocessing as mp
    from torch_points3d_amd.kpconv_blocks import PDData
    from torch_points3d_amd.multiscale_cpu import MultiScaleTransformCPU
    pos, batch = cloud(4000, 2, 7)
    t = MultiScaleTransformCPU.from_parameters([(None, 0.08, 12), (0.06, 0.08, 12), (None, 0.15, 12), (0.12, 0.15, 12)], [1, 1])
    here = t(PDData(pos=pos, batch=batch))
    want = _digest(here)
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(q, t.levels, t.up_k, pos, batch), daemon=True) for _ in range(2)]
    try:
        for p in procs:
            p.start()
        for _ in procs:
            assert q.get(timeout=120) == want
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
