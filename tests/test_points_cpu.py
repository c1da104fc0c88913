"""torch_points_kernels.points_cpu (product CPU library, torch_points3d_amd/csrc_cpu/points_cpu.c) against brute force
on the call shapes of the reference's data transforms and registration dataset builders
(core/data_transform/transforms.py:805,853,887-890,919,1044; datasets/registration/utils.py:150-166,286;
datasets/registration/base_siamese_dataset.py:133-135; datasets/registration/basetest.py:361)."""
import multiprocessing as mp

import pytest
import torch

from torch_points_kernels import points_cpu


def brute(support, query, radius):
    d2 = ((query[:, None, :] - support[None, :, :]) ** 2)
    d2 = (d2[..., 0] + d2[..., 1]) + d2[..., 2]
    return d2, d2 < radius * radius


def clouds(n, nq, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    s = torch.rand(n, 3, generator=g) * scale
    q = s[torch.randperm(n, generator=g)[:nq]] + 0.01 * torch.randn(nq, 3, generator=g) if nq <= n else \
        torch.rand(nq, 3, generator=g) * scale
    return s, q.contiguous()


@pytest.mark.parametrize("n,nq,r,max_num", [(500, 500, 0.15, 16), (2000, 100, 0.08, -1), (300, 40, 2.0, 7), (50, 50, 0.0001, 4),
                                            (1, 5, 0.5, 3), (4000, 4000, 0.04, -1)])
@pytest.mark.parametrize("sort", [False, True])
def test_ball_query_matrix_mode(n, nq, r, max_num, sort):
    """mode 0 (RandomWalkDropout transforms.py:805 with max_num, DensityFilter :1044 with max_num=-1)"""
    s, q = clouds(n, nq, n + nq)
    if n == nq:
        q = s.clone()  # the transforms query the cloud with itself
    ind, dist = points_cpu.ball_query(s, q, radius=r, max_num=max_num, mode=0, sorted=sort)
    d2, inside = brute(s, q, r)
    cnt = inside.sum(1)
    width = max_num if max_num > 0 else int(cnt.max())
    assert ind.shape == (nq, width) and dist.shape == (nq, width) and ind.dtype == torch.int64
    for i in range(nq):
        hits = inside[i].nonzero().flatten()
        if sort:
            key = d2[i, hits]
            order = sorted(range(len(hits)), key=lambda j: (float(key[j]), int(hits[j])))
            hits = hits[order]
        want = hits[:width]
        assert torch.equal(ind[i, :len(want)], want)
        assert torch.equal(dist[i, :len(want)], d2[i, want])
        assert bool((ind[i, len(want):] == -1).all()) and bool((dist[i, len(want):] == -1).all())
    # DensityFilter's neighbour count: real non-self neighbours have dist > 0
    if n == nq:
        assert torch.equal((dist > 0).sum(1), (cnt - 1).clamp(max=width - 1 if max_num > 0 else 10 ** 9)) or max_num > 0


@pytest.mark.parametrize("sort", [False, True])
def test_ball_query_pair_mode(sort):
    """mode 1 with max_num=-1: every (support, query) pair inside the ball (SphereDropout :853, SphereCrop :919,
    PatchExtractor utils.py:286); column 0 indexes the SUPPORT cloud"""
    s, q = clouds(3000, 7, 11)
    ind, dist = points_cpu.ball_query(s, q, radius=0.2, max_num=-1, mode=1, sorted=sort)
    d2, inside = brute(s, q, 0.2)
    assert ind.shape[1] == 2 and dist.shape == (ind.shape[0], 1) and ind.shape[0] == int(inside.sum())
    got = set(map(tuple, ind.tolist()))
    want = set((int(b), int(a)) for a, b in inside.nonzero().tolist())
    assert got == want
    assert torch.equal(dist[:, 0], d2[ind[:, 1], ind[:, 0]])
    assert bool((ind[1:, 1] >= ind[:-1, 1]).all())  # grouped by query, queries ascending
    # SphereCrop: one centre, the points inside survive
    centre = s[5].view(1, 3)
    ind1, dist1 = points_cpu.ball_query(s, centre, radius=0.3, max_num=-1, mode=1)
    keep = ind1[dist1[:, 0] > 0][:, 0]
    ref = (brute(s, centre, 0.3)[1][0]).nonzero().flatten()
    assert set(keep.tolist()) == set(ref.tolist()) - {5}


def test_ball_query_closest_match_pairs():
    """compute_overlap_and_matches (datasets/registration/utils.py:150-166): mode 1, max_num = num_pos, sorted=True ->
    each query's closest support point within max_distance_overlap"""
    s, q = clouds(2500, 900, 3)
    pair, dist = points_cpu.ball_query(s, q, radius=0.05, max_num=1, mode=1, sorted=True)
    d2, inside = brute(s, q, 0.05)
    has = inside.any(1)
    assert pair.shape[0] == int(has.sum())
    masked = torch.where(inside, d2, torch.full_like(d2, float("inf")))
    best = masked.argmin(1)
    assert torch.equal(pair[:, 1], has.nonzero().flatten())
    assert torch.equal(pair[:, 0], best[has])
    assert torch.equal(dist[:, 0], masked.min(1)[0][has])


@pytest.mark.parametrize("B,n,nq,k", [(1, 5000, 1, 1), (2, 700, 300, 8), (1, 3, 4, 5), (1, 20000, 50, 16)])
def test_dense_knn(B, n, nq, k):
    """basetest.py:361: dense_knn(pos.unsqueeze(0), centre, k=1); general k against brute force"""
    g = torch.Generator().manual_seed(B * n + k)
    s = torch.rand(B, n, 3, generator=g) * torch.tensor([4.0, 2.0, 0.5])
    q = torch.rand(B, nq, 3, generator=g) * torch.tensor([4.4, 2.2, 0.7]) - 0.1  # some queries outside the box
    ind, dist = points_cpu.dense_knn(s, q, k)
    assert ind.shape == (B, nq, k) and dist.shape == (B, nq, k)
    for b in range(B):
        d2, _ = brute(s[b], q[b], 1.0)
        kk = min(k, n)
        for i in range(nq):
            order = sorted(range(n), key=lambda j: (float(d2[i, j]), j))[:kk]
            assert ind[b, i, :kk].tolist() == order
            assert torch.equal(dist[b, i, :kk], d2[i, order])
        assert bool((ind[b, :, kk:] == -1).all())


def _child(q):
    s, qq = clouds(1500, 200, 5)
    points_cpu.set_num_threads(3)
    ind, dist = points_cpu.ball_query(s, qq, radius=0.1, max_num=8, mode=0)
    q.put((int(ind.sum()), float(dist.sum())))


def test_usable_in_forked_workers_after_threaded_use_in_the_parent():
    """DataLoader workers are forked (datasets/base_dataset.py:251-263): the library keeps no thread pool alive"""
    s, qq = clouds(1500, 200, 5)
    points_cpu.set_num_threads(4)
    try:
        ind, dist = points_cpu.ball_query(s, qq, radius=0.1, max_num=8, mode=0)
        ctx = mp.get_context("fork")
        q = ctx.Queue()
        procs = [ctx.Process(target=_child, args=(q,), daemon=True) for _ in range(2)]
        for p in procs:
            p.start()
        got = [q.get(timeout=60) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
            assert p.exitcode == 0
        assert all(g == (int(ind.sum()), float(dist.sum())) for g in got)
    finally:
        points_cpu.set_num_threads(1)


def test_library_has_no_gpu_or_openmp_dependency():
    import subprocess
    from torch_points3d_amd import build
    out = subprocess.run(["ldd", build.build_cpu_library()], capture_output=True, text=True).stdout
    assert "amdhip" not in out and "gomp" not in out and "hsa" not in out


def test_gpu_tensors_are_refused():
    with pytest.raises(RuntimeError):
        points_cpu.ball_query(torch.zeros(2, 3, device="meta"), torch.zeros(1, 3), radius=1.0, max_num=1)


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), -float("inf")])
def test_non_finite_coordinates_raise_instead_of_hanging(bad):
    """one NaN / Inf point made the grid's cell-coarsening loop spin forever inside a DataLoader worker."""
    import ctypes
    from torch_points_kernels import points_cpu
    pts = torch.rand(100, 3)
    pts[17, 1] = bad
    q = torch.rand(5, 3)
    with pytest.raises(ValueError):
        points_cpu.ball_query(pts, q, radius=0.2, max_num=8)
    with pytest.raises(ValueError):
        points_cpu.dense_knn(pts.unsqueeze(0), q.unsqueeze(0), 3)
    # the C entry point itself refuses (NULL), whatever the caller checked
    h = points_cpu._lib()
    assert not h.tp3d_cpu_grid_build(pts.data_ptr(), pts.shape[0], ctypes.c_float(0.2))
    # finite ends whose extent overflows float
    far = torch.tensor([[3e38, 0.0, 0.0], [-3e38, 0.0, 0.0]])
    assert not h.tp3d_cpu_grid_build(far.data_ptr(), 2, ctypes.c_float(0.2))
