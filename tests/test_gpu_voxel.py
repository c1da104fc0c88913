"""GridSampling3D on the device (csrc/voxel.hip, torch_points3d_amd/grid_sampling.py) against oracle/voxel_ref.py:
cluster ids, representative indices and majority labels bit-exact, voxel means bit-exact in fp32 (same summation
order and the same IEEE division), plus the reference's own test properties (test/test_grid_sampling.py)."""
import numpy as np
import pytest
import torch

from oracle import voxel_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cloud(n, clouds, seed, scale=1.0):
    rs = np.random.RandomState(seed)
    pos = (rs.rand(n, 3) * scale).astype(np.float32)
    if clouds > 1:
        batch = np.sort(rs.randint(0, clouds, n)).astype(np.int64)
    else:
        batch = None
    return pos, batch


@pytest.mark.parametrize("n,clouds,size", [(1, 1, 0.1), (5, 1, 0.04), (1000, 1, 0.05), (4096, 3, 0.03), (65536, 1, 0.04),
                                           (65536, 4, 0.02), (300000, 2, 0.01), (20000, 1, 10.0), (777, 5, 1e-3)])
def test_cluster_matches_oracle(n, clouds, size):
    from torch_points3d_amd import grid_sampling as gs
    pos, batch = cloud(n, clouds, n + clouds)
    ref_cluster, ref_perm = voxel_ref.consecutive_cluster(
        voxel_ref.grid_cluster_key(voxel_ref.voxel_coords(pos, size), batch))
    tb = None if batch is None else torch.from_numpy(batch).to(DEV)
    cluster, upi, order, start = gs.voxel_cluster(torch.from_numpy(pos).to(DEV), tb, size)
    assert np.array_equal(cluster.cpu().numpy(), ref_cluster)
    assert np.array_equal(upi.cpu().numpy(), ref_perm)
    # member lists: ascending point index inside every cluster, clusters in id order
    o, s = order.cpu().numpy(), start.cpu().numpy()
    assert s[0] == 0 and s[-1] == n and np.all(np.diff(s) > 0)
    assert np.array_equal(ref_cluster[o], np.repeat(np.arange(len(s) - 1), np.diff(s)))
    same = np.diff(ref_cluster[o]) == 0
    assert np.all(np.diff(o)[same] > 0)


@pytest.mark.parametrize("n,clouds,size,C", [(1000, 1, 0.08, 1), (30000, 3, 0.05, 7), (65536, 2, 0.04, 64),
                                             (5000, 1, 0.5, 130)])
def test_mean_mode_matches_oracle_bit_exact(n, clouds, size, C):
    from torch_points3d_amd.grid_sampling import GridSampling3D
    from torch_points3d_amd.kpconv_blocks import PDData
    pos, batch = cloud(n, clouds, 7 * n + C)
    rs = np.random.RandomState(n)
    x = rs.randn(n, C).astype(np.float32)
    y = rs.randint(-2, 11, n).astype(np.int64)
    inst = rs.randint(0, 5000, n).astype(np.int64)  # wide label range: exercises the pairwise-count branch
    ref = voxel_ref.grid_sampling_mean(pos, size, batch=batch, x=x, y=y)
    ref_inst = voxel_ref.majority_label(inst, ref["cluster"], ref["pos"].shape[0])
    data = PDData(pos=torch.from_numpy(pos).to(DEV), x=torch.from_numpy(x).to(DEV), y=torch.from_numpy(y).to(DEV),
                  instance_labels=torch.from_numpy(inst).to(DEV), flag=torch.from_numpy(y > 3).to(DEV), tag="kept")
    if batch is not None:
        data.batch = torch.from_numpy(batch).to(DEV)
    out = GridSampling3D(size, quantize_coords=True)(data)
    assert np.array_equal(out.pos.cpu().numpy(), ref["pos"])  # fp32 bit-exact: same order, same division
    assert np.array_equal(out.x.cpu().numpy(), ref["x"])
    assert np.array_equal(out.y.cpu().numpy(), ref["y"])
    assert np.array_equal(out.instance_labels.cpu().numpy(), ref_inst)
    assert np.array_equal(out.coords.cpu().numpy(), ref["coords"]) and out.coords.dtype == torch.int32
    if batch is not None:
        assert np.array_equal(out.batch.cpu().numpy(), ref["batch"])
    assert out.flag.dtype == torch.bool and out.flag.shape[0] == out.pos.shape[0]
    assert out.tag == "kept" and float(out.grid_size[0]) == pytest.approx(size)


def test_reference_test_properties():
    # test/test_grid_sampling.py:29-67 on the device
    from torch_points3d_amd.grid_sampling import GridSampling3D
    from torch_points3d_amd.kpconv_blocks import PDData
    pos = torch.tensor([[0, 0, 0.01], [0.01, 0, 0], [0, 0.01, 0], [0, 0.01, 0], [0.01, 0, 0.01]])
    y = torch.tensor([1, 0, 1, 1, 0])
    out = GridSampling3D(0.04)(PDData(pos=pos.to(DEV), batch=torch.zeros(5, dtype=torch.long, device=DEV), y=y.to(DEV)))
    assert out.y.tolist() == [1] and out.pos.shape == (1, 3)

    torch.manual_seed(0)
    data = PDData(pos=(torch.randn(1000, 3) * 0.1).to(DEV), x=torch.ones(1000, 1, device=DEV))
    gr, sparse = GridSampling3D(0.02), GridSampling3D(0.02, quantize_coords=True)
    shapes = []
    u = data.clone()
    for _ in range(2):
        u = gr(u)
        shapes.append(u.pos.shape[0])
    q = sparse(u)
    shapes.append(np.unique(q.pos.cpu().numpy(), axis=0).shape[0])
    assert shapes == [shapes[0]] * 3
    assert q.coords.dtype == torch.int32 and q.coords.shape[0] == q.x.shape[0]


def test_last_mode_keeps_one_input_point_per_voxel():
    from torch_points3d_amd.grid_sampling import GridSampling3D
    from torch_points3d_amd.kpconv_blocks import PDData
    pos, _ = cloud(5000, 1, 3)
    tpos = torch.from_numpy(pos).to(DEV)
    out = GridSampling3D(0.1, mode="last")(PDData(pos=tpos.clone(), x=tpos.clone()))
    ref = voxel_ref.grid_sampling_mean(pos, 0.1)
    assert out.pos.shape[0] == ref["pos"].shape[0]
    assert torch.equal(out.pos, out.x)  # every key is indexed with the same representative
    got = {tuple(r) for r in out.pos.cpu().numpy().tolist()}
    assert got <= {tuple(r) for r in pos.tolist()}
    assert np.unique(voxel_ref.voxel_coords(out.pos.cpu().numpy(), 0.1), axis=0).shape[0] == out.pos.shape[0]


def test_bad_arguments():
    from torch_points3d_amd import grid_sampling as gs
    with pytest.raises(RuntimeError):
        gs.voxel_cluster(torch.rand(10, 3), None, 0.1)  # CPU tensor: no fallback
    with pytest.raises(ValueError):
        gs.voxel_cluster(torch.rand(10, 2, device=DEV), None, 0.1)
    with pytest.raises(RuntimeError):
        gs.voxel_cluster(torch.rand(10, 3, device=DEV) * 1e9, None, 1e-3)  # coordinates beyond 2^24 voxels
    c, u, o, s = gs.voxel_cluster(torch.empty(0, 3, device=DEV), None, 0.1)
    assert c.numel() == 0 and u.numel() == 0 and s.tolist() == [0]


def test_cluster_counts_per_cloud():
    from torch_points3d_amd import grid_sampling as gs
    pos, batch = cloud(30000, 6, 17)
    batch[batch == 2] = 3  # a cloud id without points
    ref_cluster, ref_perm = voxel_ref.consecutive_cluster(voxel_ref.grid_cluster_key(voxel_ref.voxel_coords(pos, 0.07), batch))
    out = gs.voxel_cluster(torch.from_numpy(pos).to(DEV), torch.from_numpy(batch).to(DEV), 0.07, return_counts=True)
    counts = out[4]
    assert counts.tolist() == np.bincount(batch[ref_perm], minlength=6).tolist() and counts[2] == 0
    c1 = gs.voxel_cluster(torch.from_numpy(pos).to(DEV), None, 0.07, return_counts=True)[4]
    assert c1.tolist() == [len(np.unique(voxel_ref.grid_cluster_key(voxel_ref.voxel_coords(pos, 0.07), None)))]


def test_matches_reference_transform_golden():
    """tests/golden/grid_sampling.npz was produced by the reference's own GridSampling3D / group_data code
    (core/data_transform/grid_transform.py, loaded by file path in tests/golden/make_golden.py with its third-party
    calls bound to oracle/voxel_ref.py): every grouped attribute must come out identically."""
    from conftest import load_golden
    from torch_points3d_amd.grid_sampling import GridSampling3D
    from torch_points3d_amd.kpconv_blocks import PDData
    g = load_golden("grid_sampling")
    size = float(g["size"][0])
    dev = lambda k: g[k].to(DEV)  # noqa: E731
    data = PDData(pos=dev("pos"), batch=dev("batch"), x=dev("x"), y=dev("y"), instance_labels=dev("instance_labels"),
                  flag=dev("flag"), origin_id=torch.arange(g["pos"].shape[0], device=DEV),
                  scalar=torch.tensor([7.0], device=DEV))
    out = GridSampling3D(size, quantize_coords=True, mode="mean")(data)
    for k in ("pos", "batch", "x", "y", "instance_labels", "flag", "origin_id", "coords", "scalar"):
        got, ref = getattr(out, k).cpu(), g["out." + k]
        assert got.shape == ref.shape, k
        assert torch.equal(got.to(ref.dtype), ref), k
    assert out.coords.dtype == torch.int32 and out.flag.dtype == torch.bool
    assert float(out.grid_size[0]) == pytest.approx(float(g["out.grid_size"][0]))
    nb = GridSampling3D(0.1, mode="mean")(PDData(pos=dev("pos"), x=dev("x")))
    assert torch.equal(nb.pos.cpu(), g["nobatch.pos"]) and torch.equal(nb.x.cpu(), g["nobatch.x"])


def test_coordinate_bounds_hint_is_verified():
    """An enclosing extent handed in by the caller replaces the device reduction's host read; a wrong one is detected
    at the final read and the clustering is redone with the measured extent."""
    from torch_points3d_amd import grid_sampling as gs
    from torch_points3d_amd.grid_sampling import GridSampling3D
    from torch_points3d_amd.kpconv_blocks import PDData
    pos, batch = cloud(20000, 3, 5)
    tp_, tb = torch.from_numpy(pos).to(DEV), torch.from_numpy(batch).to(DEV)
    exact = gs.voxel_cluster(tp_, tb, 0.05, return_counts=True)
    loose = gs.voxel_cluster(tp_, tb, 0.05, return_counts=True, coord_bounds=[-7, -3, -9, 40, 33, 29, 2])
    wrong = gs.voxel_cluster(tp_, tb, 0.05, return_counts=True, coord_bounds=[0, 0, 0, 5, 5, 5, 0])
    for other in (loose, wrong):
        for a, b in zip(exact[:5], other[:5]):
            assert torch.equal(a.cpu(), b.cpu())
    assert exact[5].tolist() == wrong[5].tolist() and loose[5][:7].tolist() == [-7, -3, -9, 40, 33, 29, 2]
    # two chained samplers: the second one runs on the bounds the first one attached
    first = GridSampling3D(0.05)(PDData(pos=tp_, batch=tb, x=tp_.clone()))
    assert first.pos_bounds is not None
    second = GridSampling3D(0.1)(first.clone())
    ref = voxel_ref.grid_sampling_mean(first.pos.cpu().numpy(), 0.1, batch=first.batch.cpu().numpy(), x=first.x.cpu().numpy())
    assert np.array_equal(second.pos.cpu().numpy(), ref["pos"]) and np.array_equal(second.batch.cpu().numpy(), ref["batch"])
    # positions moved after sampling (stale bounds): still correct
    moved = first.clone()
    moved.pos = moved.pos + 3.0
    third = GridSampling3D(0.1)(moved)
    ref = voxel_ref.grid_sampling_mean((first.pos + 3.0).cpu().numpy(), 0.1, batch=first.batch.cpu().numpy())
    assert np.array_equal(third.pos.cpu().numpy(), ref["pos"])
