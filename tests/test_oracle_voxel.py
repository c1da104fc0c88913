"""The numpy restatement of GridSampling3D (oracle/voxel_ref.py) against the properties the reference's own test
holds for the transform (test/test_grid_sampling.py:17-67).  The reference's fragment_000003.pt fixture is a pickled
torch_geometric Data object and is not loaded (pickle loaders are not used on reference files); the random-cloud half
of each test is reproduced."""
import numpy as np

from oracle import voxel_ref


def test_majority_label_of_one_voxel():
    # test_grid_sampling.py:17-35: five points in one 0.04 voxel, the output label is the majority vote
    pos = np.array([[0, 0, 0.01], [0.01, 0, 0], [0, 0.01, 0], [0, 0.01, 0], [0.01, 0, 0.01]], np.float32)
    for seed in range(8):
        y = np.random.RandomState(seed).randint(0, 2, 5)
        uniq, counts = np.unique(y, return_counts=True)
        out = voxel_ref.grid_sampling_mean(pos, 0.04, batch=np.zeros(5, np.int64), y=y)
        assert out["pos"].shape == (1, 3)
        assert out["y"].tolist() == [uniq[np.argmax(counts)]]
        assert out["unique_pos_indices"].tolist() == [4]  # last point of the voxel represents it


def test_double_sampling_is_idempotent():
    # test_grid_sampling.py:37-58 ("random" leg): sampling the sampled cloud again keeps every point
    rs = np.random.RandomState(0)
    pos = (rs.randn(1000, 3) * 0.1).astype(np.float32)
    first = voxel_ref.grid_sampling_mean(pos, 0.02, x=np.ones((1000, 1), np.float32))
    second = voxel_ref.grid_sampling_mean(first["pos"], 0.02)
    assert second["pos"].shape[0] == first["pos"].shape[0]
    assert np.unique(second["coords"], axis=0).shape[0] == first["pos"].shape[0]
    np.testing.assert_array_equal(first["x"], 1.0)


def test_quantized_coords():
    # test_grid_sampling.py:60-67
    rs = np.random.RandomState(1)
    pos = (rs.randn(100, 3) * 0.1).astype(np.float32)
    out = voxel_ref.grid_sampling_mean(pos, 0.2, x=np.ones((100, 1), np.float32))
    assert out["coords"].dtype == np.int32 and out["coords"].shape[0] == out["x"].shape[0] == out["pos"].shape[0]


def test_cluster_ids_follow_batch_z_y_x_order_and_half_to_even():
    pos = np.array([[0.5, 0, 0], [1.5, 0, 0], [2.5, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 0]], np.float32)
    coords = voxel_ref.voxel_coords(pos, 1.0)
    assert coords[:3, 0].tolist() == [0.0, 2.0, 2.0]  # round half to even
    batch = np.array([1, 0, 0, 0, 0, 1])
    cluster, perm = voxel_ref.consecutive_cluster(voxel_ref.grid_cluster_key(coords, batch))
    # cloud 0: (x=0,y=1,z=0) < (x=2,y=0,z=0)? no: z slowest among xyz, then y, then x
    # voxels of cloud 0: (2,0,0) {1,2}, (0,1,0) {3}, (0,0,1) {4}; cloud 1: (0,0,0) {0,5}
    assert cluster.tolist() == [3, 0, 0, 1, 2, 3]
    assert perm.tolist() == [2, 3, 4, 5]
