"""Host-side logic of the partial-dense mirrors that needs no device kernel (runs on CPU tensors): the reference's
branches for a globally pooled innermost level and for precomputed interpolation tables
(core/base_conv/partial_dense.py:119-146, core/spatial_ops/interpolate.py:34-50), `group_data(mode="last")`
(core/data_transform/grid_transform.py:66-68), the segment table of a sorted batch vector, PDData."""
import pytest
import torch

from torch_points3d_amd.kpconv_blocks import PDData


def test_fp_module_innermost_branch_and_precomputed_tables():
    from torch_points3d_amd.partial_dense import FPModule_PD
    torch.manual_seed(0)
    N, clouds = 40, 3
    batch = torch.sort(torch.randint(0, clouds, (N,)))[0]
    batch[:clouds] = torch.arange(clouds)  # every cloud present
    batch = torch.sort(batch)[0]
    skip = PDData(pos=torch.rand(N, 3), x=torch.randn(N, 4), batch=batch)
    pooled = PDData(pos=torch.zeros(clouds, 3), x=torch.randn(clouds, 6), batch=torch.arange(clouds))
    fp = FPModule_PD(1, [6 + 4, 5], bn_momentum=0.1).eval()
    out = fp((pooled, skip))
    want = fp.nn(torch.cat([pooled.x[batch], skip.x], dim=1))  # one row per cloud: broadcast by batch id
    assert torch.equal(out.x, want) and out.pos is skip.pos and out.x.shape == (N, 5)

    # precomputed interpolation table (what KNNInterpolate.precompute produces), two neighbours per point
    M = 7
    coarse = PDData(pos=torch.rand(M, 3), x=torch.randn(M, 6), batch=torch.zeros(M, dtype=torch.long))
    x_idx = torch.randint(0, M, (2 * N,))
    y_idx = torch.arange(N).repeat_interleave(2)
    w = torch.rand(2 * N, 1) + 0.1
    pre = PDData(num_nodes=N, x_idx=x_idx, y_idx=y_idx, weights=w,
                 normalisation=torch.zeros(N, 1).index_add_(0, y_idx, w))
    one = PDData(pos=skip.pos, x=skip.x, batch=torch.zeros(N, dtype=torch.long))
    out = fp((coarse, one), precomputed=[pre])
    blend = torch.zeros(N, 6).index_add_(0, y_idx, coarse.x[x_idx] * w) / pre.normalisation
    torch.testing.assert_close(out.x, fp.nn(torch.cat([blend, one.x], 1)))
    assert out.up_idx == 1
    with pytest.raises(ValueError):
        fp.upsample_op(coarse, PDData(pos=torch.rand(N + 1, 3)), precomputed=pre)


def test_group_data_last_mode_and_pddata():
    from torch_points3d_amd.grid_sampling import group_data
    data = PDData(pos=torch.arange(12.).reshape(4, 3), x=torch.arange(8.).reshape(4, 2), y=torch.tensor([3, 1, 2, 0]),
                  batch=torch.zeros(4, dtype=torch.long), scalar=torch.tensor([1.0]), name="cloud", nothing=None)
    keep = torch.tensor([3, 1])
    out = group_data(data.clone(), unique_pos_indices=keep, mode="last")
    assert torch.equal(out.pos, data.pos[keep]) and torch.equal(out.y, data.y[keep]) and out.name == "cloud"
    assert torch.equal(out.scalar, data.scalar) and "nothing" not in out.keys
    with pytest.raises(ValueError):
        group_data(data.clone(), mode="last")
    with pytest.raises(ValueError):
        group_data(data.clone(), mode="mean")
    bad = data.clone()
    bad.edge_index = torch.zeros(2, 3)
    with pytest.raises(ValueError):
        group_data(bad, unique_pos_indices=keep, mode="last")
    shallow = data.shallow_copy()
    assert shallow.pos is data.pos and shallow is not data
    deep = data.clone()
    assert deep.pos is not data.pos and torch.equal(deep.pos, data.pos)


def test_segments_of_a_sorted_batch_vector():
    from torch_points3d_amd import torchpoints as tp
    bx = torch.tensor([0, 0, 0, 2, 2, 3])
    seg, nclouds, nmax = tp._segments(bx)
    assert seg.tolist() == [0, 3, 3, 5, 6] and nclouds == 4 and nmax == 3
    assert tp._segments(bx)[0] is seg  # cached for the same tensor object
    with pytest.raises(ValueError):
        tp._segments(torch.tensor([1, 0, 2]))
    primed = torch.tensor([0, 1, 1])
    tp.prime_segments(primed, torch.tensor([0, 1, 3]), 2, 2)
    assert tp._segments(primed)[1:] == (2, 2)
