"""torch_points3d_amd.kpconv_unet.unet_config against the reference's own architecture file
(applications/conf/kpconv/unet_4.yaml resolved by tests/golden/make_golden.py::make_unet4_config_case the way
utils/model_building_utils/model_definition_resolver.py resolves it).  Pure host logic: runs without a GPU."""
import json
import os

import pytest

from conftest import GOLDEN


@pytest.mark.parametrize("key,feat,in_feat,grid", [("feat3_infeat64_grid0.02", 3, 64, 0.02),
                                                   ("feat1_infeat32_grid0.05", 1, 32, 0.05)])
def test_unet_config_reproduces_reference_yaml(key, feat, in_feat, grid):
    from torch_points3d_amd.kpconv_unet import unet_config
    ref = json.load(open(os.path.join(GOLDEN, "kpconv_unet4_config.json")))[key]
    cfg = unet_config(num_layers=4, input_nc=feat, in_feat=in_feat, in_grid_size=grid, bn_momentum=0.2, max_neighbors=25)
    down, up = ref["down_conv"], ref["up_conv"]
    assert len(cfg["down_conv"]) == len(down["down_conv_nn"]) == 5 and len(cfg["up_conv"]) == len(up["up_conv_nn"]) == 4
    for i, level in enumerate(cfg["down_conv"]):
        for name in ("down_conv_nn", "block_names", "has_bottleneck", "max_num_neighbors"):
            assert level[name] == down[name][i], (i, name)
        for name in ("grid_size", "prev_grid_size"):
            assert level[name] == pytest.approx(down[name][i], rel=1e-12), (i, name)
        # the strided flag is an exact float comparison in the reference (blocks.py:58): keep it exact here too
        assert [a != b for a, b in zip(level["prev_grid_size"], level["grid_size"])] == \
               [a != b for a, b in zip(down["prev_grid_size"][i], down["grid_size"][i])]
        assert down["deformable"][i] == [False, False]
    assert down["module_name"] == "KPDualBlock" and up["module_name"] == "FPModule_PD" and up["skip"] is True
    for i, stage in enumerate(cfg["up_conv"]):
        assert stage["up_conv_nn"] == up["up_conv_nn"][i] and stage["up_k"] == up["up_k"][i]
        assert stage["bn_momentum"] == up["bn_momentum"][i]


def test_unet_builds_with_reference_names_on_cpu():
    import torch
    from torch_points3d_amd.kpconv_unet import KPConv
    model = KPConv("unet", input_nc=3, in_feat=8, in_grid_size=0.02, num_layers=4, output_nc=5)
    assert len(model.down_modules) == 5 and len(model.inner_modules) == 1 and len(model.up_modules) == 4
    assert model.has_mlp_head and model.output_nc == 5  # test/test_api.py:57-71
    assert KPConv("unet", input_nc=3, in_feat=8, num_layers=4).output_nc == 8  # :29-42
    with pytest.raises(RuntimeError):  # no CPU fallback: the radius search refuses CPU tensors
        from torch_points3d_amd.kpconv_blocks import PDData
        model(PDData(pos=torch.rand(50, 3), x=torch.rand(50, 4), batch=torch.zeros(50, dtype=torch.long)))
