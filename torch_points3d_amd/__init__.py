"""torch_points3d_amd -- MI355X-native implementation of torch-points3d's data-parallel hot path.

`torch_points3d_amd.torchpoints` serves the `torch_points_kernels` function API from hand-written HIP
kernels (libtp3d_hip.so, C-ABI in include/tp3d_hip.h); `torch_points3d_amd.dense` / `.pointnet2` are the
host-side mirror of the reference's dense PointNet++ modules that call it.
"""
from .torchpoints import (  # noqa: F401
    ball_query,
    furthest_point_sample,
    grouping_operation,
    three_interpolate,
    three_nn,
)

__version__ = "0.1.0"
