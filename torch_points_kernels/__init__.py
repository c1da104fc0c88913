"""Drop-in import name for the reference: `import torch_points_kernels as tp`
(reference torch_points3d/core/spatial_ops/sampling.py:7, core/base_conv/dense.py:19, ...).

The five point-set operators are served by torch_points3d_amd's HIP kernels on MI355X and never computed on the CPU;
`points_cpu` (host-side searches), `region_grow` and `instance_iou` (PointGroup's clustering and its IoU matrix) complete
the names the reference imports from this package.
"""
from torch_points3d_amd.torchpoints import (  # noqa: F401
    ball_query,
    furthest_point_sample,
    grouping_operation,
    three_interpolate,
    three_nn,
)

from .cluster import grow_proximity, region_grow  # noqa: F401,E402
from .metrics import instance_iou  # noqa: F401,E402

__all__ = ["furthest_point_sample", "ball_query", "three_nn", "three_interpolate", "grouping_operation", "region_grow",
           "grow_proximity", "instance_iou"]
