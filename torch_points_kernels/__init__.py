"""Drop-in import name for the reference: `import torch_points_kernels as tp`
(reference torch_points3d/core/spatial_ops/sampling.py:7, core/base_conv/dense.py:19, ...).

Every function is served by torch_points3d_amd's HIP kernels on MI355X; nothing is computed on the CPU.
"""
from torch_points3d_amd.torchpoints import (  # noqa: F401
    ball_query,
    furthest_point_sample,
    grouping_operation,
    three_interpolate,
    three_nn,
)

__all__ = ["furthest_point_sample", "ball_query", "three_nn", "three_interpolate", "grouping_operation"]
