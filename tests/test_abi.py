"""The C-ABI shared library must load and export every symbol include/tp3d_hip.h declares; the Python
boundary must refuse CPU tensors instead of falling back.  No compute is launched (no GPU needed)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tp3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tp3d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from torch_points3d_amd import _lib, build
    build.build_library()
    h = ctypes.CDLL(build.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 11
    for n in names:
        assert hasattr(h, n), "libtp3d_hip.so does not export " + n
    # the binding table and the header agree
    assert set(_lib.SIGNATURES) | set(_lib.MISC) == set(names)
    assert _lib.load().tp3d_abi_version() == _lib.ABI_VERSION
    assert _lib.load().tp3d_strerror(-1).decode().startswith("bad argument")


def test_cpu_library_exports_every_declared_symbol():
    """include/tp3d_cpu.h (torch_points_kernels.points_cpu): plain C, loadable without any GPU runtime"""
    from torch_points3d_amd import build
    text = open(os.path.join(ROOT, "include", "tp3d_cpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(tp3d_cpu_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 6
    h = ctypes.CDLL(build.build_cpu_library())
    for n in names:
        assert hasattr(h, n), "libtp3d_cpu.so does not export " + n
    import torch_points_kernels.points_cpu as pc
    assert callable(pc.ball_query) and callable(pc.dense_knn)


def test_library_contains_gfx950_code_object():
    from torch_points3d_amd import build
    blob = open(build.build_library(), "rb").read()
    assert b"gfx950" in blob


def test_drop_in_module_exposes_reference_api():
    import torch_points_kernels as tp
    for n in ("furthest_point_sample", "ball_query", "three_nn", "three_interpolate", "grouping_operation",
              "region_grow", "instance_iou"):  # every name the reference imports from the package
        assert callable(getattr(tp, n))
    import torch_points_kernels.points_cpu  # noqa: F401  (core/data_transform/transforms.py:16)


def test_cpu_tensors_are_refused_not_emulated():
    import torch_points_kernels as tp
    pos = torch.rand(1, 16, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tp.furthest_point_sample(pos, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tp.ball_query(0.2, 4, pos, pos)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tp.three_nn(pos, pos)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tp.three_interpolate(torch.rand(1, 2, 16), torch.zeros(1, 16, 3, dtype=torch.long), torch.rand(1, 16, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tp.grouping_operation(torch.rand(1, 2, 16), torch.zeros(1, 4, 2, dtype=torch.long))


def test_argument_errors_match_reference_conventions():
    import torch_points_kernels as tp
    pos = torch.rand(1, 16, 3)
    with pytest.raises(ValueError):
        tp.furthest_point_sample(pos, 17)  # npoint > N
    with pytest.raises(ValueError):
        tp.three_nn(pos, pos[:, :2])  # fewer than 3 known points
    with pytest.raises(Exception):
        tp.ball_query(0.2, 4, pos[0], pos[0], mode="partial_dense")  # batch vectors missing
    with pytest.raises(Exception):
        tp.ball_query(0.2, 4, pos, pos, mode="dense", batch_x=torch.zeros(16), batch_y=torch.zeros(16))
    with pytest.raises(Exception):
        tp.ball_query(0.2, 4, pos, pos, mode="nope")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "torch_points3d_amd")
    for base in (pkg, os.path.join(ROOT, "torch_points_kernels"), os.path.join(ROOT, "tools")):
        for dirpath, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert "import oracle" not in src and "from oracle" not in src and "tpk_ref" not in src.replace(
                        "oracle/tpk_ref_cpu.c", "").replace("tpk_ref_", ""), f
