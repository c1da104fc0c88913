"""CPU ORACLE binding -- test infrastructure, NOT product code.

Presents the ``torch_points_kernels`` function API (reference import sites:
core/spatial_ops/sampling.py:7, core/spatial_ops/neighbour_finder.py:5,
core/base_conv/dense.py:19, modules/pointnet2/dense.py:4) on CPU tensors, backed by
``oracle/libtpk_ref_cpu.so`` (oracle/tpk_ref_cpu.c).  Only tests/, ``__graft_entry__.smoke()``
and bench.py's ``cpu_baseline`` leg may import this module; the product package
``torch_points3d_amd`` never does.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtpk_ref_cpu.so")

_c_f = ctypes.POINTER(ctypes.c_float)
_c_l = ctypes.POINTER(ctypes.c_int64)
_int = ctypes.c_int
_i64 = ctypes.c_int64


def build(force=False):
    """Compile the C restatement (gcc, seconds)."""
    src = os.path.join(_HERE, "tpk_ref_cpu.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.tpk_ref_fps_f32.argtypes = [_c_f, _int, _int, _int, _c_f, _c_l]
        L.tpk_ref_ball_query_dense_f32.argtypes = [_c_f, _c_f, _int, _int, _int, ctypes.c_float, _int, _int, _c_l, _c_f]
        L.tpk_ref_ball_query_partial_dense_f32.argtypes = [
            _c_f, _c_f, _c_l, _c_l, _i64, _i64, ctypes.c_float, _int, _int, _c_l, _c_f]
        L.tpk_ref_knn_partial_dense_f32.argtypes = [_c_f, _c_f, _c_l, _c_l, _i64, _i64, _int, _c_l, _c_f]
        L.tpk_ref_knn_partial_dense_f32.restype = _int
        L.tpk_ref_three_nn_f32.argtypes = [_c_f, _c_f, _int, _int, _int, _c_f, _c_l]
        L.tpk_ref_three_interpolate_fwd_f32.argtypes = [_c_f, _c_l, _c_f, _int, _int, _int, _int, _c_f]
        L.tpk_ref_three_interpolate_bwd_f32.argtypes = [_c_f, _c_l, _c_f, _int, _int, _int, _int, _c_f]
        L.tpk_ref_group_fwd_f32.argtypes = [_c_f, _c_l, _int, _int, _int, _int, _int, _c_f]
        L.tpk_ref_group_bwd_f32.argtypes = [_c_f, _c_l, _int, _int, _int, _int, _int, _c_f]
        L.tpk_ref_set_num_threads.argtypes = [_int]
        L.tpk_ref_num_threads.restype = _int
        for name in ("tpk_ref_fps_f32", "tpk_ref_ball_query_dense_f32", "tpk_ref_ball_query_partial_dense_f32",
                     "tpk_ref_three_nn_f32", "tpk_ref_three_interpolate_fwd_f32",
                     "tpk_ref_three_interpolate_bwd_f32", "tpk_ref_group_fwd_f32", "tpk_ref_group_bwd_f32"):
            getattr(L, name).restype = _int
        _lib = L
    return _lib


def set_num_threads(n):
    lib().tpk_ref_set_num_threads(int(n))


def num_threads():
    return lib().tpk_ref_num_threads()


def _f(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.cast(t.data_ptr(), _c_f)


def _l(t):
    assert t.dtype == torch.int64 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.cast(t.data_ptr(), _c_l)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("oracle %s failed with code %d" % (what, rc))


def _prep(t):
    return t.detach().to(torch.float32).contiguous()


def furthest_point_sample(xyz, npoint):
    if npoint > xyz.shape[1]:
        raise ValueError("caanot sample %i points from an input set of %i points" % (npoint, xyz.shape[1]))
    xyz = _prep(xyz)
    B, N, _ = xyz.shape
    out = torch.empty(B, npoint, dtype=torch.int64)
    scratch = torch.empty(B, N, dtype=torch.float32)
    _check(lib().tpk_ref_fps_f32(_f(xyz), B, N, npoint, _f(scratch), _l(out)), "fps")
    return out


def ball_query(radius, nsample, x, y, mode="dense", batch_x=None, batch_y=None, sort=False):
    if mode is None:
        raise Exception('The mode should be defined within ["partial_dense | dense"]')
    if mode.lower() == "partial_dense":
        if batch_x is None or batch_y is None:
            raise Exception("batch_x and batch_y should be provided")
        assert x.dim() == 2 and y.dim() == 2
        x, y = _prep(x), _prep(y)
        bx, by = batch_x.to(torch.int64).contiguous(), batch_y.to(torch.int64).contiguous()
        idx = torch.empty(y.shape[0], nsample, dtype=torch.int64)
        d2 = torch.empty(y.shape[0], nsample, dtype=torch.float32)
        _check(lib().tpk_ref_ball_query_partial_dense_f32(
            _f(x), _f(y), _l(bx), _l(by), x.shape[0], y.shape[0], float(radius), nsample, int(sort), _l(idx), _f(d2)),
            "ball_query_partial_dense")
        return idx, d2
    elif mode.lower() == "dense":
        if batch_x is not None or batch_y is not None:
            raise Exception("batch_x and batch_y should not be provided")
        assert x.dim() == 3 and y.dim() == 3
        x, y = _prep(x), _prep(y)
        B, N, _ = x.shape
        np_ = y.shape[1]
        idx = torch.empty(B, np_, nsample, dtype=torch.int64)
        d2 = torch.empty(B, np_, nsample, dtype=torch.float32)
        _check(lib().tpk_ref_ball_query_dense_f32(
            _f(x), _f(y), B, N, np_, float(radius), nsample, int(sort), _l(idx), _f(d2)), "ball_query_dense")
        return idx, d2
    raise Exception("unrecognized mode {}".format(mode))


def knn(k, x, y, batch_x=None, batch_y=None):
    """k nearest support rows x (M,3) of every query y (Nq,3) inside its own cloud -> (idx (Nq,k), dist2 (Nq,k))"""
    x, y = _prep(x), _prep(y)
    bx = torch.zeros(x.shape[0], dtype=torch.int64) if batch_x is None else batch_x.to(torch.int64).contiguous()
    by = torch.zeros(y.shape[0], dtype=torch.int64) if batch_y is None else batch_y.to(torch.int64).contiguous()
    idx = torch.empty(y.shape[0], k, dtype=torch.int64)
    d2 = torch.empty(y.shape[0], k, dtype=torch.float32)
    _check(lib().tpk_ref_knn_partial_dense_f32(_f(x), _f(y), _l(bx), _l(by), x.shape[0], y.shape[0], int(k), _l(idx),
                                               _f(d2)), "knn_partial_dense")
    return idx, d2


def three_nn(unknown, known):
    if known.shape[1] < 3:
        raise ValueError("Not enough points. unknown should ahve at least 3 points.")
    unknown, known = _prep(unknown), _prep(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist = torch.empty(B, n, 3, dtype=torch.float32)
    idx = torch.empty(B, n, 3, dtype=torch.int64)
    _check(lib().tpk_ref_three_nn_f32(_f(unknown), _f(known), B, n, m, _f(dist), _l(idx)), "three_nn")
    return dist, idx


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        features = features.contiguous()
        idx = idx.to(torch.int64).contiguous()
        weight = weight.contiguous()
        B, C, m = features.shape
        n = idx.shape[1]
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        out = torch.empty(B, C, n, dtype=torch.float32)
        _check(lib().tpk_ref_three_interpolate_fwd_f32(_f(features), _l(idx), _f(weight), B, C, m, n, _f(out)),
               "three_interpolate_fwd")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, n = grad_out.shape
        g = torch.empty(B, C, ctx.m, dtype=torch.float32)
        _check(lib().tpk_ref_three_interpolate_bwd_f32(_f(grad_out), _l(idx), _f(weight), B, C, ctx.m, n, _f(g)),
               "three_interpolate_bwd")
        return g, None, None


def three_interpolate(features, idx, weight):
    return _ThreeInterpolate.apply(features, idx, weight)


class _Grouping(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        features = features.contiguous()
        idx = idx.to(torch.int64).contiguous()
        B, C, N = features.shape
        _, np_, ns = idx.shape
        ctx.save_for_backward(idx)
        ctx.N = N
        out = torch.empty(B, C, np_, ns, dtype=torch.float32)
        _check(lib().tpk_ref_group_fwd_f32(_f(features), _l(idx), B, C, N, np_, ns, _f(out)), "group_fwd")
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, np_, ns = grad_out.shape
        g = torch.empty(B, C, ctx.N, dtype=torch.float32)
        _check(lib().tpk_ref_group_bwd_f32(_f(grad_out), _l(idx), B, C, ctx.N, np_, ns, _f(g)), "group_bwd")
        return g, None


def grouping_operation(features, idx):
    return _Grouping.apply(features, idx)
