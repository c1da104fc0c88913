"""The reduction pass of the BatchNorm + activation backward (tp3d_bn_bwd_reduce_f32: two read streams, dbeta / dgamma /
c1 / c2 out) on the shapes of the BASELINE step: time and algorithmic read rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_points3d_amd import _lib  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for M, C in [(524288, 128), (1048576, 64), (262144, 128), (262144, 256), (1048576, 128), (2097152, 96)]:
    Y = torch.randn(M, C, device=DEV)
    dA = torch.randn(M, C, device=DEV)
    v = [torch.rand(C, device=DEV) + 0.5 for _ in range(4)]
    red = torch.empty(4, C, device=DEV)
    ws = _lib.bn_workspace(M, C, Y.device)
    st = _lib.stream_ptr(Y.device)

    def run():
        _lib.call("tp3d_bn_bwd_reduce_f32", _lib.ptr(dA), None, _lib.ptr(Y), _lib.ptr(v[0]), _lib.ptr(v[1]), _lib.ptr(v[2]), _lib.ptr(v[3]),
                  0.01, M, 1, C, 1, _lib.ptr(red[0]), _lib.ptr(red[1]), _lib.ptr(red[2]), _lib.ptr(red[3]), _lib.ptr(ws), 0, st)
    t = timeit(run)
    print("M=%8d C=%4d  %7.1f us  %5.2f TB/s   checksum %.6e %.6e" % (M, C, t, 8.0 * M * C / t / 1e6, float(red[0].sum()), float(red[1].sum())),
          flush=True)
