"""Times the shipped split-role entry points (forward with prologue + statistics + side output; input gradient with dY side
output) on the big layers of the BASELINE step, back to back -- for A/B builds of csrc/gemm_rows_sp.hip."""
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


h = _lib.load()
tot_f = tot_b = 0.0
for M, N, K in [(524288, 128, 128), (1048576, 128, 64), (1048576, 64, 64), (262144, 256, 128), (262144, 128, 128)]:
    Y = torch.randn(M, K, device=DEV)
    dA = torch.randn(M, K, device=DEV)
    Bt = torch.randn(N, K, device=DEV) * 0.1
    v = [torch.rand(K, device=DEV) * 0.5 + 0.5 for _ in range(5)]
    C = torch.empty(M, N, device=DEV)
    side = torch.empty(M, K, device=DEV)
    chunks = h.tp3d_gemm_rows_sp_chunks(M, N, K, 1)
    part = torch.empty(chunks * 4 * N, device=DEV)
    st = _lib.stream_ptr(Y.device)

    def fwd():
        _lib.call("tp3d_gemm_rows_bnact_sp_f32", _lib.ptr(Y), _lib.ptr(v[0]), _lib.ptr(v[1]), _lib.ptr(v[2]), 0.01, _lib.ptr(Bt), M, N, K,
                  _lib.ptr(C), _lib.ptr(part), _lib.ptr(side), 0, st)

    def bwd():
        _lib.call("tp3d_gemm_rows_bnbwd_sp_f32", _lib.ptr(Y), _lib.ptr(dA), _lib.ptr(v[0]), _lib.ptr(v[1]), _lib.ptr(v[2]), _lib.ptr(v[3]),
                  _lib.ptr(v[4]), 0.01, _lib.ptr(Bt), M, N, K, _lib.ptr(C), N, _lib.ptr(side), None, 1, 0, st)
    tf, tb = timeit(fwd), timeit(bwd)
    tot_f += tf
    tot_b += tb
    print("M=%8d N=%3d K=%3d   forward %7.1f us %6.1f TF   input gradient %7.1f us %6.1f TF" % (
        M, N, K, tf, 2.0 * M * N * K / tf / 1e6, tb, 2.0 * M * N * K / tb / 1e6), flush=True)
print("sum: forward %.1f us, input gradient %.1f us" % (tot_f, tot_b))
