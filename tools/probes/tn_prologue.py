"""Weight-gradient contraction dW = dY^T A' with A' = leaky(BatchNorm(Yp)) either read from memory (the forward kernel's side
output) or formed on the fly from Yp by the A-operand prologue of tp3d_gemm_tn_bn_f32: what does the prologue cost?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_points3d_amd import _lib, fused  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=20):
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


for M, N, K in [(524288, 128, 128), (1048576, 128, 64), (1048576, 64, 64), (262144, 256, 128), (262144, 128, 128)]:
    dY = torch.randn(M, N, device=DEV)
    Yp = torch.randn(M, K, device=DEV)
    mean, scale, beta = torch.randn(K, device=DEV) * 0.1, torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
    act = torch.empty_like(Yp)
    st = _lib.stream_ptr(dY.device)
    _lib.call("tp3d_bn_act_f32", _lib.ptr(Yp), _lib.ptr(mean), _lib.ptr(scale), _lib.ptr(beta), 0.01, M, K, _lib.ptr(act), st)
    ref = fused.gemm_tn(dY, act)
    out = torch.empty(N, K, device=DEV)
    ws = _lib.gemm_tn_workspace(M, N, K, dY.device)

    def fusedtn():
        _lib.call("tp3d_gemm_tn_bn_f32", _lib.ptr(dY), None, None, 1, None, None, None, None, None, 0.01, _lib.ptr(Yp), _lib.ptr(mean),
                  _lib.ptr(scale), _lib.ptr(beta), 0.01, M, N, K, _lib.ptr(out), _lib.ptr(ws), st)
    fusedtn()
    err = float((out - ref).abs().max() / ref.abs().max())
    t0, t1 = timeit(lambda: fused.gemm_tn(dY, act)), timeit(fusedtn)
    print("M=%8d N=%3d K=%3d  plain %7.1f us   A' formed on the fly %7.1f us   (rel diff %.1e)" % (M, N, K, t0, t1, err), flush=True)
