"""Where the time of gemm_tn_x3 goes: the same launch with 9, 6 and 1 term pairs (1 = the hi*hi pair only: loaders and
memory system unchanged, a ninth of the MFMAs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from torch_points3d_amd import fused  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    best = float("inf")
    for _ in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


for M, N, K in [(524288, 128, 128), (1048576, 128, 64), (262144, 256, 128), (524288, 128, 132)]:
    dY = torch.randn(M, N, device=DEV)
    A = torch.randn(M, K, device=DEV)
    row = "M=%8d N=%4d K=%4d" % (M, N, K)
    for terms in (0, 9, 6, 1):
        us = timeit(lambda: fused.gemm_tn(dY, A, x3=terms))
        row += "  terms %d: %6.1f us (%4.2f TB/s)" % (terms, us, 4.0 * M * (N + K) / us / 1e6)
    # a plain copy of the same bytes for scale
    both = torch.cat([dY.reshape(-1), A.reshape(-1)])
    us = timeit(lambda: both.sum())
    row += "  | torch sum of the same bytes %6.1f us" % us
    print(row, flush=True)
