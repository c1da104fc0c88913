"""tp3d_gemm_rows_f32 against torch.mm (hipBLASLt) on the few-row GEMMs of the KPConv U-Net's deep levels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch_points3d_amd import fused

DEV = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M, N, K in [(27, 2048, 1024), (216, 512, 3072), (1331, 256, 1024), (216, 256, 1024), (27, 512, 1024), (216, 1024, 256),
                (1331, 128, 256), (1331, 512, 256), (27, 1024, 256), (9261, 64, 256), (9261, 128, 960), (65536, 64, 256),
                (65536, 32, 128), (65536, 64, 960)]:
    A = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) * 0.1
    mine = timeit(lambda: fused.gemm_rows(A, W))
    lib = timeit(lambda: torch.mm(A, W.t()))
    err = float((fused.gemm_rows(A, W)[0] - torch.mm(A, W.t())).abs().max())
    print("M=%6d N=%5d K=%5d  rows kernel %7.1f us   torch.mm %7.1f us   max diff %.1e" % (M, N, K, mine, lib, err), flush=True)
