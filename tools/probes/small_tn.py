"""tp3d_gemm_tn_f32 against torch.mm(dY.t(), A) (hipBLASLt) on the few-row weight gradients (global / decoder layers, the head)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch_points3d_amd import fused

DEV = torch.device("cuda:0")


def timeit(fn, n=30):
    # replayed from a graph: the host side of either path is not what a captured training step pays
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M, N, K in [(4096, 1024, 512), (4096, 256, 1280), (16384, 256, 384), (4096, 512, 256), (4096, 256, 260), (16384, 128, 384),
                (16384, 256, 256), (524288, 10, 128), (524288, 128, 12), (65536, 128, 128), (32768, 512, 768), (16384, 512, 1280)]:
    dY = torch.randn(M, N, device=DEV)
    A = torch.randn(M, K, device=DEV)
    mine = timeit(lambda: fused.gemm_tn(dY, A))
    lib = timeit(lambda: torch.mm(dY.t(), A))
    err = float((fused.gemm_tn(dY, A) - torch.mm(dY.t(), A)).abs().max())
    print("M=%7d N=%5d K=%5d  tn kernel %7.1f us   torch.mm %7.1f us   max diff %.1e" % (M, N, K, mine, lib, err), flush=True)
