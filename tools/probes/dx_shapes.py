"""Input-gradient contraction dA (M, Cin) = dY (M, Cout) @ W (Cout, Cin) on the shapes of the BASELINE step: the vendor
library (torch.mm) against the rows kernel, per shape -- is a per-shape choice worth anything?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_points3d_amd import fused  # noqa: E402

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


def main():
    # (rows, Cin, Cout) of the layers whose input gradient is needed (the first layer of SA1 has none)
    shapes = [(1048576, 64, 64), (1048576, 64, 128), (524288, 132, 128), (524288, 128, 128), (524288, 128, 256),
              (262144, 260, 256), (262144, 256, 256), (262144, 256, 512), (4096, 516, 512), (4096, 512, 1024),
              (16384, 384, 256), (16384, 256, 256), (524288, 128, 128), (524288, 128, 10), (4096, 1536, 512),
              (4096, 512, 512), (262144, 132, 128)]
    tot_lib = tot_rows = tot_best = 0.0
    for M, cin, cout in shapes:
        dY = torch.randn(M, cout, device=DEV)
        W = torch.randn(cout, cin, device=DEV)
        Wt = W.t().contiguous()
        t_lib = timeit(lambda: torch.mm(dY, W))
        if cout % 4 == 0 and cin >= 32:
            t_rows = timeit(lambda: fused.gemm_rows(dY, Wt))
        else:
            t_rows = float("inf")
        tot_lib += t_lib
        tot_rows += min(t_rows, t_lib) if t_rows == float("inf") else t_rows
        tot_best += min(t_lib, t_rows)
        print("M=%7d Cin=%4d Cout=%4d   library %7.1f us %6.1f TF   rows kernel %7.1f us %6.1f TF" % (
            M, cin, cout, t_lib, 2.0 * M * cin * cout / t_lib / 1e6, t_rows, 2.0 * M * cin * cout / t_rows / 1e6), flush=True)
    print("sum: library %.1f us, rows kernel %.1f us, best of both per shape %.1f us" % (tot_lib, tot_rows, tot_best))


if __name__ == "__main__":
    main()
