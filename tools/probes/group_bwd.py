"""grouping_operation / three_interpolate backward (the API ops, reference layout) on ball-query tables with hub points."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_points3d_amd import torchpoints as tp  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=10):
    for _ in range(2):
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
    g = torch.Generator().manual_seed(0)
    for B, N, npnt, r, ns, C in [(32, 16384, 512, 0.2, 64, 64), (32, 512, 128, 0.8, 128, 128), (32, 512, 128, 0.4, 64, 128),
                                 (32, 2048, 512, 0.1, 32, 64)]:
        pos = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(DEV)
        q = torch.gather(pos, 1, tp.furthest_point_sample(pos, npnt).unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        idx = tp.ball_query(r, ns, pos, q)[0]
        runs = torch.stack([torch.bincount(idx[b].reshape(-1), minlength=N) for b in range(B)])
        feat = torch.randn(B, C, N, device=DEV, requires_grad=True)
        out = tp.grouping_operation(feat, idx)
        cot = torch.randn_like(out)
        rnd = torch.randint(0, N, idx.shape, generator=g).to(DEV)
        out_r = tp.grouping_operation(feat, rnd)
        t = timeit(lambda: torch.autograd.grad(out, feat, cot, retain_graph=True))
        tr = timeit(lambda: torch.autograd.grad(out_r, feat, cot, retain_graph=True))
        tf = timeit(lambda: tp.grouping_operation(feat, idx))
        print("B=%d N=%d np=%d r=%.1f ns=%d C=%d  max slots/point %5d   fwd %7.1f us   bwd %8.1f us (random table of the same size %8.1f us)" % (
            B, N, npnt, r, ns, C, int(runs.max()), tf, t, tr))


if __name__ == "__main__":
    main()
