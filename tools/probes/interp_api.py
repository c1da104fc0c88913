"""three_interpolate (API op, reference layout) forward / backward timings at decoder shapes."""
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
    for B, C, m, n in [(32, 128, 512, 16384), (32, 256, 128, 512), (32, 128, 512, 2048)]:
        pos = (torch.rand(B, n, 3, generator=g) * 2 - 1).to(DEV)
        known = torch.gather(pos, 1, tp.furthest_point_sample(pos, m).unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        dist, idx = tp.three_nn(pos, known)
        w = 1.0 / (dist + 1e-8)
        w = w / w.sum(-1, keepdim=True)
        feat = torch.randn(B, C, m, device=DEV, requires_grad=True)
        out = tp.three_interpolate(feat, idx, w)
        cot = torch.randn_like(out)
        tf = timeit(lambda: tp.three_interpolate(feat, idx, w))
        tb = timeit(lambda: torch.autograd.grad(out, feat, cot, retain_graph=True))
        mb = out.numel() * 4 / 1e6
        print("B=%d C=%d m=%d n=%d   fwd %7.1f us (%5.2f TB/s of output)   bwd %8.1f us" % (B, C, m, n, tf, mb / tf / 1e6 * 1e6 / 1e6 * 1e0, tb))


if __name__ == "__main__":
    main()
