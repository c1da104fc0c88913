"""Where does tp3d_rows_scatter_apply_f32 spend 0.65 ms on the charlesmsg SA2 table (B=32, 128 centres x 128 slots over 512
points, 320 channels)?  Times the gather on that table, on a random table of the same size, and at other widths."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_points3d_amd import _lib, fused, torchpoints as tp  # noqa: E402

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
    B, N = 32, 2048
    g = torch.Generator().manual_seed(0)
    pos = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(DEV)
    p1 = torch.gather(pos, 1, tp.furthest_point_sample(pos, 512).unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    p2 = torch.gather(p1, 1, tp.furthest_point_sample(p1, 128).unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    for r, ns in [(0.8, 128), (0.4, 64)]:
        idx = tp.ball_query(r, ns, p1, p2)[0]
        runs = torch.stack([torch.bincount(idx[b].reshape(-1), minlength=512) for b in range(B)])
        print("r=%.1f ns=%d: slots per support point: mean %.1f max %d; points with none %.0f%%" % (
            r, ns, runs.float().mean(), int(runs.max()), 100 * float((runs == 0).float().mean())))
        rnd = torch.randint(0, 512, idx.shape, generator=g).to(DEV)
        for name, table_idx in (("ball-query table", idx), ("uniform random table", rnd)):
            table = fused.scatter_table(table_idx, None, 512, 1)
            for C, ld in [(320, 324), (256, 260), (128, 132), (64, 68)]:
                rows = torch.randn(B * 128 * ns, ld, device=DEV)
                out = torch.empty(B, 512, C, device=DEV)

                def run():
                    _lib.call("tp3d_rows_scatter_apply_f32", rows.data_ptr(), B, 128 * ns, 1, 512, ld, 3 if ld - C >= 3 else 0, C, 0,
                              out.data_ptr(), table.data_ptr(), table.numel(), _lib.stream_ptr(rows.device))
                t = timeit(run)
                print("  %-22s C=%3d ld=%3d  %8.1f us  %6.2f TB/s" % (name, C, ld, t, rows.numel() * 4 * C / ld / t / 1e6))


if __name__ == "__main__":
    main()
