"""ONE north-star kernel at ONE shape, a few launches: the target of the rocprofv3 --pmc passes behind
profiles/pmc_north_star.json (one process per shape, so a row's counters belong to its shape alone).

    python3 tools/pmc_north_star.py CASE        CASE in: fps_sa1 fps_sa2 ball_sa1 ball_sa2 three_nn_fp3 three_nn_fp2
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_points3d_amd import torchpoints as tp  # noqa: E402

CASES = {  # entry point, the size arguments bench.py records for it
    "fps_sa1": ("tp3d_fps_f32", (32, 16384, 512)),
    "fps_sa2": ("tp3d_fps_f32", (32, 512, 128)),
    "ball_sa1": ("tp3d_ball_query_dense_f32", (32, 16384, 512, 64)),
    "ball_sa2": ("tp3d_ball_query_dense_f32", (32, 512, 128, 64)),
    "three_nn_fp3": ("tp3d_three_nn_f32", (32, 16384, 512)),
    "three_nn_fp2": ("tp3d_three_nn_f32", (32, 512, 128)),
}


def main():
    case = sys.argv[1]
    entry, sizes = CASES[case]
    dev = "cuda:0"
    g = torch.Generator().manual_seed(1234)
    B, N = sizes[0], sizes[1]
    pos = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
    if entry == "tp3d_fps_f32":
        fn = lambda: tp.furthest_point_sample(pos, sizes[2])  # noqa: E731
    else:
        m = sizes[2]
        sub = pos[:, torch.randperm(N, generator=g)[:m].to(dev)].contiguous()
        if entry == "tp3d_ball_query_dense_f32":
            r = 0.2 if N == 16384 else 0.4
            fn = lambda: tp.ball_query(r, sizes[3], pos, sub)  # noqa: E731
        else:
            fn = lambda: tp.three_nn(pos, sub)  # noqa: E731
    for _ in range(8):
        fn()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
