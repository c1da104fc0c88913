"""Which half of the dense ball query holds the time on the headline shape (B=32, N=16384, 512 FPS centres, r=0.2, 64 slots):
run under `rocprofv3 --kernel-trace --stats` -- grid_build_kernel vs grid_query_kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_points3d_amd import torchpoints as tp  # noqa: E402

pos = (torch.rand(32, 16384, 3, device="cuda:0") * 2 - 1).contiguous()
sel = tp.furthest_point_sample(pos, 512)
q = torch.gather(pos, 1, sel.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
for _ in range(30):
    tp.ball_query(0.2, 64, pos, q)
torch.cuda.synchronize()
