"""One shape of the bf16-pipe weight-gradient kernel, a few launches: the target of rocprofv3 --pmc passes.
    python3 tools/probes/tn_x3_one.py [terms] [M N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_points3d_amd import fused  # noqa: E402

terms = int(sys.argv[1]) if len(sys.argv) > 1 else 9
M, N, K = [int(v) for v in sys.argv[2:5]] if len(sys.argv) > 4 else (524288, 128, 128)
dY = torch.randn(M, N, device="cuda:0")
A = torch.randn(M, K, device="cuda:0")
for _ in range(6):
    fused.gemm_tn(dY, A, x3=terms)
torch.cuda.synchronize()
