"""tp3d_bn_act_maxpool_f32 on the set-abstraction shapes of the BASELINE step and of config 3: time and read rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_points3d_amd import _lib  # noqa: E402

DEV = "cuda:0"
for G, ns, C in [(16384, 64, 128), (4096, 64, 256), (16384, 128, 128), (16384, 32, 64), (4096, 128, 256)]:
    Y = torch.randn(G * ns, C, device=DEV)
    v = [torch.rand(C, device=DEV) + 0.5 for _ in range(3)]
    out = torch.empty(G, C, device=DEV)
    arg = torch.empty(G, C, dtype=torch.int32, device=DEV)
    st = _lib.stream_ptr(Y.device)

    def run():
        _lib.call("tp3d_bn_act_maxpool_f32", Y.data_ptr(), v[0].data_ptr(), v[1].data_ptr(), v[2].data_ptr(), 0.01, G, ns, C,
                  out.data_ptr(), arg.data_ptr(), st)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        run()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / 30 * 1e3
    print("G=%6d ns=%3d C=%3d  %7.1f us  %5.2f TB/s" % (G, ns, C, t, 4.0 * G * ns * C / t / 1e6), flush=True)
