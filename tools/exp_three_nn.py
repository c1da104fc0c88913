"""Tuning sweep for the in-LDS grid form of three_nn: compiles csrc/three_nn.hip with different workgroup shapes and
cell occupancies into /tmp (on the GPU box) and times each on the decoder shape of the headline workload."""
import ctypes
import itertools
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch_points3d_amd import torchpoints as tp  # noqa: E402

SRC = os.path.join(ROOT, "torch_points3d_amd", "csrc")
DEV = "cuda:0"


def variant(block, qpt, target):
    out = "/tmp/tnn_%d_%d_%s.so" % (block, qpt, str(target).replace(".", "p"))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans",
           "-fPIC", "-shared", "-I" + SRC, "-I" + os.path.join(ROOT, "include"),
           "-DTP3D_NG_BLOCK=%d" % block, "-DTP3D_NG_QPT=%d" % qpt, "-DTP3D_NG_TARGET=%sf" % target,
           os.path.join(SRC, "three_nn.hip"), os.path.join(SRC, "api.hip"), "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return ctypes.CDLL(out)


def main():
    B, n, m = 32, 16384, 512
    pos = torch.rand(B, n, 3, device=DEV) * 2 - 1
    sel = tp.furthest_point_sample(pos, m)
    known = torch.gather(pos, 1, sel.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    ref_d, ref_i = tp.three_nn(pos, known)
    dist = torch.empty(B, n, 3, device=DEV)
    idx = torch.empty(B, n, 3, dtype=torch.int64, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    for block, qpt, target in itertools.product((512, 1024), (1, 2), (1.0, 1.5, 2.0, 2.5)):
        lib = variant(block, qpt, target)
        f = lib.tp3d_three_nn_f32
        f.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
        f.restype = ctypes.c_int

        def run():
            rc = f(pos.data_ptr(), known.data_ptr(), B, n, m, dist.data_ptr(), idx.data_ptr(), stream)
            assert rc == 0
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            run()
        b.record()
        torch.cuda.synchronize()
        same = bool(torch.equal(idx, ref_i) and torch.equal(dist, ref_d))
        print("block %4d  qpt %d  target %.1f   %7.1f us   same=%s" % (block, qpt, target, a.elapsed_time(b) / 20 * 1e3, same),
              flush=True)


if __name__ == "__main__":
    main()
