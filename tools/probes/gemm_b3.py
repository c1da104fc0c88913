"""The bf16x3 rows GEMM (tools/probes/gemm_rows_b3.hip: fp32 contraction as six bf16 MFMA term pairs, split done by the loader
waves) against the shipped fp32-MFMA rows kernel: error against float64, then time on the grouped-MLP shapes."""
import ctypes
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_points3d_amd import _lib, fused  # noqa: E402

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


def build():
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    src = os.path.join(root, "torch_points3d_amd", "csrc")
    out = "/tmp/gemm_rows_b3.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans",
                    "-fPIC", "-shared", "-I" + src, "-I" + os.path.join(root, "include"),
                    os.path.join(root, "tools", "probes", "gemm_rows_b3.hip"), os.path.join(src, "api.hip"), "-o", out],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return ctypes.CDLL(out)


def main():
    f = build().tp3d_gemm_rows_b3_f32
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p]
    g = torch.Generator().manual_seed(0)
    for M, N, K in [(1000, 128, 128), (4097, 96, 68), (300, 256, 132), (70000, 128, 36)]:
        A = (torch.randn(M, K, generator=g) * 1.3 + 0.2).to(DEV)
        Bt = (torch.randn(N, K, generator=g) * 0.2).to(DEV)
        C = torch.full((M, N), float("nan"), device=DEV)
        ref64 = A.double() @ Bt.double().t()
        for grid in (512, 256):
            C.fill_(float("nan"))
            rc = f(A.data_ptr(), Bt.data_ptr(), M, N, K, C.data_ptr(), grid, None, None, None, 0.0, _lib.stream_ptr(A.device))
            assert rc == 0, rc
            assert float((C.double() - ref64).abs().max()) < 2e-6 * float(ref64.abs().max()), (M, N, K, grid)
        c32 = fused.gemm_rows(A, Bt)[0]
        scale = float(ref64.abs().max())
        print("M=%6d N=%3d K=%3d  max error / max|C|: bf16x3 %.2e   fp32 MFMA %.2e" % (
            M, N, K, float((C.double() - ref64).abs().max()) / scale, float((c32.double() - ref64).abs().max()) / scale), flush=True)
    for M, N, K in [(524288, 128, 128), (262144, 128, 128), (262144, 256, 128), (1048576, 128, 64), (65536, 256, 256), (65536, 512, 256)]:
        A = torch.randn(M, K, device=DEV)
        Bt = torch.randn(N, K, device=DEV)
        C = torch.empty(M, N, device=DEV)
        t0 = timeit(lambda: fused.gemm_rows(A, Bt))
        line = "M=%7d N=%3d K=%3d  fp32 MFMA rows kernel %7.1f us %6.1f TF |" % (M, N, K, t0, 2.0 * M * N * K / t0 / 1e6)
        for grid in (256, 512):
            t = timeit(lambda: f(A.data_ptr(), Bt.data_ptr(), M, N, K, C.data_ptr(), grid, None, None, None, 0.0, _lib.stream_ptr(A.device)))
            line += "  bf16x3 (grid %d) %7.1f us %6.1f TF-equivalent" % (grid, t, 2.0 * M * N * K / t / 1e6)
        print(line, flush=True)
    # a hidden layer as the step runs it: BatchNorm + LeakyReLU of the previous layer, then the contraction
    h = _lib.load()
    for M, N, K in [(524288, 128, 128), (262144, 256, 128), (1048576, 128, 64), (262144, 128, 128)]:
        Y = torch.randn(M, K, device=DEV)
        Bt = torch.randn(N, K, device=DEV) * 0.1
        mean, scale, beta = torch.randn(K, device=DEV) * 0.1, torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
        act = torch.empty_like(Y)
        C0, C1, C2 = (torch.empty(M, N, device=DEV) for _ in range(3))
        st = _lib.stream_ptr(Y.device)

        def separate():
            _lib.call("tp3d_bn_act_f32", Y.data_ptr(), mean.data_ptr(), scale.data_ptr(), beta.data_ptr(), 0.01, M, K, act.data_ptr(), st)
            _lib.call("tp3d_gemm_rows_f32", act.data_ptr(), Bt.data_ptr(), M, N, K, C0.data_ptr(), None, None, st)

        def shipped():
            _lib.call("tp3d_gemm_rows_bnact_sp_f32", Y.data_ptr(), mean.data_ptr(), scale.data_ptr(), beta.data_ptr(), 0.01, Bt.data_ptr(),
                      M, N, K, C1.data_ptr(), None, None, 0, st)

        def b3(grid=256):
            rc = f(Y.data_ptr(), Bt.data_ptr(), M, N, K, C2.data_ptr(), grid, mean.data_ptr(), scale.data_ptr(), beta.data_ptr(), 0.01, st)
            assert rc == 0, rc
        separate(), shipped(), b3()
        ref = act.double() @ Bt.double().t()
        sc_ = float(ref.abs().max())
        e1, e2 = float((C1.double() - ref).abs().max()) / sc_, float((C2.double() - ref).abs().max()) / sc_
        ts, t1, t2 = timeit(separate), timeit(shipped), timeit(b3)
        b3(512)
        e3 = float((C2.double() - ref).abs().max()) / sc_
        t3 = timeit(lambda: b3(512))
        print("M=%7d N=%3d K=%3d  pass + GEMM %7.1f us | shipped split-role (fp32 MFMA) %7.1f us (err %.1e) | bf16x3 split-role "
              "%7.1f us (err %.1e), two workgroups per CU %7.1f us (err %.1e)" % (M, N, K, ts, t1, e1, t2, e2, t3, e3), flush=True)


if __name__ == "__main__":
    main()
