"""Split-role rows GEMM (tools/probes/gemm_rows_sp.hip: four MFMA waves fed by four loader waves; an experiment, not part of
libtp3d_hip.so -- compiled here into /tmp on the GPU box) against the shipped rows kernel on the grouped-MLP shapes of
the BASELINE step.  Checks the result against an fp64 product first, then asks what bounds it: input served from cache,
stores skipped, staggered workgroup starts.  Round-2 outcome in DESIGN.md section 5."""
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
    out = "/tmp/gemm_rows_sp.so"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans",
                    "-fPIC", "-shared", "-I" + src, "-I" + os.path.join(root, "include"),
                    os.path.join(root, "tools", "probes", "gemm_rows_sp.hip"), os.path.join(src, "api.hip"), "-o", out],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return ctypes.CDLL(out)


def prologue_cases(lib):
    """BatchNorm + LeakyReLU between two layers: separate pass + plain GEMM (shipped), the prologue GEMM of the fused chain
    (registers of the MFMA waves), the prologue in the loader waves of the split-role kernel."""
    g = lib.tp3d_gemm_rows_sp_bnact_f32
    g.restype = ctypes.c_int
    g.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_float, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    for M, N, K in [(524288, 128, 128), (262144, 256, 128), (1048576, 128, 64), (262144, 128, 128), (65536, 512, 256)]:
        Y = torch.randn(M, K, device=DEV)
        Bt = torch.randn(N, K, device=DEV) * 0.1
        mean, scale, beta = torch.randn(K, device=DEV) * 0.1, torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV) * 0.1
        act = torch.empty_like(Y)
        C0, C1, C2 = (torch.empty(M, N, device=DEV) for _ in range(3))
        st = _lib.stream_ptr(Y.device)

        def separate():
            _lib.call("tp3d_bn_act_f32", Y.data_ptr(), mean.data_ptr(), scale.data_ptr(), beta.data_ptr(), 0.01, M, K, act.data_ptr(), st)
            _lib.call("tp3d_gemm_rows_f32", act.data_ptr(), Bt.data_ptr(), M, N, K, C0.data_ptr(), None, None, st)

        def chain():
            _lib.call("tp3d_gemm_rows_bnact_f32", Y.data_ptr(), mean.data_ptr(), scale.data_ptr(), beta.data_ptr(), 0.01, Bt.data_ptr(),
                      M, N, K, C1.data_ptr(), None, st)

        def split():
            rc = g(Y.data_ptr(), mean.data_ptr(), scale.data_ptr(), beta.data_ptr(), 0.01, Bt.data_ptr(), M, N, K, C2.data_ptr(), 512, st)
            assert rc == 0, rc
        separate(), chain(), split()
        e1, e2 = float((C1 - C0).abs().max()), float((C2 - C0).abs().max())
        ts, tc, tp_ = timeit(separate), timeit(chain), timeit(split)
        print("M=%7d N=%3d K=%3d  pass + GEMM %7.1f us | prologue in the MFMA waves %7.1f us (diff %.1e) | in the loader waves "
              "%7.1f us (diff %.1e)" % (M, N, K, ts, tc, e1, tp_, e2), flush=True)


def main():
    lib = build()
    f = lib.tp3d_gemm_rows_sp_f32
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                  ctypes.c_int, ctypes.c_void_p]
    g = torch.Generator().manual_seed(0)
    # exactness on ragged shapes: small integers make every partial sum exact in fp32
    for M, N, K in [(1000, 128, 128), (4097, 96, 68), (300, 256, 132), (128 * 9 + 5, 64, 4), (70000, 128, 36)]:
        A = torch.randint(-4, 5, (M, K), generator=g).float().to(DEV)
        Bt = torch.randint(-4, 5, (N, K), generator=g).float().to(DEV)
        C = torch.full((M, N), float("nan"), device=DEV)
        for grid in (8, 256):
            rc = f(A.data_ptr(), Bt.data_ptr(), M, N, K, C.data_ptr(), grid, 0, _lib.stream_ptr(A.device))
            assert rc == 0, rc
            ref = (A.double() @ Bt.double().t()).float()
            assert torch.equal(C, ref), (M, N, K, grid, float((C - ref).abs().max()))
    print("exact on the ragged integer cases", flush=True)
    for M, N, K in [(524288, 128, 128), (524288, 128, 64), (524288, 64, 64), (262144, 128, 128), (262144, 256, 128),
                    (1048576, 128, 64), (524288, 128, 132), (65536, 256, 256), (65536, 512, 256)]:
        A = torch.randn(M, K, device=DEV)
        Bt = torch.randn(N, K, device=DEV)
        C = torch.empty(M, N, device=DEV)
        ref = fused.gemm_rows(A, Bt)[0]
        t0 = timeit(lambda: fused.gemm_rows(A, Bt))
        line = "M=%7d N=%3d K=%3d  rows kernel %7.1f us %6.1f TF |" % (M, N, K, t0, 2.0 * M * N * K / t0 / 1e6)
        for grid in (256, 512):
            def run():
                f(A.data_ptr(), Bt.data_ptr(), M, N, K, C.data_ptr(), grid, 0, _lib.stream_ptr(A.device))
            run()
            err = float((C - ref).abs().max())
            t = timeit(run)
            line += "  split grid %d %7.1f us %6.1f TF (max diff %.1e)" % (grid, t, 2.0 * M * N * K / t / 1e6, err)
        print(line, flush=True)
        line = "    what bounds it (grid 512):"
        for probe, what in ((1, "input from cache"), (2, "no stores"), (3, "neither")):
            t = timeit(lambda: f(A.data_ptr(), Bt.data_ptr(), M, N, K, C.data_ptr(), 512, probe, _lib.stream_ptr(A.device)))
            line += "  %s %7.1f us %6.1f TF" % (what, t, 2.0 * M * N * K / t / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
    prologue_cases(build())
