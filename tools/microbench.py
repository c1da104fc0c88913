"""Kernel micro-benchmarks (device time via HIP events on the launch stream). Usage:
   python tools/microbench.py gemm_tn|fps|ball|rows|all [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch_points3d_amd import _lib, fused, torchpoints as tp  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def bench_gemm_tn(reps):
    for M, N, K in [(524288, 128, 128), (524288, 128, 131), (262144, 128, 131), (262144, 256, 128), (1048576, 128, 64),
                    (1048576, 64, 6), (1048576, 64, 64), (524288, 10, 128), (262144, 128, 128), (4096, 1024, 512),
                    (4096, 512, 256), (4096, 256, 259), (16384, 256, 1280)]:
        dY = torch.randn(M, N, device=DEV)
        A = torch.randn(M, K, device=DEV)
        t = timeit(lambda: fused.gemm_tn(dY, A), reps)
        t2 = timeit(lambda: torch.mm(dY.t(), A), max(2, reps // 4))
        fl = 2.0 * M * N * K
        by = 4.0 * M * (N + K)
        print("gemm_tn M=%8d N=%4d K=%4d  %8.1f us  %6.1f TF/s  %6.2f TB/s   (torch.mm %8.1f us)" % (
            M, N, K, t * 1e3, fl / t / 1e9, by / t / 1e9, t2 * 1e3))


def bench_gemm_rows(reps):
    for M, N, K in [(524288, 128, 132), (524288, 128, 128), (1048576, 64, 8), (1048576, 64, 64), (1048576, 128, 64),
                    (262144, 128, 132), (262144, 128, 128), (262144, 256, 128), (16384, 256, 1280), (16384, 128, 384),
                    (524288, 132, 128), (1048576, 8, 64), (4096, 1024, 512), (4096, 256, 1280), (4096, 256, 260), (4096, 512, 256)]:
        A = torch.randn(M, K, device=DEV)
        Bm = torch.randn(K, N, device=DEV)
        Bt = Bm.t().contiguous()
        t = timeit(lambda: fused.gemm_rows(A, Bt, want_stats=True), reps)
        t0 = timeit(lambda: fused.gemm_rows(A, Bt, want_stats=False), reps)
        t2 = timeit(lambda: torch.mm(A, Bm), reps)
        fl = 2.0 * M * N * K
        print("gemm_rows M=%8d N=%4d K=%4d  %8.1f us (+stats %8.1f)  %6.1f TF/s   torch.mm %8.1f us %6.1f TF/s" % (
            M, N, K, t0 * 1e3, t * 1e3, fl / t0 / 1e9, t2 * 1e3, fl / t2 / 1e9))


def bench_fps(reps):
    for B, N, n in [(32, 16384, 512), (32, 512, 128), (32, 2048, 512), (32, 4096, 1024), (8, 16384, 2048)]:
        pos = torch.rand(B, N, 3, device=DEV) * 2 - 1
        t = timeit(lambda: tp.furthest_point_sample(pos, n), reps)
        print("fps B=%d N=%d npoint=%d  %8.1f us  (%.2f us/step)" % (B, N, n, t * 1e3, t * 1e3 / n))


def bench_ball(reps):
    for B, N, n, r, ns in [(32, 16384, 512, 0.2, 64), (32, 512, 128, 0.4, 64), (32, 2048, 512, 0.1, 32), (32, 16384, 2048, 0.2, 64)]:
        pos = torch.rand(B, N, 3, device=DEV) * 2 - 1
        q = pos[:, :n].contiguous()
        t = timeit(lambda: tp.ball_query(r, ns, pos, q), reps)
        by = B * (N * 12 + n * 12 + n * ns * 12)
        print("ball_query B=%d N=%d np=%d r=%.2f ns=%d  %8.1f us  %7.1f GB/s algorithmic" % (B, N, n, r, ns, t * 1e3, by / t / 1e6))
        t = timeit(lambda: tp.three_nn(pos, q), reps)
        sel = tp.furthest_point_sample(pos, n)
        qf = torch.gather(pos, 1, sel.unsqueeze(-1).expand(-1, -1, 3)).contiguous()  # the decoder's case: FPS subset
        t2 = timeit(lambda: tp.three_nn(pos, qf), reps)
        print("three_nn   n=%d m=%d  %8.1f us (known = first m)  %8.1f us (known = FPS subset)" % (N, n, t * 1e3, t2 * 1e3))


def bench_kpconv(reps):
    """BASELINE config 4 shapes: one cloud of 65536 points, partial-dense radius search + KPConv rigid forward."""
    from torch_points3d_amd.kpconv import KPConv_ops
    N = 65536
    pos = torch.rand(N, 3, device=DEV) * 2.0  # 2 m cube
    batch = torch.zeros(N, dtype=torch.long, device=DEV)
    for Mn, Cin, Cout, r in [(25, 64, 128, 0.06), (25, 1, 64, 0.06), (38, 128, 256, 0.08)]:
        t = timeit(lambda: tp.ball_query(r, Mn, pos, pos, mode="partial_dense", batch_x=batch, batch_y=batch), reps)
        nbr, _ = tp.ball_query(r, Mn, pos, pos, mode="partial_dense", batch_x=batch, batch_y=batch)
        filled = float((nbr >= 0).float().mean())
        print("partial_dense ball_query N=%d r=%.2f max_num=%d  %8.1f us  (slots filled %.0f%%)" % (N, r, Mn, t * 1e3, 100 * filled))
        x = torch.randn(N, Cin, device=DEV)
        kp = (torch.rand(15, 3, device=DEV) - 0.5) * r
        W = torch.randn(15, Cin, Cout, device=DEV) * 0.1
        with torch.no_grad():
            t = timeit(lambda: KPConv_ops(pos, pos, nbr, x, kp, W, r / 2.5, "linear", "sum"), reps)
        by = N * Mn * (8 + 4 * Cin) + N * 15 * Cin * 4
        print("KPConv_ops   N=%d Mn=%d Cin=%d Cout=%d  %8.1f us  (stage-1 algorithmic %.0f MB)" % (N, Mn, Cin, Cout, t * 1e3, by / 1e6))


def bench_grid(reps):
    """Radius search on single clouds at KPConv density (one point per 0.02 voxel, r = 0.05, 25 slots): in-LDS grid
    build (one workgroup per cloud, clouds up to 65 536 points) and the sort-based build beyond."""
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_kpconv import synthetic_cloud
    for n, clouds in [(8192, 1), (16384, 1), (32768, 1), (65536, 1), (65536, 4), (262144, 16)]:
        pos, batch = synthetic_cloud(n, clouds, 0.02)
        pos, batch = pos.to(DEV), batch.to(DEV)
        t = timeit(lambda: tp.ball_query(0.05, 25, pos, pos, mode="partial_dense", batch_x=batch, batch_y=batch), reps)
        print("ball_query partial N=%d clouds=%d %8.1f us" % (n, clouds, t * 1e3))


def bench_kpconv_bwd(reps):
    from torch_points3d_amd.kpconv import KPConv_ops
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from bench_kpconv import synthetic_cloud
    pos, batch = synthetic_cloud(65536, 1, 0.02)
    pos, batch = pos.to(DEV), batch.to(DEV)
    nbr, _ = tp.ball_query(0.05, 25, pos, pos, mode="partial_dense", batch_x=batch, batch_y=batch)
    for Cin, Cout in [(32, 32), (64, 64), (128, 128)]:
        x = torch.randn(pos.shape[0], Cin, device=DEV, requires_grad=True)
        kp = (torch.rand(15, 3, device=DEV) - 0.5) * 0.06
        W = (torch.randn(15, Cin, Cout, device=DEV) * 0.1).requires_grad_(True)
        gout = torch.randn(pos.shape[0], Cout, device=DEV)

        def fb():
            out = KPConv_ops(pos, pos, nbr, x, kp, W, 0.02, "linear", "sum")
            out.backward(gout)
            x.grad = None
            W.grad = None
        t = timeit(fb, reps)
        with torch.no_grad():
            tf = timeit(lambda: KPConv_ops(pos, pos, nbr, x, kp, W, 0.02, "linear", "sum"), reps)
        print("KPConv_ops fwd+bwd N=65536 Mn=25 Cin=%d Cout=%d  fwd %8.1f us  fwd+bwd %8.1f us" % (Cin, Cout, tf * 1e3, t * 1e3))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    _lib.load()
    if what in ("gemm_tn", "all"):
        bench_gemm_tn(reps)
    if what in ("gemm_rows", "all"):
        bench_gemm_rows(reps)
    if what in ("kpconv", "all"):
        bench_kpconv(reps)
    if what in ("fps", "all"):
        bench_fps(reps)
    if what in ("ball", "all"):
        bench_ball(reps)
    if what in ("grid", "all"):
        bench_grid(reps)
    if what in ("kpconv_bwd", "all"):
        bench_kpconv_bwd(reps)
