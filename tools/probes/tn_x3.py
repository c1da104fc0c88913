"""Weight-gradient contraction on the bf16 matrix pipe (csrc/gemm_tn_x3.hip) against the fp32 MFMA kernel and float64:
error of both against a float64 product, exactness on small-integer data, device time per shape (HIP events).

    python tools/probes/tn_x3.py [reps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torch_points3d_amd import fused  # noqa: E402

DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timeit(fn):
    """best of five rounds of `reps` back-to-back launches (the box's clock and memory state drift between rounds)"""
    for _ in range(3):
        fn()
    best = float("inf")
    for _ in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps * 1e3)
    return best


shapes = [(524288, 128, 128), (524288, 128, 132), (1048576, 128, 64), (262144, 256, 128), (1048576, 64, 64),
          (262144, 128, 132), (262144, 128, 128), (65536, 64, 64), (40000, 128, 160), (70001, 192, 100), (32768, 64, 68)]
for M, N, K in shapes:
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g).to(DEV)
    A = torch.randn(M, K, generator=g).to(DEV)
    ref = None
    if M <= 300000:
        ref = torch.mm(dY.double().t(), A.double())
    row = "M=%8d N=%4d K=%4d" % (M, N, K)
    for name, terms in (("fp32", 0), ("x3/9", 9), ("x3/6", 6)):
        out = fused.gemm_tn(dY, A, x3=terms)
        us = timeit(lambda: fused.gemm_tn(dY, A, x3=terms))
        err = float((out.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
        same = torch.equal(out, fused.gemm_tn(dY, A, x3=terms))
        row += "  | %s %7.1f us %5.1f TF %4.2f TB/s err %.1e%s" % (
            name, us, 2.0 * M * N * K / us / 1e6, 4.0 * M * (N + K) / us / 1e6, err, "" if same else " NOT-REPRODUCIBLE")
    print(row, flush=True)
    # integers: any operand / accumulator layout mix-up shows as a wrong integer
    Mi = min(M, 70000)
    dYi = torch.randint(-3, 4, (Mi, N), generator=g).float().to(DEV)
    Ai = torch.randint(-3, 4, (Mi, K), generator=g).float().to(DEV)
    want = torch.mm(dYi.double().t(), Ai.double()).float()
    for terms in (9, 6):
        got = fused.gemm_tn(dYi, Ai, x3=terms)
        if not torch.equal(got, want):
            bad = (got != want).nonzero()
            print("   INTEGER MISMATCH terms=%d: %d of %d entries, first at %s got %s want %s" % (
                terms, bad.shape[0], want.numel(), bad[0].tolist(), float(got[tuple(bad[0])]), float(want[tuple(bad[0])])))
# special values: inf / nan / tiny operands behave as in an fp32 product
M, N, K = 40000, 64, 64
dY = torch.randn(M, N, device=DEV)
A = torch.randn(M, K, device=DEV)
dY[5, 3] = float("inf")
dY[7, 9] = float("nan")
A[11, 2] = 1e-38
A[12, 4] = -float("inf")
a, b = fused.gemm_tn(dY, A, x3=9), fused.gemm_tn(dY, A, x3=0)
print("special values: same finite mask", bool(torch.equal(torch.isfinite(a), torch.isfinite(b))),
      "same nan mask", bool(torch.equal(torch.isnan(a), torch.isnan(b))),
      "same inf entries", bool(torch.equal(torch.where(torch.isinf(a), a, torch.zeros_like(a)), torch.where(torch.isinf(b), b, torch.zeros_like(b)))))
