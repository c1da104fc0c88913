"""Numerical side of the untried bf16x3 route, on the CPU: an fp32 contraction written as products of
bf16 terms with fp32 accumulation (what v_mfma_f32_*_bf16 computes), against float64.  Products of two bf16 values are
exact in fp32, so torch's fp32 matmul on bf16-representable operands stands in for the bf16 MFMA."""
import torch

torch.manual_seed(0)
M, K, N = 20000, 128, 128
A = torch.randn(M, K) * 1.3 + 0.2
B = torch.randn(K, N) * 0.2
ref = A.double() @ B.double()


def split(x):
    hi = x.bfloat16().float()
    r = x - hi
    mid = r.bfloat16().float()
    lo = (r - mid).bfloat16().float()
    return hi, mid, lo


def err(C):
    d = (C.double() - ref).abs()
    return float(d.max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


a, b = split(A), split(B)
print("fp32 matmul                 max %.2e  rms %.2e" % err(A @ B))
for name, terms in (("three bf16 terms, 6 products", [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]),
                    ("two bf16 terms, 3 products ", [(0, 0), (0, 1), (1, 0)])):
    C = sum(a[i] @ b[j] for i, j in reversed(terms))  # small terms first
    print("%s max %.2e  rms %.2e" % ((name,) + err(C)))
