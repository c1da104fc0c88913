"""BASELINE config 5 leg: random subsample + exact kNN (k = 16) on one S3DIS-shaped scene of N = 10^6 points
(a room: floor, ceiling, four walls and box-shaped furniture sampled on their surfaces), the path RandLA-Net's
down-convolutions run (reference modules/RandLANet/modules.py:57-67, core/base_conv/message_passing.py:44-58).

    python tools/bench_knn.py [--n 1000000] [--k 16] [--ratio 0.25] [--iters 10] [--check 2000]

Prints ms for sampler + kNN, queries/s and the algorithmic HBM bytes / time (support xyz + query xyz read once,
idx + dist2 written once).  --check N verifies N random queries against a brute-force evaluation in plain torch ops.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def room(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 3, generator=g)
    kind = torch.randint(0, 10, (n,), generator=g)
    pts = torch.empty(n, 3)
    L, W, H = 10.0, 8.0, 3.0
    for kd in range(10):
        m = kind == kd
        a = u[m]
        if kd == 0:    p = torch.stack([a[:, 0] * L, a[:, 1] * W, torch.zeros(a.shape[0])], 1)          # floor
        elif kd == 1:  p = torch.stack([a[:, 0] * L, a[:, 1] * W, torch.full((a.shape[0],), H)], 1)    # ceiling
        elif kd == 2:  p = torch.stack([a[:, 0] * L, torch.zeros(a.shape[0]), a[:, 2] * H], 1)
        elif kd == 3:  p = torch.stack([a[:, 0] * L, torch.full((a.shape[0],), W), a[:, 2] * H], 1)
        elif kd == 4:  p = torch.stack([torch.zeros(a.shape[0]), a[:, 1] * W, a[:, 2] * H], 1)
        elif kd == 5:  p = torch.stack([torch.full((a.shape[0],), L), a[:, 1] * W, a[:, 2] * H], 1)
        else:          # furniture: surfaces of boxes
            c = torch.tensor([2.0 + 1.5 * (kd - 6), 2.0 + (kd - 6), 0.5])
            s = torch.tensor([1.2, 0.8, 1.0])
            p = c + (a - 0.5) * s
            face = torch.randint(0, 3, (a.shape[0],), generator=g)
            side = (torch.rand(a.shape[0], generator=g) < 0.5).float() - 0.5
            for ax in range(3):
                sel = face == ax
                p[sel, ax] = c[ax] + side[sel] * s[ax]
        pts[m] = p
    return pts + 0.002 * torch.randn(n, 3, generator=g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--ratio", type=float, default=0.25)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--check", type=int, default=2000)
    ap.add_argument("--conv", action="store_true", help="also time one RandlaConv forward (first layer of randlanet.yaml)")
    args = ap.parse_args()
    from torch_points3d_amd import torchpoints as tp
    from torch_points3d_amd.randla import RandomSampler
    dev = torch.device("cuda:0")
    pos = room(args.n).to(dev)
    batch = torch.zeros(args.n, dtype=torch.long, device=dev)
    sampler = RandomSampler(ratio=args.ratio)

    def step():
        idx = sampler(pos, batch=batch)
        q = pos[idx]
        return idx, q, tp.knn(args.k, pos, q, batch, batch[idx])

    for _ in range(2):
        idx, q, (nbr, d2) = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        idx, q, (nbr, d2) = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.iters
    # kNN alone, device time
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    by = batch[idx]
    a.record()
    for _ in range(args.iters):
        tp.knn(args.k, pos, q, batch, by)
    b.record()
    torch.cuda.synchronize()
    knn_ms = a.elapsed_time(b) / args.iters
    nq = q.shape[0]
    alg_bytes = args.n * 12 + nq * 12 + nq * args.k * (8 + 4)
    out = {"workload": "randla_sample_knn", "points": args.n, "queries": nq, "k": args.k, "ms": round(ms, 3),
           "knn_ms": round(knn_ms, 3), "queries_per_s": round(nq / knn_ms * 1e3),
           "algorithmic_GBps": round(alg_bytes / knn_ms / 1e6, 1)}
    if args.check:
        # brute force on the device in plain torch ops (each op rounds to fp32, same evaluation order as the kernel):
        # the returned distances must be the k smallest of each checked query, and belong to the returned indices
        sel = torch.randperm(nq, device=dev)[: args.check]
        ok = True
        for c0 in range(0, sel.numel(), 100):
            qs = q[sel[c0:c0 + 100]]
            dx = pos[None, :, 0] - qs[:, 0:1]
            dy = pos[None, :, 1] - qs[:, 1:2]
            dz = pos[None, :, 2] - qs[:, 2:3]
            dall = (dx * dx + dy * dy) + dz * dz
            best = torch.topk(dall, args.k, dim=1, largest=False, sorted=True)[0]
            got_i, got_d = nbr[sel[c0:c0 + 100]], d2[sel[c0:c0 + 100]]
            ok = ok and bool(torch.equal(best, got_d)) and bool(torch.equal(torch.gather(dall, 1, got_i), got_d))
        out["checked"] = int(sel.numel())
        out["exact"] = ok
    if args.conv:
        # first down-convolution of conf/models/segmentation/randlanet.yaml (Randlanet_Conv): ratio 0.25, k 16,
        # FEAT = 3 input features
        from torch_points3d_amd.kpconv_blocks import PDData
        from torch_points3d_amd.randla import RandlaConv
        feat = 3
        conv = RandlaConv(ratio=args.ratio, k=args.k, point_pos_nn=[10, 8, feat], attention_nn=[2 * feat, 8, 2 * feat],
                          down_conv_nn=[2 * feat, 8, 16]).to(dev).eval()
        x = torch.randn(args.n, feat, device=dev)
        with torch.no_grad():
            for _ in range(2):
                conv(PDData(pos=pos, batch=batch, x=x))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                o = conv(PDData(pos=pos, batch=batch, x=x))
            torch.cuda.synchronize()
        out["randla_conv_ms"] = round((time.perf_counter() - t0) * 1e3 / args.iters, 3)
        out["edges"] = int(o.neighbors.numel())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
