"""BASELINE config 4: KPConv rigid segmentation forward on one synthetic cloud of N = 65536 points (one point per
0.02 voxel, i.e. a cloud that already went through the dataset's first grid subsampling), unet_4 architecture
(in_feat 64, in_grid_size 0.02, 25 neighbours).  Prints ms per forward (and forward+backward with --train) plus the
per-entry-point device time of the library calls.

    python tools/bench_kpconv.py [--n 65536] [--iters 20] [--train] [--clouds 1]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def synthetic_cloud(n, clouds, grid, seed=0):
    g = torch.Generator().manual_seed(seed)
    per = n // clouds
    side = int(round(per ** (1.0 / 3.0))) + 1
    pts, bs = [], []
    for b in range(clouds):
        cells = torch.stack(torch.meshgrid(torch.arange(side), torch.arange(side), torch.arange(side), indexing="ij"), -1)
        cells = cells.reshape(-1, 3)[torch.randperm(side ** 3, generator=g)[:per]].float()
        pts.append((cells + 0.1 + 0.8 * torch.rand(per, 3, generator=g)) * grid)
        bs.append(torch.full((per,), b, dtype=torch.long))
    return torch.cat(pts), torch.cat(bs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--clouds", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--in-feat", type=int, default=64)
    ap.add_argument("--train", action="store_true", help="time forward + backward in training mode")
    args = ap.parse_args()
    from torch_points3d_amd import _lib
    from torch_points3d_amd.kpconv_blocks import PDData
    from torch_points3d_amd.kpconv_unet import KPConv
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = KPConv("unet", input_nc=3, in_feat=args.in_feat, in_grid_size=0.02, num_layers=4, output_nc=13).to(dev)
    pos, batch = synthetic_cloud(args.n, args.clouds, 0.02)
    x = torch.cat([torch.ones(pos.shape[0], 1), torch.randn(pos.shape[0], 3)], 1)
    pos, batch, x = pos.to(dev), batch.to(dev), x.to(dev)
    model.train(args.train)

    def step():
        data = PDData(pos=pos, batch=batch, x=x)
        if args.train:
            out = model(data)
            out.x.square().mean().backward()
            for p in model.parameters():
                p.grad = None
        else:
            with torch.no_grad():
                out = model(data)
        return out

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        out = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.iters
    timer = _lib.KernelTimer()
    _lib.set_timer(timer)
    step()
    torch.cuda.synchronize()
    _lib.set_timer(None)
    per_entry = {}
    for (name, _), (cnt, tot) in timer.summary().items():
        c, t = per_entry.get(name, (0, 0.0))
        per_entry[name] = (c + cnt, t + tot)
    top = sorted(per_entry.items(), key=lambda kv: -kv[1][1])
    print(json.dumps({"workload": "kpconv_unet4_%s" % ("train" if args.train else "forward"), "points": pos.shape[0],
                      "clouds": args.clouds, "in_feat": args.in_feat, "ms": round(ms, 3),
                      "points_per_s": round(pos.shape[0] / ms * 1e3),
                      "library_ms": round(sum(v[1] for v in per_entry.values()), 3),
                      "entries": {k: [v[0], round(v[1], 3)] for k, v in top}}))


if __name__ == "__main__":
    main()
