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
    ap.add_argument("--precomputed", action="store_true",
                    help="sampling / neighbour search / interpolation tables computed once by MultiScaleTransform "
                         "(the reference's data-loader precompute, on the device); the timed forward only convolves")
    ap.add_argument("--graph", action="store_true", help="with --precomputed: replay the forward from one HIP graph")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="experiment switch: an attribute of torch_points3d_amd.fused (as bench.py --set)")
    args = ap.parse_args()
    from torch_points3d_amd import _lib, fused
    for kv in args.set:
        name, val = kv.split("=")
        setattr(fused, name, type(getattr(fused, name))(int(val)))
    from torch_points3d_amd.kpconv_blocks import PDData
    from torch_points3d_amd.kpconv_unet import KPConv
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = KPConv("unet", input_nc=3, in_feat=args.in_feat, in_grid_size=0.02, num_layers=4, output_nc=13).to(dev)
    pos, batch = synthetic_cloud(args.n, args.clouds, 0.02)
    x = torch.cat([torch.ones(pos.shape[0], 1), torch.randn(pos.shape[0], 3)], 1)
    pos, batch, x = pos.to(dev), batch.to(dev), x.to(dev)
    model.train(args.train)
    tables = None
    pre_ms = None
    if args.precomputed:
        from torch_points3d_amd.multiscale import MultiScaleTransform
        transform = MultiScaleTransform(model.get_spatial_ops())
        tables = transform(PDData(pos=pos, batch=batch))  # first call: workspaces, lazy initialisation
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            tables = transform(PDData(pos=pos, batch=batch))
        torch.cuda.synchronize()
        pre_ms = (time.perf_counter() - t0) * 1e3 / 5

    def step():
        data = PDData(pos=pos, batch=batch, x=x)
        if tables is not None:
            data.multiscale, data.upsample = tables.multiscale, tables.upsample
        if args.train:
            out = model(data)
            out.x.square().mean().backward()
            for p in model.parameters():
                p.grad = None
        else:
            with torch.no_grad():
                out = model(data)
        return out

    if not (args.graph and args.train):
        # (the graph-replayed training step warms up inside ShardedStep: a .backward() here would leave gradient
        #  accumulators bound to the default stream, which the capture on a side stream must not touch)
        for _ in range(args.warmup):
            out = step()
    torch.cuda.synchronize()
    run = step
    if args.graph and args.train:
        # training step (forward, loss, backward, Adam) replayed from two HIP graphs, as bench.py does for PointNet++
        if not args.precomputed:
            raise SystemExit("--graph needs --precomputed (static shapes, no host reads)")
        from torch_points3d_amd.dp import ShardedStep

        def loss_fn():
            data = PDData(pos=pos, batch=batch, x=x)
            data.multiscale, data.upsample = tables.multiscale, tables.upsample
            return model(data).x.square().mean()
        trainer = ShardedStep(model, lambda ps: torch.optim.Adam(ps, lr=1e-3, capturable=True), loss_fn, use_graph=True,
                              log=lambda m: print("[bench_kpconv] " + m, file=sys.stderr))
        if not trainer.warmup_and_capture(3):
            raise SystemExit("graph capture failed")
        run = trainer.step
    elif args.graph:
        if not args.precomputed:
            raise SystemExit("--graph needs --precomputed (static shapes, no host reads)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            graph_out = step()
        torch.cuda.synchronize()
        eager_out = step()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(graph_out.x, eager_out.x), "graph replay differs from the eager forward"
        run = g.replay
    t0 = time.perf_counter()
    for _ in range(args.iters):
        out = run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.iters
    timer = _lib.KernelTimer()
    _lib.set_timer(timer)
    (trainer.eager_step if (args.graph and args.train) else step)()
    torch.cuda.synchronize()
    _lib.set_timer(None)
    per_entry = {}
    for (name, _), (cnt, tot) in timer.summary().items():
        c, t = per_entry.get(name, (0, 0.0))
        per_entry[name] = (c + cnt, t + tot)
    top = sorted(per_entry.items(), key=lambda kv: -kv[1][1])
    mode = ("train" if args.train else "forward") + ("_precomputed" if args.precomputed else "") + \
           ("_graph" if args.graph else "")
    print(json.dumps({"workload": "kpconv_unet4_%s" % mode, "precompute_ms": None if pre_ms is None else round(pre_ms, 3),
                      "points": pos.shape[0],
                      "clouds": args.clouds, "in_feat": args.in_feat, "ms": round(ms, 3),
                      "points_per_s": round(pos.shape[0] / ms * 1e3),
                      "library_ms": round(sum(v[1] for v in per_entry.values()), 3),
                      "entries": {k: [v[0], round(v[1], 3)] for k, v in top}}))


if __name__ == "__main__":
    main()
