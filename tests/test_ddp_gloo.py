"""N > 1 path on CPU: world_size 2, gloo.  The batch is sharded by cloud (no data-path collective); the only
exchange is DDP's gradient all-reduce.  Kernels are the CPU oracle here (the HIP path needs a GPU), the
sharding / all-reduce logic is the product's (bench.py uses the same wrapper and DDP settings)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG = dict(npoint=[48, 12], radii=[[0.45], [0.9]], nsample=[[8], [8]],
           down_conv_nn=[[[3 + 3, 8, 8, 12]], [[12 + 3, 12, 12, 16]]], innermost=[16 + 3, 16, 24],
           up_conv_nn=[[24 + 16, 16, 16], [16 + 12, 16, 12], [12 + 3, 12, 12, 12]],
           normalize_xyz=[False, False], save_sampling_id=[False, False])


def _make(seed=0):
    sys.path.insert(0, ROOT)
    import bench
    from oracle import tpk_ref
    from torch_points3d_amd.pointnet2 import PointNet2Unet
    torch.manual_seed(seed)
    net = PointNet2Unet(3, output_nc=5, config=CFG, kernels=tpk_ref)
    return bench.SegStep(net).train()


def _inputs():
    g = torch.Generator().manual_seed(42)
    pos = torch.rand(4, 200, 3, generator=g) * 2 - 1
    x = torch.randn(4, 200, 3, generator=g)
    y = torch.randint(0, 5, (4, 200), generator=g)
    return pos, x, y


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = torch.nn.parallel.DistributedDataParallel(_make(), bucket_cap_mb=16, gradient_as_bucket_view=True)
    pos, x, y = _inputs()
    sl = slice(rank * 2, rank * 2 + 2)  # each rank owns whole clouds
    loss = torch.nn.functional.cross_entropy(model(pos[sl], x[sl]), y[sl])
    loss.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(grads) for _ in range(world)]
    dist.all_gather(gathered, grads)
    if rank == 0:
        torch.save({"grads": grads, "same": bool(torch.equal(gathered[0], gathered[1]))}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_average_shard_gradients(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    assert res["same"], "ranks hold different gradients after the all-reduce"
    # expectation: mean over ranks of the gradient each shard produces on its own (BatchNorm stays per rank)
    pos, x, y = _inputs()
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)  # same summation order as the workers
    local = []
    try:
        for r in range(2):
            m = _make()
            sl = slice(r * 2, r * 2 + 2)
            torch.nn.functional.cross_entropy(m(pos[sl], x[sl]), y[sl]).backward()
            local.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]))
    finally:
        torch.set_num_threads(nthreads)
    want = (local[0] + local[1]) / 2
    torch.testing.assert_close(res["grads"], want, rtol=1e-5, atol=1e-6)


def _sharded_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_points3d_amd.dp import ShardedStep
    model = _make()
    pos, x, y = _inputs()
    sl = slice(rank * 2, rank * 2 + 2)
    tr = ShardedStep(model, lambda ps: torch.optim.SGD(ps, lr=0.1),
                     lambda: torch.nn.functional.cross_entropy(model(pos[sl], x[sl]), y[sl]), world_size=world,
                     use_graph=False)
    before = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    tr._forward_backward()
    tr._reduce()
    grads = torch.cat([p.grad.reshape(-1) for p in tr.params]).clone()  # views into the (padded) flat buffer
    assert tr.flat.numel() % 64 == 0 and int(tr.flat.count_nonzero()) == int(grads.count_nonzero())  # padding stays zero
    tr.opt.step()  # ONE flat parameter inside the optimizer; the model's parameters are views of it
    weights = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert torch.allclose(weights, before - 0.1 * grads, rtol=0, atol=1e-7)
    gathered = [torch.zeros_like(weights) for _ in range(world)]
    dist.all_gather(gathered, weights)
    if rank == 0:
        torch.save({"grads": grads, "same_weights": bool(torch.equal(gathered[0], gathered[1]))}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_step_flat_allreduce(tmp_path):
    """bench.py's multi-GPU path (torch_points3d_amd.dp.ShardedStep): one flat gradient all-reduce per step."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_sharded_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    assert res["same_weights"], "ranks diverged after one optimizer step"
    pos, x, y = _inputs()
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    local = []
    try:
        for r in range(2):
            m = _make()
            sl = slice(r * 2, r * 2 + 2)
            torch.nn.functional.cross_entropy(m(pos[sl], x[sl]), y[sl]).backward()
            local.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]))
    finally:
        torch.set_num_threads(nthreads)
    torch.testing.assert_close(res["grads"], (local[0] + local[1]) / 2, rtol=1e-5, atol=1e-6)


def _unequal_seed_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_points3d_amd.dp import ShardedStep
    model = _make(seed=100 + rank)  # ranks build DIFFERENT initial weights ...
    extra = torch.nn.Parameter(torch.ones(3))  # ... and one parameter the loss never reaches
    model.register_parameter("unused_extra", extra)
    with torch.no_grad():
        for b in model.buffers():
            if b.dtype.is_floating_point:
                b.add_(float(rank))  # BatchNorm buffers differ as well
    pos, x, y = _inputs()
    sl = slice(rank * 2, rank * 2 + 2)
    tr = ShardedStep(model, lambda ps: torch.optim.SGD(ps, lr=0.1),
                     lambda: torch.nn.functional.cross_entropy(model(pos[sl], x[sl]), y[sl]), world_size=world,
                     use_graph=False)
    state = torch.cat([p.detach().reshape(-1) for p in model.parameters()] +
                      [b.detach().reshape(-1).float() for b in model.buffers()])
    gathered = [torch.zeros_like(state) for _ in range(world)]
    dist.all_gather(gathered, state)
    tr.step()  # must not raise on the unused parameter; its gradient is zero
    weights = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    after = [torch.zeros_like(weights) for _ in range(world)]
    dist.all_gather(after, weights)
    if rank == 0:
        torch.save({"same_start": bool(torch.equal(gathered[0], gathered[1])),
                    "same_after": bool(torch.equal(after[0], after[1])),
                    "extra_unchanged": bool(torch.equal(model.unused_extra.detach(), torch.ones(3)))}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_step_synchronises_replicas_and_tolerates_unused_parameters(tmp_path):
    """ranks seeded differently must still start from rank 0's parameters and buffers (DDP broadcasts at construction;
    averaging gradients of mismatched replicas trains none of them)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_unequal_seed_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    assert res["same_start"], "replicas differ after ShardedStep construction"
    assert res["same_after"], "replicas diverged after one step"
    assert res["extra_unchanged"]


def _pipelined_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_points3d_amd.dp import PipelinedStep, ShardedStep
    pos, x, y = _inputs()
    sl = slice(rank * 2, rank * 2 + 2)
    ce = torch.nn.functional.cross_entropy
    plain = _make()
    a = ShardedStep(plain, lambda ps: torch.optim.Adam(ps, lr=1e-2), lambda: ce(plain(pos[sl], x[sl]), y[sl]),
                    world_size=world, use_graph=False)
    piped = _make()
    calls = []

    def geometry(slot):
        calls.append(slot)
        return piped.net.precompute_geometry(pos[sl], backward_tables=True)

    b = PipelinedStep(piped, lambda ps: torch.optim.Adam(ps, lr=1e-2), geometry,
                      lambda geo: ce(piped(pos[sl], x[sl], geometry=geo), y[sl]), world_size=world, use_graph=False)
    assert b._side is None  # CPU: the second stream does not exist, the slot logic is what runs
    for _ in range(3):
        a.step()
        b.step()
    wa = torch.cat([p.detach().reshape(-1) for p in plain.parameters()])
    wb = torch.cat([p.detach().reshape(-1) for p in piped.parameters()])
    bufa = torch.cat([t.detach().reshape(-1).float() for t in plain.buffers()])
    bufb = torch.cat([t.detach().reshape(-1).float() for t in piped.buffers()])
    gathered = [torch.zeros_like(wb) for _ in range(world)]
    dist.all_gather(gathered, wb)
    if rank == 0:
        torch.save({"same_across_ranks": bool(torch.equal(gathered[0], gathered[1])), "plain": wa, "piped": wb,
                    "buf_plain": bufa, "buf_piped": bufb, "geometry_calls": torch.tensor(calls)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_step_two_ranks_matches_sharded_step(tmp_path):
    """The stepper bench.py runs for N > 1 (dp.PipelinedStep: geometry of the next batch one step ahead) against the
    plain ShardedStep, three Adam steps on two gloo ranks: same weights, replicas identical, both slots used."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_pipelined_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=True)
    assert res["same_across_ranks"], "ranks diverged under PipelinedStep"
    torch.testing.assert_close(res["piped"], res["plain"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(res["buf_piped"], res["buf_plain"], rtol=1e-6, atol=1e-7)
    # first step fills the current slot and prefetches the other; every later step prefetches one slot
    assert res["geometry_calls"].tolist() == [0, 1, 0, 1]
