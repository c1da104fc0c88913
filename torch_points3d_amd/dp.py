"""Batch-sharded data parallelism for the hot path: one process per GPU, whole clouds per rank, and ONE
gradient all-reduce per step over RCCL (SURVEY.md 8e -- the reference itself has no distributed path).

Every op of the path is per cloud, so ranks never exchange activations.  All trainable parameters share one
flat fp32 gradient buffer (their `.grad` are views into it): a step is

    [forward, loss, backward, pack gradients into flat]  ->  all_reduce(flat) / world  ->  [optimizer step]

The two bracketed phases are pure device work on static tensors, so each is captured once into a HIP graph and
replayed (no per-launch host cost); the collective runs between them as a normal RCCL call on the same stream.
The model has 1.38 M parameters (5.5 MB): a single all-reduce of ~50-100 us per ~12 ms step, so overlapping it
with backward (what DDP's bucketing buys) is not worth giving up graph replay for.
BatchNorm statistics stay per rank (the reference has no SyncBN).  At construction every rank receives rank 0's
parameters and buffers, so replicas that were seeded differently still start identical.
"""
import torch
import torch.distributed as dist


class ShardedStep(object):
    def __init__(self, model, make_optimizer, loss_fn, world_size=1, use_graph=True, log=None, flatten_params=True,
                 reduce_always=False):
        """loss_fn() -> scalar loss of this rank's shard (closes over static input tensors).

        flatten_params: the parameters become views of ONE flat fp32 tensor that is handed to the optimizer as a
        single parameter (with the flat gradient as its .grad): an element-wise optimizer such as Adam then runs as
        ~10 kernels over 1.4 M elements instead of ~150 small ones over 56 tensors -- same arithmetic per element.
        (Per-parameter options such as different weight decays need flatten_params=False.)

        Build the stepper BEFORE the model has seen a `.backward()`: gradient accumulators created on the default
        stream make the autograd engine synchronise with it, which is illegal inside the side-stream capture (the
        capture then crashes at instantiation -- seen with tools/bench_kpconv.py when it warmed up that way)."""
        self.model = model
        self.loss_fn = loss_fn
        try:
            from . import fused as _fused_mod
        except Exception:  # noqa: BLE001 -- the CPU tests drive this class without the HIP library
            _fused_mod = None
        self._fused = _fused_mod
        self.world = world_size
        self.reduce_always = reduce_always  # issue the collective even on one rank (rehearsal of the N > 1 path)
        self.log = log or (lambda msg: None)
        params = [p for p in model.parameters() if p.requires_grad]
        dev = params[0].device
        self.params = params
        # every parameter starts on a 256-byte boundary of the flat buffers (GEMM operands stay 16-byte aligned);
        # the padding elements are zero in both the parameter and the gradient buffer, so they never move
        align = 64
        offsets, total = [], 0
        for p in params:
            offsets.append(total)
            total += (p.numel() + align - 1) // align * align
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self._pads = [torch.zeros((o2 - o1) - p.numel(), dtype=torch.float32, device=dev)
                      for p, o1, o2 in zip(params, offsets, offsets[1:] + [total])]
        self.flat_param = None
        if flatten_params:
            init = torch.zeros(total, dtype=torch.float32, device=dev)
            for p, off in zip(params, offsets):
                init[off:off + p.numel()] = p.detach().reshape(-1).float()
            self.flat_param = torch.nn.Parameter(init)
            self.flat_param.grad = self.flat
        for p, off in zip(params, offsets):
            n = p.numel()
            if flatten_params:
                p.data = self.flat_param.data[off:off + n].view_as(p)  # the model reads the optimizer's tensor
            p.grad = self.flat[off:off + n].view_as(p)  # (per-parameter optimizers read these views)
        if world_size > 1:
            self._sync_replicas()
        self.opt = make_optimizer([self.flat_param] if flatten_params else params)
        self.graph_fb = None
        self.graph_opt = None
        self.graphed = False
        self._want_graph = use_graph and dev.type == "cuda"

    def _sync_replicas(self):
        """Every rank starts from rank 0's parameters and buffers (what DistributedDataParallel does at construction):
        averaging gradients of replicas that were initialised differently would train none of them."""
        with torch.no_grad():
            if self.flat_param is not None:
                dist.broadcast(self.flat_param.data, src=0)
            else:
                for p in self.params:
                    dist.broadcast(p.data, src=0)
            for b in self.model.buffers():
                dist.broadcast(b, src=0)

    # -- phases ------------------------------------------------------------------------------------------
    def _forward_backward(self):
        loss = self.loss_fn()
        # fresh gradient tensors (no per-parameter "+=" kernels), packed into the flat buffer by one concatenation
        grads = torch.autograd.grad(loss, self.params, allow_unused=True)
        pieces = []
        for g, p, pad in zip(grads, self.params, self._pads):
            # a parameter the loss does not reach gets a zero gradient (DDP: find_unused_parameters)
            pieces.append(g.reshape(-1) if g is not None else torch.zeros(p.numel(), dtype=torch.float32, device=p.device))
            if pad.numel():
                pieces.append(pad)
        torch.cat(pieces, out=self.flat)
        return loss

    def _reduce(self):
        if self.world > 1 or self.reduce_always:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(self.world)

    def _eager_step(self):
        self._forward_backward()
        self._reduce()
        self.opt.step()

    # -- capture -----------------------------------------------------------------------------------------
    def warmup_and_capture(self, warmup_steps=3):
        for _ in range(warmup_steps):
            self._eager_step()
        if not self._want_graph:
            return False
        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):  # allocator / lazy-init warm-up on a non-default stream
                    self._eager_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            # thread-local capture mode: RCCL's watchdog thread may poll events while this thread captures
            g1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                self._forward_backward()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                self.opt.step()
            torch.cuda.synchronize()
            self.graph_fb, self.graph_opt, self.graphed = g1, g2, True
            self.log("train step captured into two HIP graphs (forward+backward, optimizer)")
        except Exception as exc:  # stay on the eager path rather than lose the run
            self.log("graph capture unavailable (%s: %s); eager launches" % (type(exc).__name__, exc))
            torch.cuda.synchronize()
            self.graphed = False
        return self.graphed

    def step(self):
        if self.graphed:
            if self._fused is not None:
                self._fused.note_graph_replay()  # eval-mode BatchNorm caches must not outlive a replayed update
            self.graph_fb.replay()
            self._reduce()
            self.graph_opt.replay()
        else:
            self._eager_step()

    def eager_step(self):
        self._eager_step()


class PipelinedStep(ShardedStep):
    """ShardedStep with the geometry of the NEXT batch computed on a second stream while the current batch trains.

    Everything a dense PointNet++ derives from the positions alone -- farthest point sampling, radius searches, 3-NN
    tables (`model.precompute_geometry`) -- is ~1 ms of mostly serial work that occupies a fraction of the chip (FPS runs
    one workgroup per cloud: 32 of 256 CUs); the feature path does not need it before the step starts.  The reference
    has the same split for its partial-dense networks (MultiScaleTransform: geometry precomputed by the data loader,
    core/data_transform/transforms.py:579-654); here it is a second HIP stream instead of CPU workers.

        main stream :  ... | fwd+bwd(i) using G[i%2] | all-reduce | Adam | fwd+bwd(i+1) using G[(i+1)%2] | ...
        side stream :  ... | geometry(i+1) -> G[(i+1)%2]           |      | geometry(i+2) -> G[i%2]       | ...

    geometry_fn(slot) -> geometry of the batch that will train next (written to fresh tensors; the stepper keeps two
    slots alive); loss_fn(geometry) -> scalar loss of the current batch.  Every step still does all of its work -- one
    geometry pass and one training pass -- only their order across the two streams differs."""

    def __init__(self, model, make_optimizer, geometry_fn, loss_fn, **kw):
        self._geometry_fn = geometry_fn
        self._loss_with = loss_fn
        self._slots = [None, None]
        self._cur = 0
        super().__init__(model, make_optimizer, lambda: self._loss_with(self._slots[self._cur]), **kw)
        dev = self.params[0].device
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._geo_done = [None, None]   # events: geometry slot written (recorded on the side stream)
        self._slot_free = [None, None]  # events: training pass that read the slot finished (main stream)
        self.graph_geo = [None, None]
        self.graph_fbs = [None, None]

    # -- eager form (also the warm-up) -------------------------------------------------------------------
    def _eager_step(self):
        nxt = self._cur ^ 1
        if self._slots[self._cur] is None:  # very first step: nothing was prefetched
            self._slots[self._cur] = self._geometry_fn(self._cur)
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream())  # slot `nxt` was last read by the previous step
            with torch.cuda.stream(self._side):
                self._slots[nxt] = self._geometry_fn(nxt)
        else:
            self._slots[nxt] = self._geometry_fn(nxt)
        self._forward_backward()
        self._reduce()
        self.opt.step()
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._cur = nxt

    # -- capture -----------------------------------------------------------------------------------------
    def warmup_and_capture(self, warmup_steps=3):
        for _ in range(max(warmup_steps, 2)):
            self._eager_step()
        if not self._want_graph:
            return False
        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # allocator / lazy-init warm-up on a non-default stream
                for slot in (0, 1):
                    self._slots[slot] = self._geometry_fn(slot)
                    self._cur = slot
                    self._forward_backward()
                    self.opt.step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for slot in (0, 1):  # geometry graphs first: their outputs are the training graphs' static inputs
                g = torch.cuda.CUDAGraph()
                # captured ON the stream they are replayed on: scratch buffers are keyed by (device, stream, purpose)
                # (_lib.workspace), so the geometry graphs and the training graphs, which run concurrently, can never
                # bake in the same buffer -- whatever tags either side uses
                with torch.cuda.graph(g, stream=self._side, capture_error_mode="thread_local"):
                    self._slots[slot] = self._geometry_fn(slot)
                self.graph_geo[slot] = g
            for slot in (0, 1):
                self._cur = slot
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._forward_backward()
                self.graph_fbs[slot] = g
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                self.opt.step()
            self.graph_opt = g2
            torch.cuda.synchronize()
            # prime the pipeline: geometry of the first batch, on the side stream like all later ones
            self._cur = 0
            with torch.cuda.stream(self._side):
                self.graph_geo[0].replay()
                self._geo_done[0] = torch.cuda.Event()
                self._geo_done[0].record(self._side)
            torch.cuda.synchronize()
            self.graphed = True
            self.log("train step captured: 2 geometry graphs (side stream), 2 forward+backward graphs, optimizer graph")
        except Exception as exc:  # stay on the eager path rather than lose the run
            self.log("graph capture unavailable (%s: %s); eager launches" % (type(exc).__name__, exc))
            torch.cuda.synchronize()
            self.graphed = False
        return self.graphed

    def step(self):
        if not self.graphed:
            return self._eager_step()
        cur, nxt = self._cur, self._cur ^ 1
        if self._fused is not None:
            self._fused.note_graph_replay()
        main = torch.cuda.current_stream()
        # geometry of the next batch: may start as soon as the training pass that last read slot `nxt` is done
        if self._slot_free[nxt] is not None:
            self._side.wait_event(self._slot_free[nxt])
        with torch.cuda.stream(self._side):
            self.graph_geo[nxt].replay()
            ev = torch.cuda.Event()
            ev.record(self._side)
            self._geo_done[nxt] = ev
        main.wait_event(self._geo_done[cur])
        self.graph_fbs[cur].replay()
        done = torch.cuda.Event()
        done.record(main)
        self._slot_free[cur] = done
        self._reduce()
        self.graph_opt.replay()
        self._cur = nxt

    def eager_step(self):
        self._eager_step()

    def serial_eager_step(self):
        """Geometry and training pass back to back on the current stream: what bench.py's per-kernel HIP-event pass runs
        (with the second stream active, event brackets on one stream also cover time the other stream's kernels hold
        the chip)."""
        self._slots[self._cur] = self._geometry_fn(self._cur)
        self._forward_backward()
        self._reduce()
        self.opt.step()
