"""Channel-last fused implementation of the grouped-MLP aggregation.

The reference materialises (B, C, npoint, nsample) tensors and runs Conv2d 1x1 -> BatchNorm2d -> LeakyReLU ->
max_pool2d on them (modules/pointnet2/dense.py:36-75, core/common_modules/dense_modules.py:5-29).  Here the same
arithmetic runs on (rows, C) row-major activations: hand-written HIP kernels (csrc/rows.hip) do the gather /
centre / concat, the BatchNorm statistics, the folded affine + LeakyReLU (+ max over nsample) and every
backward pass; only the dense contractions are issued as plain library GEMMs (torch.mm -> rocBLAS/hipBLASLt).
Parameters and buffers are read from the very same nn.Conv2d / nn.BatchNorm2d modules the reference layout
defines, so state_dict keys and checkpoints are unchanged.
"""
import torch
import torch.nn as nn

from . import _lib


USE_ROWS_GEMM = True  # False: forward / input-gradient contractions through torch.mm (library GEMM)


def _cl(x):
    """(B, C, N) tensor -> its (B, N, C) contiguous channel-last form (free when x is a transposed view)."""
    return x.transpose(1, 2).contiguous()


def _pad4(c):
    return (c + 3) & ~3


def _slope_of(act):
    if act is None:
        return 1.0
    if isinstance(act, nn.LeakyReLU):
        return float(act.negative_slope)
    if isinstance(act, nn.ReLU):
        return 0.0
    return None


def layer_parts(block):
    """(conv, bn, slope) of a Conv2D/Conv1D Seq block if the fused kernels can run it, else None."""
    mods = list(block.children())
    if not mods or not isinstance(mods[0], (nn.Conv2d, nn.Conv1d)):
        return None
    conv = mods[0]
    if conv.bias is not None or any(k != 1 for k in conv.kernel_size):
        return None
    bn, act = None, None
    for m in mods[1:]:
        if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)) and bn is None and act is None:
            bn = m
        elif act is None:
            act = m
        else:
            return None
    if bn is None or not bn.affine or not bn.track_running_stats or bn.momentum is None:
        return None
    slope = _slope_of(act)
    if slope is None:
        return None
    return conv, bn, slope


def mlp_parts(mlp):
    parts = [layer_parts(b) for b in mlp.children()]
    return None if (not parts or any(p is None for p in parts)) else parts


FWD_X3 = True  # forward contractions of the hidden layers (more than 64 output columns) on the bf16 matrix pipe as well
               # (csrc/gemm_rows_x3.hip: exact three-term split, six term pairs, fp32 accumulation)
WGRAD_X3_ACT = True  # ... and its loader waves form the activated A operand from the previous layer's pre-BatchNorm output,
                     # so that the forward kernels write no activated side output for those layers (one (M, K) store less)
WGRAD_X3 = 6  # weight-gradient contractions of the large layers on the bf16 matrix pipe, every fp32 value split exactly into
              # three bf16 terms (csrc/gemm_tn_x3.hip): 6 = the six term pairs of weight >= 2^-15 (what is dropped is below
              # 2^-21 of a product), 9 = all nine (every product exact), 0 = the fp32 MFMA kernel for every shape.
              # Measured against float64 on the layers of the BASELINE step: 2.1-3.3e-7 of the scale with 6 or 9 terms
              # (hi * hi products in an accumulator of their own), 4.2-5.4e-7 for the fp32 MFMA kernel; 524288 x 128 x 128:
              # 133 us (6), 141 us (9), 185 us (fp32 MFMA) -- the matrix pipe's load lowers the shader clock (2.2 -> 1.8 GHz
              # in the counters), so the nine-term form costs more than its extra MFMAs


_ident = {}


def _identity_stats(K, dev):
    """(zeros, ones) rows of K floats: the statistics under which a BatchNorm + LeakyReLU(1) prologue is the identity"""
    key = (dev.index, K)
    hit = _ident.get(key)
    if hit is None:
        hit = _ident[key] = (torch.zeros(K, dtype=torch.float32, device=dev), torch.ones(K, dtype=torch.float32, device=dev))
    return hit


def _chain_wgrad_red(dY, l, Ys, stats, slopes, dA_prev, training, reverse=0):
    """weight gradient of layer l >= 1 of a fused chain AND the BatchNorm-backward reductions of layer l - 1 from dA_prev,
    the gradient of its activated output: both need Y_{l-1}, which streams through the contraction's loader waves once
    (tp3d_gemm_tn_x3_act_red_f32).  Returns (dW, red) with red (4, C_{l-1}) = dbeta, dgamma, c1, c2, or None where the
    kernel does not serve the shape."""
    dev = dY.device
    M, N = dY.shape
    K = Ys[l - 1].shape[1]
    h = _lib.load()
    chunks = h.tp3d_gemm_tn_x3_red_chunks(M, N, K) if WGRAD_X3 == 6 else 0
    if not chunks:
        return None
    ps = stats[l - 1]
    out = torch.empty((N, K), dtype=torch.float32, device=dev)
    red = torch.empty((4, K), dtype=torch.float32, device=dev)
    ws = _lib.gemm_tn_workspace(M, N, K, dev, x3=True)
    rws = _lib.workspace("gemm_tn_red", 4 * 2 * chunks * K, dev)
    with _lib.on_device(dev):
        _lib.call("tp3d_gemm_tn_x3_act_red_f32", _lib.ptr(dY), _lib.ptr(Ys[l - 1]), _lib.ptr(ps[0]), _lib.ptr(ps[2]), _lib.ptr(ps[3]),
                  _lib.ptr(ps[1]), float(slopes[l - 1]), _lib.ptr(dA_prev), int(training), M, N, K, 6, _lib.ptr(out), _lib.ptr(ws),
                  _lib.ptr(red), _lib.ptr(rws), int(reverse), _lib.stream_ptr(dev))
    return out, red


def _chain_wgrad(dY, l, A0, acts, Ys, stats, slopes, reverse=0):
    """weight gradient of layer l of a fused chain: dY^T @ (activated input of the layer)"""
    if l == 0:
        return gemm_tn(dY, A0, reverse=reverse)
    if acts[l - 1] is not None:
        return gemm_tn(dY, acts[l - 1], reverse=reverse)
    ps = stats[l - 1]  # the forward pass kept no activated rows: the contraction's loader waves form them from Y_{l-1}
    return gemm_tn(dY, Ys[l - 1], act=(ps[0], ps[2], ps[3], slopes[l - 1]), reverse=reverse)


def gemm_tn(dY, A, x3=None, act=None, reverse=0):
    """dY (M,N), A (M,K) -> dY^T @ A (N,K), split over the rows, partial tiles summed in fixed order (reproducible):
    csrc/gemm_tn_x3.hip for the shapes it serves (x3 terms, default WGRAD_X3), else the fp32 MFMA kernel csrc/gemm_tn.hip.
    reverse: the bf16-pipe kernel walks the row blocks last to first (include/tp3d_hip.h, `reverse`)."""
    dev = dY.device
    dY, A = dY.contiguous(), A.contiguous()
    M, N = dY.shape
    K = A.shape[1]
    out = torch.empty((N, K), dtype=torch.float32, device=dev)
    terms = WGRAD_X3 if x3 is None else x3
    use_x3 = bool(terms) and M > 0 and bool(_lib.load().tp3d_gemm_tn_x3_serves(M, N, K))
    if act is not None:
        # A = LeakyReLU((Yp - mean) * scale + beta) formed by the contraction's loader waves (act = mean, scale, beta, slope)
        if not use_x3:
            raise ValueError("gemm_tn(act=): only the bf16-pipe contraction forms its A operand (shape %s x %s x %s)" % (M, N, K))
        ws = _lib.gemm_tn_workspace(M, N, K, dev, x3=True)
        with _lib.on_device(dev):
            _lib.call("tp3d_gemm_tn_x3_act_f32", _lib.ptr(dY), _lib.ptr(A), _lib.ptr(act[0]), _lib.ptr(act[1]), _lib.ptr(act[2]),
                      float(act[3]), M, N, K, int(terms), _lib.ptr(out), _lib.ptr(ws), int(reverse), _lib.stream_ptr(dev))
        return out
    ws = _lib.gemm_tn_workspace(M, N, K, dev, x3=use_x3)
    with _lib.on_device(dev):
        if use_x3:
            _lib.call("tp3d_gemm_tn_x3_f32", _lib.ptr(dY), _lib.ptr(A), M, N, K, int(terms), _lib.ptr(out), _lib.ptr(ws), int(reverse),
                      _lib.stream_ptr(dev))
        else:
            _lib.call("tp3d_gemm_tn_f32", _lib.ptr(dY), _lib.ptr(A), M, N, K, _lib.ptr(out), _lib.ptr(ws),
                      _lib.stream_ptr(dev))
    return out


SKINNY_MAX = 32       # tp3d_gemm_skinny_f32: both channel counts at most this
SKINNY_MIN_ROWS = 32768


def gemm_skinny(A, W):
    """A (M,K) @ W (N,K)^T -> (M,N) for K, N <= 32 and many rows: one row per lane (csrc/gemm_skinny.hip)."""
    dev = A.device
    A, W = A.contiguous(), W.contiguous()
    M, K = A.shape
    N = W.shape[0]
    Y = torch.empty((M, N), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        _lib.call("tp3d_gemm_skinny_f32", _lib.ptr(A), _lib.ptr(W), M, N, K, K, _lib.ptr(Y), _lib.stream_ptr(dev))
    return Y


def _is_skinny(M, a, b):
    return M >= SKINNY_MIN_ROWS and a <= SKINNY_MAX and b <= SKINNY_MAX


def gemm_rows(A, Bt, want_stats=False):
    """A (M,K) @ Bt (N,K)^T -> (M,N) with the fp32 MFMA rows kernel (csrc/gemm_rows.hip); optionally also the shifted
    partial column sums for BatchNorm (then returned as `part`).  K must be a multiple of 4.  Without statistics a
    long contraction with few output tiles is split over K-ranges (two-level summation, more workgroups)."""
    dev = A.device
    M, K = A.shape
    N = Bt.shape[0]
    Bm = Bt.contiguous()
    C = torch.empty((M, N), dtype=torch.float32, device=dev)
    part, slabs = None, None
    h = _lib.load()
    if want_stats:
        part = _lib.workspace("gemm_rows_stats", 4 * h.tp3d_gemm_rows_stat_floats(M, N), dev)
    else:
        n = h.tp3d_gemm_rows_workspace_floats(M, N, K)
        if n:
            slabs = _lib.workspace("gemm_rows_slabs", 4 * n, dev)
    with _lib.on_device(dev):
        _lib.call("tp3d_gemm_rows_f32", _lib.ptr(A), _lib.ptr(Bm), M, N, K, _lib.ptr(C), _lib.ptr(part), _lib.ptr(slabs),
                  _lib.stream_ptr(dev))
    return C, part


def _finalize_stats(part, M, C, gamma, beta, bn, dev, st, chunks=None):
    """(4, C) = mean, invstd, scale, beta from the shifted partial sums a rows GEMM left in `part` (training mode)."""
    stats = torch.empty((4, C), dtype=torch.float32, device=dev)
    if chunks is None:
        chunks = _lib.load().tp3d_gemm_rows_stat_chunks(M, C)
    _lib.call("tp3d_bn_finalize_f32", _lib.ptr(part), chunks, M, C, float(bn.eps),
              float(bn.momentum), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var),
              _lib.ptr(bn.num_batches_tracked), _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), st)
    _note_training_pass(bn)
    return stats


def _note_training_pass(bn):
    """The kernels update running_mean / running_var / num_batches_tracked in place without touching the tensors'
    version counters: eval-mode statistics cached on the module (_bn_stats) are keyed on this count as well."""
    bn._tp3d_train_passes = getattr(bn, "_tp3d_train_passes", 0) + 1


_replay_epoch = 0
_outer_grad = True  # grad mode at the call site of the autograd Functions below (inside Function.forward it is always off)


def _apply(fn, *args):
    """fn.apply(*args) with the caller's grad mode recorded: under torch.no_grad() the parameters still report
    requires_grad through ctx.needs_input_grad, and the forward passes would keep side outputs for a backward that
    cannot come"""
    global _outer_grad
    _outer_grad = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _outer_grad = True


def note_graph_replay():
    """A captured training step was replayed: parameters and running statistics of every module may have changed
    without any Python running (no version counter moves, _note_training_pass does not fire, and with flattened
    parameters the optimizer only touches the flat tensor).  dp.ShardedStep / PipelinedStep call this on every replay;
    the count is part of the eval-statistics cache key, so a validation pass after replayed training recomputes them."""
    global _replay_epoch
    _replay_epoch += 1


ROWS_GEMM_MIN_COLS = 32   # narrower outputs (class scores, edge MLPs) stay on the library / skinny kernels
ROWS_GEMM_WITHOUT_STATS = True  # the rows kernel also where it has no BatchNorm statistics to fold in.  False: library GEMM
                                # there -- the choice for inference replayed from a HIP graph (KPConv unet_4 forward 1.46 ->
                                # 1.10 ms); launched eagerly the library's host side makes the forward slower (1.96 -> 2.24 ms).
                                # One setting for both, so that a captured step stays bit-identical to its eager twin
ROWS_GEMM_EPILOGUE = True  # eval-mode layers without a gradient request: BatchNorm + activation in the rows GEMM's epilogue
ROWS_GEMM_NARROW = True   # widths served by the 128 x 64 tiles (N % 128 in 1..64) on the rows kernel (else library GEMM)
CHAIN_MIN_ROWS = 32768    # the fused layer chain serves the large row matrices (grouped / per-point activations)


def _rows_gemm_serves(cout):
    if not USE_ROWS_GEMM or cout < ROWS_GEMM_MIN_COLS:
        return False
    return ROWS_GEMM_NARROW or not (0 < cout % 128 <= 64)


def _long_k(M, N, K):
    """few output tiles and a long contraction: served by the K-split launch (no fused statistics)"""
    return _lib.load().tp3d_gemm_rows_workspace_floats(M, N, K) > 0


def _bn_stats(Y, M, C, gamma, beta, bn, training, dev, st, bias=None):
    """(4, C) = mean, invstd, scale = gamma * invstd, beta of BatchNorm over the rows of Y.  In eval mode they depend only on the
    module's parameters and running statistics, so they are computed once and reused until any of those changes."""
    key = None
    if not training:
        key = tuple((t.data_ptr(), t._version) for t in (gamma, beta, bn.running_mean, bn.running_var)
                    + ((bias,) if bias is not None else ())) + (getattr(bn, "_tp3d_train_passes", 0), _replay_epoch)
        hit = getattr(bn, "_tp3d_eval_stats", None)
        if hit is not None and hit[0] == key:
            return hit[1]
    stats = torch.empty((4, C), dtype=torch.float32, device=dev)
    ws = _lib.bn_workspace(M, C, dev)
    _lib.call("tp3d_bn_stats_f32", _lib.ptr(Y), M, C, float(bn.eps), float(bn.momentum), _lib.ptr(gamma), _lib.ptr(beta),
              _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var), _lib.ptr(bn.num_batches_tracked) if training else None,
              int(training), _lib.ptr(stats[0]), _lib.ptr(stats[1]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), _lib.ptr(ws), st)
    if training:
        _note_training_pass(bn)
    if key is not None:
        if bias is not None:
            stats[0].sub_(bias.detach())  # (Y + b - running_mean) * scale + beta == (Y - (running_mean - b)) * scale + beta
        bn._tp3d_eval_stats = (key, stats)
    return stats


class _LinearBNAct(torch.autograd.Function):
    """out = LeakyReLU(BatchNorm(A @ W^T)) on rows; with pool_ns > 0 also the max over groups of pool_ns rows."""

    @staticmethod
    def forward(ctx, A, weight, gamma, beta, bn, slope, pool_ns, bias=None):
        # bias: the Linear's bias (reference MLP default, core/common_modules/base_modules.py:29-43).  The GEMM runs
        # without it: under batch statistics it cancels in the normalised output (it only shifts the running mean);
        # with running statistics it folds into the affine shift.
        dev = A.device
        A = A.contiguous()
        M, Kp = A.shape  # Kp >= Cin: producers pad rows with zero columns to a multiple of 4 floats
        Cout = weight.shape[0]
        W2 = weight.reshape(Cout, -1)
        Cin = W2.shape[1]
        if Kp != Cin:
            W2 = torch.nn.functional.pad(W2, (0, Kp - Cin))
        training = bn.training
        st = _lib.stream_ptr(dev)
        # the dense contraction on the fp32 MFMA rows kernel (128- or 64-column tiles); BatchNorm statistics come out of
        # its epilogue, except for the long contractions with few output tiles (the 4096-row global / decoder layers),
        # which run as a K-split launch followed by the separate statistics pass over their small output
        if (not training and _is_skinny(M, Kp, Cout) and not (_outer_grad and any(ctx.needs_input_grad))):
            # eval-mode edge MLP layer, no gradient wanted: Linear + BatchNorm (running statistics) + activation in ONE
            # pass over the rows (tp3d_gemm_skinny_bnact_f32) instead of GEMM, statistics lookup and affine pass
            with _lib.on_device(dev):
                stats = _bn_stats(A, M, Cout, gamma, beta, bn, False, dev, st, bias)
                rows_out = torch.empty((M, Cout), dtype=torch.float32, device=dev)
                _lib.call("tp3d_gemm_skinny_bnact_f32", _lib.ptr(A), _lib.ptr(W2.contiguous()), M, Cout, Kp, Kp,
                          _lib.ptr(stats[0]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), slope, _lib.ptr(rows_out), st)
            if pool_ns:
                rows_out = rows_out.view(M // pool_ns, pool_ns, Cout).max(1)[0]
            return rows_out
        own_gemm = _rows_gemm_serves(Cout) and Kp % 4 == 0
        # few output tiles and a contraction of 512..1023 channels: the rows kernel has nothing to hide its K walk
        # behind and a K-split would move more slab bytes than it saves -- library GEMM + separate statistics pass
        # (93 -> 27 + 12 us on the 4096-row global layer); longer contractions go K-split for their summation order
        if own_gemm and ((M + 127) // 128) * ((Cout + 127) // 128) < 128 and 512 <= Kp < 1024:
            own_gemm = False
        want_stats = training and not _long_k(M, Cout, Kp)
        if (own_gemm and ROWS_GEMM_EPILOGUE and not training and not pool_ns and ROWS_GEMM_WITHOUT_STATS
                and not (_outer_grad and any(ctx.needs_input_grad))):
            # inference: BatchNorm (running statistics) + activation applied to the accumulators -- one launch per layer
            with _lib.on_device(dev):
                stats = _bn_stats(A, M, Cout, gamma, beta, bn, False, dev, st, bias)
                out = torch.empty((M, Cout), dtype=torch.float32, device=dev)
                n = _lib.load().tp3d_gemm_rows_workspace_floats(M, Cout, Kp)
                slabs = _lib.workspace("gemm_rows_slabs", 4 * n, dev) if n else None
                _lib.call("tp3d_gemm_rows_epi_f32", _lib.ptr(A), _lib.ptr(W2.contiguous()), M, Cout, Kp, _lib.ptr(stats[0]),
                          _lib.ptr(stats[2]), _lib.ptr(stats[3]), slope, _lib.ptr(out), _lib.ptr(slabs), st)
            return out
        if own_gemm and not want_stats and not ROWS_GEMM_WITHOUT_STATS:
            # nothing to fold into the epilogue (eval mode; K-split shapes): the library GEMM's KERNEL is faster on every
            # such shape of the KPConv / PointNet++ networks (tools/probes/small_gemm.py: 65536 x 64 x 256 24 vs 40 us,
            # 27 x 2048 x 1024 18 vs 63 us) but its host side costs more than this library's launch: KPConv unet_4 forward
            # replayed from a HIP graph 1.46 -> 1.10 ms with it, launched eagerly 1.96 -> 2.24 ms
            own_gemm = False
        if own_gemm:
            Y, part = gemm_rows(A, W2, want_stats=want_stats)
        elif _is_skinny(M, Kp, Cout):
            Y = gemm_skinny(A, W2)  # edge-wise MLPs of a few channels: a pure stream, one row per lane
            part = None
        else:
            Y = torch.mm(A, W2.t())  # plain library GEMM (a handful of output columns, e.g. the 10-class head)
            part = None
        with _lib.on_device(dev):
            if part is not None:
                stats = _finalize_stats(part, M, Cout, gamma, beta, bn, dev, st)  # mean, invstd, scale, beta
            else:
                stats = _bn_stats(Y, M, Cout, gamma, beta, bn, training, dev, st, bias)
            if pool_ns:
                G = M // pool_ns
                out = torch.empty((G, Cout), dtype=torch.float32, device=dev)
                arg = torch.empty((G, Cout), dtype=torch.int32, device=dev)
                _lib.call("tp3d_bn_act_maxpool_f32", _lib.ptr(Y), _lib.ptr(stats[0]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), slope, G,
                          pool_ns, Cout, _lib.ptr(out), _lib.ptr(arg), st)
            else:
                arg = None
                out = torch.empty((M, Cout), dtype=torch.float32, device=dev)
                _lib.call("tp3d_bn_act_f32", _lib.ptr(Y), _lib.ptr(stats[0]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), slope, M, Cout,
                          _lib.ptr(out), st)
        if training:  # (num_batches_tracked was advanced by the statistics kernel)
            if bias is not None:
                bn.running_mean.add_(bias.detach(), alpha=float(bn.momentum))  # the kernels saw the mean without it
        ctx.save_for_backward(A, W2, Y, stats, arg)
        ctx.cfg = (slope, pool_ns, training, tuple(weight.shape), Cin, bias is not None)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        A, W2, Y, stats, arg = ctx.saved_tensors
        slope, pool_ns, training, wshape, Cin, has_bias = ctx.cfg
        dev = grad_out.device
        grad_out = grad_out.contiguous()
        M, Cout = Y.shape
        dY = torch.empty_like(Y)
        dgb = torch.empty((2, Cout), dtype=torch.float32, device=dev)  # dbeta, dgamma
        ws = _lib.bn_workspace(M, Cout, dev)
        with _lib.on_device(dev):
            _lib.call("tp3d_bn_act_bwd_f32", _lib.ptr(grad_out), _lib.ptr(arg), _lib.ptr(Y), _lib.ptr(stats[2]),
                      _lib.ptr(stats[3]), _lib.ptr(stats[0]), _lib.ptr(stats[1]), slope, M, max(pool_ns, 1), Cout,
                      int(training), _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), _lib.ptr(dY), _lib.ptr(ws),
                      _lib.stream_ptr(dev))
        dW = None
        if ctx.needs_input_grad[1]:
            dW = gemm_tn(dY, A)[:, :Cin].reshape(wshape)
        dA = None
        if ctx.needs_input_grad[0]:
            if _is_skinny(M, W2.shape[1], Cout):
                dA = gemm_skinny(dY, W2.t())
            else:
                # (the library GEMM: the rows kernel was measured 0.22 ms/step slower on these input-gradient shapes, and
                #  issuing the weight gradients on a second stream 0.3 ms slower -- round 2)
                dA = torch.mm(dY, W2)
        dbias = None
        if has_bias and ctx.needs_input_grad[7]:
            # batch statistics remove the bias from the output (gradient exactly zero); running statistics do not
            dbias = torch.zeros(Y.shape[1], dtype=torch.float32, device=dev) if training else dY.sum(0)
        return dA, dW, dgb[1], dgb[0], None, None, None, dbias



class _MLPChain(torch.autograd.Function):
    """A whole shared MLP -- [1x1 conv -> BatchNorm -> LeakyReLU] x L (+ max over groups of pool_ns rows) -- on a large
    row matrix, with every BatchNorm / activation pass folded into the GEMM that consumes its result
    (core/common_modules/dense_modules.py:25-29 forward, autograd backward):

      forward   Y_0 = A_0 W_0^T;  Y_l = act(BN(Y_{l-1})) W_l^T  -- the activated tensor is formed by the GEMM's loader waves
                from Y_{l-1} (tp3d_gemm_rows_bnact_x3_f32 / _sp_f32), never written; BatchNorm statistics come out of each
                GEMM's epilogue
      backward  per layer the reductions (dbeta, dgamma -- a pass of their own, or riding on the weight-gradient kernel of
                the layer above, tp3d_gemm_tn_x3_act_red_f32), then
                dA_{l-1} = dY_l W_l                  dY_l formed by the loader waves, written once as a side output
                                                     (tp3d_gemm_rows_bnbwd_sp_f32)
                dW_l     = dY_l^T act(BN(Y_{l-1}))   the activated operand formed by the loader waves (tp3d_gemm_tn_x3_act_f32)
                with dY_l = BatchNorm+activation backward of (dA_l, Y_l)

    Only the pre-BatchNorm outputs Y_l are kept for the backward pass (plus the activated rows of the layers whose weight
    gradient the bf16-pipe kernel does not serve)."""

    @staticmethod
    def forward(ctx, A0, pool_ns, layers, grad_cols, *params):
        # layers: [(bn module, slope)], params: [weight_0, gamma_0, beta_0, weight_1, ...]; grad_cols: None or (first, count) --
        # the only columns of A0 whose gradient the producer of A0 reads (grouped rows: the feature columns)
        dev = A0.device
        A0 = A0.contiguous()
        M = A0.shape[0]
        st = _lib.stream_ptr(dev)
        L = len(layers)
        Ys, stats, W2s, cins, acts = [], [], [], [], []
        training = layers[0][0].training
        # under torch.no_grad() the parameters still report requires_grad: no side outputs for a backward that cannot come
        keep_acts = _outer_grad and any(ctx.needs_input_grad)
        h = _lib.load()
        prev_rev = 0  # the producer of A0 wrote front to back
        with _lib.on_device(dev):
            for l, (bn, slope) in enumerate(layers):
                weight, gamma, beta = params[3 * l], params[3 * l + 1], params[3 * l + 2]
                Cout = weight.shape[0]
                W2 = weight.reshape(Cout, -1)
                Kp = A0.shape[1] if l == 0 else Ys[-1].shape[1]
                cins.append(W2.shape[1])
                if W2.shape[1] != Kp:
                    W2 = torch.nn.functional.pad(W2, (0, Kp - W2.shape[1]))
                W2 = W2.contiguous()
                Y = torch.empty((M, Cout), dtype=torch.float32, device=dev)
                # the activated rows are the A operand of this layer's weight gradient; where the bf16-pipe contraction serves
                # that shape its loader waves form them again from Y_{l-1} (tp3d_gemm_tn_x3_act_f32) and nothing is kept
                keep_act = keep_acts and not (WGRAD_X3 and WGRAD_X3_ACT and l > 0 and ctx.needs_input_grad[4 + 3 * l]
                                              and h.tp3d_gemm_tn_x3_serves(M, Cout, Kp))
                sp_chunks = h.tp3d_gemm_rows_sp_chunks(M, Cout, Kp, int(keep_act)) if l > 0 else 0
                sp_entry = "tp3d_gemm_rows_bnact_sp_f32"
                if sp_chunks and FWD_X3 and h.tp3d_gemm_rows_x3_chunks(M, Cout, Kp, int(keep_act)):
                    sp_chunks = h.tp3d_gemm_rows_x3_chunks(M, Cout, Kp, int(keep_act))
                    sp_entry = "tp3d_gemm_rows_bnact_x3_f32"  # the same contraction as bf16 term pairs on the matrix pipe
                # alternate the direction the row blocks are walked in, layer by layer: a layer starts where the previous one
                # (or the producer of A0, front to back) ended, on the rows the memory-side cache still holds
                rev = int(ROW_ORDER_ALTERNATE and not prev_rev)  # (used by the kernels that take a direction; the others walk
                prev_rev = rev                                   # front to back -- corrected below)
                chunks = None
                x3_first = h.tp3d_gemm_rows_x3_chunks(M, Cout, Kp, 0) if (l == 0 and FWD_X3) else 0
                if x3_first:
                    # the first layer on the bf16 pipe as well: the same kernel with the identity as its prologue
                    # ((y - 0) * 1 + 0, slope 1) -- 524288 x 128 x 132: 229 us on the fp32 rows kernel
                    ident = _identity_stats(Kp, dev)
                    part = _lib.workspace("gemm_rows_stats", 16 * x3_first * Cout, dev) if training else None
                    _lib.call("tp3d_gemm_rows_bnact_x3_f32", _lib.ptr(A0), _lib.ptr(ident[0]), _lib.ptr(ident[1]), _lib.ptr(ident[0]),
                              1.0, _lib.ptr(W2), M, Cout, Kp, _lib.ptr(Y), _lib.ptr(part), None, rev, st)
                    chunks = x3_first
                elif l == 0 and FWD_NARROW and h.tp3d_gemm_tn_bn_narrow_serves(M, Cout, Kp):
                    # a handful of input channels (grouped rows: relative position + features): the streaming kernel
                    chunks = h.tp3d_gemm_rows_narrow_chunks(M)
                    part = _lib.workspace("gemm_rows_stats", 16 * chunks * Cout, dev) if training else None
                    _lib.call("tp3d_gemm_rows_narrow_f32", _lib.ptr(A0), _lib.ptr(W2), M, Cout, Kp, _lib.ptr(Y), _lib.ptr(part), rev, st)
                elif l == 0:
                    part = _lib.workspace("gemm_rows_stats", 4 * h.tp3d_gemm_rows_stat_floats(M, Cout), dev) if training else None
                    _lib.call("tp3d_gemm_rows_f32", _lib.ptr(A0), _lib.ptr(W2), M, Cout, Kp, _lib.ptr(Y), _lib.ptr(part), None, st)
                    prev_rev = 0
                elif sp_chunks:
                    # the previous layer's BatchNorm + activation in the loader waves of the split-role kernel; with a
                    # backward pass to come, the activated rows leave as a side output of the same kernel
                    ps = stats[-1]
                    part = _lib.workspace("gemm_rows_stats", 16 * sp_chunks * Cout, dev) if training else None
                    act = torch.empty((M, Kp), dtype=torch.float32, device=dev) if keep_act else None
                    _lib.call(sp_entry, _lib.ptr(Ys[-1]), _lib.ptr(ps[0]), _lib.ptr(ps[2]), _lib.ptr(ps[3]),
                              layers[l - 1][1], _lib.ptr(W2), M, Cout, Kp, _lib.ptr(Y), _lib.ptr(part), _lib.ptr(act), rev, st)
                    chunks = sp_chunks
                    acts.append(act)
                else:
                    # a width the split-role kernel does not serve: the separate pass, then the plain rows GEMM
                    ps = stats[-1]
                    act = torch.empty((M, Kp), dtype=torch.float32, device=dev)
                    _lib.call("tp3d_bn_act_f32", _lib.ptr(Ys[-1]), _lib.ptr(ps[0]), _lib.ptr(ps[2]), _lib.ptr(ps[3]), layers[l - 1][1], M,
                              Kp, _lib.ptr(act), st)
                    part = _lib.workspace("gemm_rows_stats", 4 * h.tp3d_gemm_rows_stat_floats(M, Cout), dev) if training else None
                    _lib.call("tp3d_gemm_rows_f32", _lib.ptr(act), _lib.ptr(W2), M, Cout, Kp, _lib.ptr(Y), _lib.ptr(part), None, st)
                    acts.append(act if keep_act else None)
                    prev_rev = 0
                if training:
                    stats.append(_finalize_stats(part, M, Cout, gamma, beta, bn, dev, st, chunks))
                else:
                    stats.append(_bn_stats(Y, M, Cout, gamma, beta, bn, False, dev, st))
                Ys.append(Y)
                W2s.append(W2)
            Y, ls, slope = Ys[-1], stats[-1], layers[-1][1]
            C = Y.shape[1]
            if pool_ns:
                G = M // pool_ns
                out = torch.empty((G, C), dtype=torch.float32, device=dev)
                arg = torch.empty((G, C), dtype=torch.int32, device=dev)
                _lib.call("tp3d_bn_act_maxpool_f32", _lib.ptr(Y), _lib.ptr(ls[0]), _lib.ptr(ls[2]), _lib.ptr(ls[3]), slope, G,
                          pool_ns, C, _lib.ptr(out), _lib.ptr(arg), st)
            else:
                arg = None
                out = torch.empty((M, C), dtype=torch.float32, device=dev)
                _lib.call("tp3d_bn_act_f32", _lib.ptr(Y), _lib.ptr(ls[0]), _lib.ptr(ls[2]), _lib.ptr(ls[3]), slope, M, C,
                          _lib.ptr(out), st)
        layerwise = keep_acts
        ctx.save_for_backward(A0, arg, *Ys, *stats, *W2s, *(acts if layerwise else []))
        ctx.cfg = (L, pool_ns, training, [s_ for _, s_ in layers], [tuple(params[3 * l].shape) for l in range(L)], cins)
        ctx.layerwise = layerwise
        ctx.grad_cols = grad_cols
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L, pool_ns, training, slopes, wshapes, cins = ctx.cfg
        saved = ctx.saved_tensors
        A0, arg = saved[0], saved[1]
        Ys, stats, W2s = saved[2:2 + L], saved[2 + L:2 + 2 * L], saved[2 + 2 * L:2 + 3 * L]
        dev = grad_out.device
        st = _lib.stream_ptr(dev)
        M = A0.shape[0]
        dcur = grad_out.contiguous()
        grads = [None] * (3 * L)
        dA0 = None
        if not ctx.layerwise:  # (the forward pass saw no gradient request: it kept no operands for this)
            raise RuntimeError("_MLPChain.backward: the forward pass ran without a gradient request (call through fused._apply)")
        # the activated rows were kept (side outputs of the forward kernels): the layer-wise backward passes
        acts = saved[2 + 3 * L:2 + 3 * L + (L - 1)]
        h = _lib.load()

        def route(l):
            """how layer l's backward runs: (kind, columns of the input gradient that are contracted)"""
            C_, Kp_ = W2s[l].shape
            pooled_ = bool(pool_ns) and l == L - 1
            want_prev_ = l > 0 or ctx.needs_input_grad[0]
            # the grouped rows' producer reads the gradient of the feature columns only: contract just those
            cols_ = ctx.grad_cols if (l == 0 and ctx.grad_cols is not None and ctx.grad_cols[1] >= ROWS_GEMM_MIN_COLS) else None
            ncol_ = cols_[1] if cols_ else Kp_
            pow2 = pooled_ and pool_ns >= 64 and (pool_ns & (pool_ns - 1)) == 0
            if (CHAIN_BWD_LOADER and (not pooled_ or (CHAIN_BWD_POOLED and pow2)) and want_prev_
                    and h.tp3d_gemm_rows_bnbwd_sp_serves(M, ncol_, C_)):
                return "loader", cols_
            if (WGRAD_NARROW and not pooled_ and not want_prev_ and ctx.needs_input_grad[4 + 3 * l] and l == 0
                    and h.tp3d_gemm_tn_bn_narrow_serves(M, C_, Kp_)):
                return "narrow", cols_
            return "passes", cols_

        # consecutive big kernels walk the rows in opposite directions: each starts on the rows its predecessor touched
        # last, which the memory-side cache (256 MB against 268 MB per activation matrix) still holds
        turn = [True]  # the producer of grad_out wrote front to back

        def direction():
            rev_ = int(ROW_ORDER_ALTERNATE and turn[0])
            turn[0] = not turn[0]
            return rev_

        def reduce_pass(l, dA_l):
            """dbeta, dgamma, c1, c2 of layer l from the gradient of its activated (or pooled) output"""
            C_ = W2s[l].shape[0]
            pooled_ = bool(pool_ns) and l == L - 1
            a_ptr_, ns_ = (_lib.ptr(arg), pool_ns) if pooled_ else (None, 1)
            red_ = torch.empty((4, C_), dtype=torch.float32, device=dev)
            ls_ = stats[l]
            _lib.call("tp3d_bn_bwd_reduce_f32", _lib.ptr(dA_l), a_ptr_, _lib.ptr(Ys[l]), _lib.ptr(ls_[2]), _lib.ptr(ls_[3]),
                      _lib.ptr(ls_[0]), _lib.ptr(ls_[1]), slopes[l], M, ns_, C_, int(training), _lib.ptr(red_[0]),
                      _lib.ptr(red_[1]), _lib.ptr(red_[2]), _lib.ptr(red_[3]), _lib.ptr(_lib.bn_workspace(M, C_, dev)), direction(), st)
            return red_

        red_next = None  # reductions of the next layer down, when the weight-gradient kernel above it produced them
        with _lib.on_device(dev):
            for l in range(L - 1, -1, -1):
                Y, ls, W2, slope = Ys[l], stats[l], W2s[l], slopes[l]
                C, Kp = W2.shape
                pooled = bool(pool_ns) and l == L - 1
                kind, cols = route(l)
                ncol = cols[1] if cols else Kp
                red_have, red_next = red_next, None
                if kind == "loader":
                    # reduction pass, then the input-gradient GEMM whose loader waves form dY (side output for dW)
                    dY = torch.empty_like(Y)
                    a_ptr, ns = (_lib.ptr(arg), pool_ns) if pooled else (None, 1)
                    red = red_have if red_have is not None else reduce_pass(l, dcur)
                    grads[3 * l + 1], grads[3 * l + 2] = red[1], red[0]
                    # dA_{l-1}[M,Kp] = dY_l[M,C] (W^T)[Kp,C]^T (a transposed copy of the weight: reading it as stored, four
                    # strided scalars per slot, made the loader waves the bottleneck -- 8.55 vs 8.32 ms/step)
                    dprev = torch.empty((M, Kp), dtype=torch.float32, device=dev)
                    c0 = cols[0] if cols else 0
                    pad_hi = Kp - c0 - ncol
                    Wt = W2.t()[c0:c0 + ncol].contiguous()
                    if c0 > 32 or pad_hi > 32:  # (never with grouped / interpolated rows: 3 and <= 3 columns)
                        dprev.zero_()
                        pad_lo_k = pad_hi_k = 0
                    else:
                        pad_lo_k, pad_hi_k = c0, pad_hi
                    _lib.call("tp3d_gemm_rows_bnbwd_sp_f32", _lib.ptr(Y), _lib.ptr(dcur), _lib.ptr(ls[0]), _lib.ptr(ls[2]),
                              _lib.ptr(ls[3]), _lib.ptr(red[2]), _lib.ptr(red[3]), slope, _lib.ptr(Wt), M, ncol, C,
                              _lib.ptr(dprev) + 4 * c0, Kp, pad_lo_k, pad_hi_k,
                              _lib.ptr(dY) if ctx.needs_input_grad[4 + 3 * l] else None, a_ptr, ns, direction(), st)
                    if ctx.needs_input_grad[4 + 3 * l]:
                        both = None
                        if WGRAD_RED and l > 0 and acts[l - 1] is None and route(l - 1)[0] != "passes":
                            # Y_{l-1} streams through this contraction anyway: the layer below gets its reductions here
                            both = _chain_wgrad_red(dY, l, Ys, stats, slopes, dprev, training, direction())
                        if both is not None:
                            dW, red_next = both
                        else:
                            dW = _chain_wgrad(dY, l, A0, acts, Ys, stats, slopes, direction())
                        grads[3 * l] = dW[:, :cins[l]].reshape(wshapes[l])
                    dcur = dprev
                    if l == 0:
                        dA0 = dprev
                    continue
                if kind == "narrow":
                    # the first layer of grouped rows (a handful of input channels, nobody reads their gradient): the
                    # reduction pass, then dW straight from (Y, dA, A0) -- dY is never written
                    red = red_have if red_have is not None else reduce_pass(l, dcur)
                    grads[3 * l + 1], grads[3 * l + 2] = red[1], red[0]
                    dW = torch.empty((C, Kp), dtype=torch.float32, device=dev)
                    nws = _lib.workspace("gemm_tn_narrow", 4 * h.tp3d_gemm_tn_bn_narrow_workspace_floats(M, C, Kp), dev)
                    _lib.call("tp3d_gemm_tn_bn_narrow_f32", _lib.ptr(Y), _lib.ptr(dcur), _lib.ptr(ls[0]), _lib.ptr(ls[2]),
                              _lib.ptr(ls[3]), _lib.ptr(red[2]), _lib.ptr(red[3]), slope, _lib.ptr(A0), M, C, Kp, _lib.ptr(dW),
                              _lib.ptr(nws), direction(), st)
                    grads[3 * l] = dW[:, :cins[l]].reshape(wshapes[l])
                    continue
                dY = torch.empty_like(Y)
                ws = _lib.bn_workspace(M, C, dev)
                dgb = torch.empty((2, C), dtype=torch.float32, device=dev)  # dbeta, dgamma
                _lib.call("tp3d_bn_act_bwd_f32", _lib.ptr(dcur), _lib.ptr(arg) if pooled else None, _lib.ptr(Y), _lib.ptr(ls[2]),
                          _lib.ptr(ls[3]), _lib.ptr(ls[0]), _lib.ptr(ls[1]), slope, M, pool_ns if pooled else 1, C,
                          int(training), _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), _lib.ptr(dY), _lib.ptr(ws), st)
                grads[3 * l + 1], grads[3 * l + 2] = dgb[1], dgb[0]
                if ctx.needs_input_grad[4 + 3 * l]:
                    grads[3 * l] = _chain_wgrad(dY, l, A0, acts, Ys, stats, slopes)[:, :cins[l]].reshape(wshapes[l])
                if l > 0 or ctx.needs_input_grad[0]:
                    dcur = torch.mm(dY, W2)
                    if l == 0:
                        dA0 = dcur
        return (dA0, None, None, None) + tuple(grads)


def _chain_ok(rows, parts):
    """the fused layer chain serves MLPs on large row matrices whose widths suit the rows kernel's operands"""
    if not USE_MLP_CHAIN or rows.shape[0] < CHAIN_MIN_ROWS or rows.shape[1] % 4:
        return False
    for conv, bn, slope in parts:
        cout = conv.weight.shape[0]
        if getattr(conv, "bias", None) is not None or cout % 4 or cout < ROWS_GEMM_MIN_COLS or cout > 1536:
            return False
    return True


# The switches of the chain (each the A/B handle of a measured step, `bench.py --set NAME=0|1`; DESIGN.md section 5 has
# the numbers).  The first design -- prologues in the MFMA waves of the plain rows / weight-gradient kernels, the backward
# pass fused the same way (12.0 vs 10.3 ms/step) -- left the tree in round 3 together with its three entry points.
USE_MLP_CHAIN = True        # whole shared MLPs as one autograd node on the split-role / bf16-pipe kernels (else layer by layer)
CHAIN_BWD_LOADER = True     # input-gradient GEMMs form dY in their loader waves (else: apply pass + library GEMM)
CHAIN_BWD_POOLED = True     # ... also for the max-pooled last layer of a set-abstraction MLP (groups of 64, 128 ... rows)
ROW_ORDER_ALTERNATE = True  # consecutive big kernels of a chain walk the rows in opposite directions (`reverse`, tp3d_hip.h)
WGRAD_NARROW = True  # first layer of grouped rows (<= 16 input channels, no input gradient wanted): dW from (Y, dA, A0) in one
                     # streaming kernel (tp3d_gemm_tn_bn_narrow_f32) instead of the dY pass + the 64-column MFMA tile kernel
FWD_NARROW = True    # ... and its forward contraction (tp3d_gemm_rows_narrow_f32) instead of the MFMA tile kernel
WGRAD_RED = True     # a hidden layer's weight-gradient kernel also runs the BatchNorm-backward reductions of the layer below
                     # (tp3d_gemm_tn_x3_act_red_f32: Y of that layer streams through its loader waves anyway)


class _BNAct(torch.autograd.Function):
    """out = LeakyReLU(BatchNorm(Y)) on rows (M, C): the BatchNorm + activation that follows a KPConv (SimpleBlock,
    modules/KPConv/blocks.py:86-89) or any other row-major producer; with pool_ns > 0 also the max over groups of
    pool_ns consecutive rows (first maximum wins)."""

    @staticmethod
    def forward(ctx, Y, gamma, beta, bn, slope, pool_ns=0):
        dev = Y.device
        Y = Y.contiguous()
        M, C = Y.shape
        training = bn.training
        st = _lib.stream_ptr(dev)
        arg = None
        with _lib.on_device(dev):
            stats = _bn_stats(Y, M, C, gamma, beta, bn, training, dev, st)  # mean, invstd, scale, shift
            if pool_ns:
                G = M // pool_ns
                out = torch.empty((G, C), dtype=torch.float32, device=dev)
                arg = torch.empty((G, C), dtype=torch.int32, device=dev)
                _lib.call("tp3d_bn_act_maxpool_f32", _lib.ptr(Y), _lib.ptr(stats[0]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), slope, G, pool_ns, C,
                          _lib.ptr(out), _lib.ptr(arg), st)
            else:
                out = torch.empty((M, C), dtype=torch.float32, device=dev)
                _lib.call("tp3d_bn_act_f32", _lib.ptr(Y), _lib.ptr(stats[0]), _lib.ptr(stats[2]), _lib.ptr(stats[3]), slope, M, C, _lib.ptr(out), st)
        ctx.save_for_backward(Y, stats, arg)
        ctx.cfg = (slope, training, pool_ns)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        Y, stats, arg = ctx.saved_tensors
        slope, training, pool_ns = ctx.cfg
        dev = grad_out.device
        grad_out = grad_out.contiguous()
        M, C = Y.shape
        dY = torch.empty_like(Y)
        dgb = torch.empty((2, C), dtype=torch.float32, device=dev)  # dbeta, dgamma
        ws = _lib.bn_workspace(M, C, dev)
        with _lib.on_device(dev):
            _lib.call("tp3d_bn_act_bwd_f32", _lib.ptr(grad_out), _lib.ptr(arg), _lib.ptr(Y), _lib.ptr(stats[2]),
                      _lib.ptr(stats[3]), _lib.ptr(stats[0]), _lib.ptr(stats[1]), slope, M, max(pool_ns, 1), C, int(training),
                      _lib.ptr(dgb[0]), _lib.ptr(dgb[1]), _lib.ptr(dY), _lib.ptr(ws), _lib.stream_ptr(dev))
        return dY, dgb[1], dgb[0], None, None, None


def bn_act(Y, bn, slope, pool_ns=0):
    return _BNAct.apply(Y, bn.weight, bn.bias, bn, slope, pool_ns)


def relation_rows(pos, new_pos, idx):
    """(B*np*ns, 12) rows [ |d|, centroid xyz, neighbour xyz, d, 0, 0 ] of Relation-Shape convolution (no gradient:
    positions are data)."""
    dev = pos.device
    B, N, _ = pos.shape
    _, npnt, ns = idx.shape
    out = torch.empty((B * npnt * ns, 12), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        _lib.call("tp3d_relation_rows_f32", _lib.ptr(pos.contiguous()), _lib.ptr(new_pos.contiguous()),
                  _lib.ptr(idx.contiguous()), B, N, npnt, ns, 12, _lib.ptr(out), _lib.stream_ptr(dev))
    return out


class _NbrMaxPool(torch.autograd.Function):
    """max over each query's neighbours of the support rows, shadow neighbours contributing zeros."""

    @staticmethod
    def forward(ctx, x, nbr):
        dev = x.device
        xf = x.contiguous()
        Nq, Mn = nbr.shape
        M, C = xf.shape
        out = torch.empty((Nq, C), dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad[0]
        arg = torch.empty((Nq, C), dtype=torch.int32, device=dev) if need else None
        with _lib.on_device(dev):
            _lib.call("tp3d_nbr_maxpool_fwd_f32", _lib.ptr(xf), _lib.ptr(nbr), Nq, M, Mn, C, _lib.ptr(out), _lib.ptr(arg),
                      _lib.stream_ptr(dev))
        ctx.save_for_backward(nbr, arg)
        ctx.cfg = (M, C)
        return out

    @staticmethod
    def backward(ctx, g):
        nbr, arg = ctx.saved_tensors
        M, C = ctx.cfg
        dev = g.device
        g = g.contiguous()
        Nq, Mn = nbr.shape
        dx = torch.empty((M, C), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            inv, inv_bytes, ready, token = _lib.neighbour_inverse(nbr, M, dev)
            _lib.call("tp3d_nbr_maxpool_bwd_f32", _lib.ptr(g), _lib.ptr(arg), _lib.ptr(nbr), Nq, M, Mn, C, _lib.ptr(dx),
                      _lib.ptr(inv), inv_bytes, ready, _lib.stream_ptr(dev))
            _lib.inverse_built(token, dev)
        return dx, None


def nbr_maxpool(x, nbr):
    return _NbrMaxPool.apply(x.float(), nbr)


def linear_bn_act(A, conv, bn, slope, pool_ns=0):
    return _apply(_LinearBNAct, A, conv.weight, bn.weight, bn.bias, bn, slope, pool_ns, getattr(conv, "bias", None))



def _bn1d_of(m):
    """nn.BatchNorm1d inside a FastBatchNorm1d-style wrapper (attribute `batch_norm`) or the module itself."""
    inner = getattr(m, "batch_norm", m)
    if isinstance(inner, nn.BatchNorm1d) and inner.affine and inner.track_running_stats and inner.momentum is not None:
        return inner
    return None


def seq_parts(seq):
    """(linear, bn1d, slope) of nn.Sequential(Linear, BatchNorm[, activation]) or None."""
    mods = list(seq.children()) if isinstance(seq, nn.Sequential) else []
    if len(mods) not in (2, 3) or not isinstance(mods[0], nn.Linear):
        return None
    bn = _bn1d_of(mods[1])
    slope = _slope_of(mods[2] if len(mods) == 3 else None)
    if bn is None or slope is None:
        return None
    return mods[0], bn, slope


def rows_seq(seq, x):
    """Linear -> BatchNorm1d -> activation on rows (N, C) through the fused kernels when the block has that shape and
    the rows live on the GPU; otherwise the module itself."""
    parts = seq_parts(seq) if x.is_cuda else None
    if parts is None:
        return seq(x)
    lin, bn, slope = parts
    return linear_bn_act(x.float(), lin, bn, slope)


def rows_mlp(mlp, x):
    """A partial-dense MLP: nn.Sequential of [Linear, BatchNorm, activation] blocks."""
    for block in mlp.children():
        x = rows_seq(block, x)
    return x


def run_mlp(rows, parts, pool_ns=0):
    """Shared MLP over rows; the last layer optionally max-pools groups of pool_ns consecutive rows."""
    if _chain_ok(rows, parts):
        flat = []
        for conv, bn, slope in parts:
            flat += [conv.weight, bn.weight, bn.bias]
        return _apply(_MLPChain, rows, pool_ns, [(bn, slope) for _, bn, slope in parts], getattr(rows, "_tp3d_grad_cols", None), *flat)
    for i, (conv, bn, slope) in enumerate(parts):
        rows = linear_bn_act(rows, conv, bn, slope, pool_ns if i == len(parts) - 1 else 0)
    return rows


def scatter_table(idx, weight, nbins, div):
    """The inverted form of a neighbour table idx (B, rows, slots) over `nbins` support points (with the interpolation
    weights lined up, if given): what the backward pass of group_concat / interp_concat gathers through.  It depends on
    the geometry only, so a stepper that computes the geometry ahead of the training pass (dp.PipelinedStep) builds it
    there as well; pass it as `table=` and the backward pass skips the inversion."""
    dev = idx.device
    B = idx.shape[0]
    L = idx[0].numel()
    nbytes = _lib.load().tp3d_scatter_workspace_bytes(B, L, nbins, int(weight is not None))
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    idx = idx.contiguous()
    w = None if weight is None else weight.contiguous()
    with _lib.on_device(dev):
        _lib.call("tp3d_rows_scatter_invert", _lib.ptr(idx), _lib.ptr(w), B, L, div, nbins, _lib.ptr(buf), nbytes,
                  _lib.stream_ptr(dev))
    return buf


class _GroupConcat(torch.autograd.Function):
    """rows[(b,j,s)] = [pos[b,idx]-new_pos[b,j] (/r), x_cl[b,idx], 0-pad to 4k columns]; differentiable wrt x_cl."""

    @staticmethod
    def forward(ctx, pos, new_pos, x_cl, idx, radius, normalize, table=None):
        dev = pos.device
        B, N, _ = pos.shape
        _, npnt, ns = idx.shape
        C = 0 if x_cl is None else x_cl.shape[2]
        pos, new_pos, idx = pos.contiguous(), new_pos.contiguous(), idx.contiguous()
        xc = None if x_cl is None else x_cl.contiguous()
        ld = _pad4(C + 3)
        out = torch.empty((B * npnt * ns, ld), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.call("tp3d_group_concat_fwd_f32", _lib.ptr(pos), _lib.ptr(new_pos), _lib.ptr(xc), _lib.ptr(idx), B, N,
                      npnt, ns, C, ld, float(radius), int(bool(normalize)), _lib.ptr(out), _lib.stream_ptr(dev))
        ctx.save_for_backward(idx, table)
        ctx.dims = (B, N, npnt, ns, C, ld)
        return out

    @staticmethod
    def backward(ctx, grad_rows):
        idx, table = ctx.saved_tensors
        B, N, npnt, ns, C, ld = ctx.dims
        if C == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None, None, None, None
        dev = grad_rows.device
        grad_rows = grad_rows.contiguous()
        g = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        L = npnt * ns
        with _lib.on_device(dev):
            if table is not None:  # inverted ahead of the pass (scatter_table)
                _lib.call("tp3d_rows_scatter_apply_f32", _lib.ptr(grad_rows), B, L, 1, N, ld, 3, C, 0, _lib.ptr(g),
                          _lib.ptr(table), table.numel(), _lib.stream_ptr(dev))
            else:
                ws, ws_bytes = _lib.scatter_workspace(B, L, N, False, dev)
                _lib.call("tp3d_rows_scatter_bwd_f32", _lib.ptr(grad_rows), _lib.ptr(idx), None, B, L, 1, N, ld, 3, C,
                          _lib.ptr(g), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev))
        return None, None, g, None, None, None, None


def group_concat(pos, new_pos, x_cl, idx, radius, normalize, table=None):
    rows = _GroupConcat.apply(pos, new_pos, x_cl, idx, radius, normalize, table)
    if x_cl is not None:
        rows._tp3d_grad_cols = (3, x_cl.shape[2])  # _GroupConcat.backward reads these columns of the gradient and no others
    return rows


class _InterpConcat(torch.autograd.Function):
    """rows[(b,i)] = [sum_t w_t * feat_cl[b, idx_t], skip_cl[b,i]]; differentiable wrt feat_cl and skip_cl."""

    @staticmethod
    def forward(ctx, feat_cl, idx, weight, skip_cl, table=None):
        dev = feat_cl.device
        B, m, C1 = feat_cl.shape
        n = idx.shape[1]
        C2 = 0 if skip_cl is None else skip_cl.shape[2]
        feat_cl, idx, weight = feat_cl.contiguous(), idx.contiguous(), weight.contiguous()
        sk = None if skip_cl is None else skip_cl.contiguous()
        ld = _pad4(C1 + C2)
        out = torch.empty((B * n, ld), dtype=torch.float32, device=dev)
        with _lib.on_device(dev):
            _lib.call("tp3d_interp_concat_fwd_f32", _lib.ptr(feat_cl), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(sk), B,
                      m, n, C1, C2, ld, _lib.ptr(out), _lib.stream_ptr(dev))
        ctx.save_for_backward(idx, weight, table)
        ctx.dims = (B, m, n, C1, C2, ld)
        return out

    @staticmethod
    def backward(ctx, grad_rows):
        idx, weight, table = ctx.saved_tensors
        B, m, n, C1, C2, ld = ctx.dims
        dev = grad_rows.device
        grad_rows = grad_rows.contiguous()
        g_feat = None
        if ctx.needs_input_grad[0]:
            g_feat = torch.empty((B, m, C1), dtype=torch.float32, device=dev)
            with _lib.on_device(dev):
                if table is not None:  # inverted ahead of the pass (scatter_table)
                    _lib.call("tp3d_rows_scatter_apply_f32", _lib.ptr(grad_rows), B, 3 * n, 3, m, ld, 0, C1, 1,
                              _lib.ptr(g_feat), _lib.ptr(table), table.numel(), _lib.stream_ptr(dev))
                else:
                    ws, ws_bytes = _lib.scatter_workspace(B, 3 * n, m, True, dev)
                    _lib.call("tp3d_rows_scatter_bwd_f32", _lib.ptr(grad_rows), _lib.ptr(idx), _lib.ptr(weight), B, 3 * n,
                              3, m, ld, 0, C1, _lib.ptr(g_feat), _lib.ptr(ws), ws_bytes, _lib.stream_ptr(dev))
        g_skip = None
        if C2 and ctx.needs_input_grad[3]:
            g_skip = grad_rows.view(B, n, ld)[:, :, C1:C1 + C2]
        return g_feat, None, None, g_skip, None


def interp_concat(feat_cl, idx, weight, skip_cl, table=None):
    rows = _InterpConcat.apply(feat_cl, idx, weight, skip_cl, table)
    if skip_cl is not None and not (skip_cl.requires_grad and torch.is_grad_enabled()):
        # the skip columns carry no gradient (the network's input features): _InterpConcat.backward reads the interpolated
        # columns of the rows' gradient and no others
        rows._tp3d_grad_cols = (0, feat_cl.shape[2])
    return rows


def cat_rows(parts):
    """cat of (B, n, C_i) tensors on the channel axis -> (B*n, pad4(sum C_i)) rows with zero padding columns."""
    B, n = parts[0].shape[0], parts[0].shape[1]
    width = sum(p.shape[2] for p in parts)
    if _pad4(width) != width:
        parts = list(parts) + [parts[0].new_zeros((B, n, _pad4(width) - width))]
    return torch.cat(parts, 2).reshape(B * n, -1)


def idw_weights(dist):
    """(1/(d+1e-8)) / sum over the 3 neighbours, evaluated in the reference's order (dense.py:137-139)."""
    dev = dist.device
    dist = dist.contiguous()
    w = torch.empty_like(dist)
    with _lib.on_device(dev):
        _lib.call("tp3d_idw_weights_f32", _lib.ptr(dist), dist.numel() // 3, _lib.ptr(w), _lib.stream_ptr(dev))
    return w
