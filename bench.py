#!/usr/bin/env python
"""bench.py -- point-clouds/sec, fwd+bwd, PointNet++ SSG, B=32 N=16384 per GPU (BASELINE.json metric).

One "step" = one training pass of the hot path over one synthetic batch: zero_grad -> forward (FPS,
ball_query, grouping, grouped MLP, three_nn, three_interpolate, FP MLPs, head) -> cross-entropy -> backward
-> Adam step.  Inputs are resident in HBM before the timed region.  One process per GPU; with N > 1 the batch
is sharded per rank (B=32 per GPU, weak scaling) and the only collective is one flat gradient all-reduce over
RCCL per step (torch_points3d_amd/dp.py).  Rank 0 prints ONE JSON line.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 32
N_POINTS = 16384
FEAT = 3
NUM_CLASSES = 10
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak (MI355X_MICROARCH.md, "Peak FP32 (matrix)")


class SegStep(nn.Module):
    """tensor-in / tensor-out wrapper so DDP can own the module."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, pos, x, geometry=None):
        from torch_points3d_amd.dense import Data
        return self.net(Data(pos=pos, x=x), geometry=geometry).x  # (B, classes, N)


def make_inputs(B, N, seed, device):
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand(B, N, 3, generator=g) * 2 - 1
    x = torch.randn(B, N, FEAT, generator=g)
    y = torch.randint(0, NUM_CLASSES, (B, N), generator=g)
    return pos.to(device), x.to(device), y.to(device)


MODEL_CONFIG = "unet_3_ss"  # the reference's default SSG (applications/conf/pointnet2/unet_3_ss.yaml)


def MODEL_CONFIG_set(name):
    global MODEL_CONFIG
    MODEL_CONFIG = name


class PartSegStep(nn.Module):
    """BASELINE configs[2]: PointNet2_D (models/segmentation/pointnet2.py) -- per-cloud object category one-hot fed to
    the classifier; tensor-in / tensor-out like SegStep (scores as (B, classes, N))."""

    def __init__(self, net, num_categories):
        super().__init__()
        self.net = net
        self.num_categories = num_categories

    def forward(self, pos, x, geometry=None):
        from torch_points3d_amd.dense import Data
        B, n = pos.shape[0], pos.shape[1]
        # one object category per cloud, a fixed function of the cloud's position in the batch (synthetic)
        cat = (torch.arange(B, device=pos.device) % self.num_categories).view(B, 1).expand(B, n)
        return self.net(Data(pos=pos, x=x), cat, geometry=geometry).view(B, n, -1).transpose(1, 2)


FUSED = True  # --reference-graph: the reference's (B,C,np,ns) module graph around the HIP kernels (pure drop-in mode)


def build_model(kernels, device):
    torch.manual_seed(0)
    if MODEL_CONFIG == "pointnet2_charlesmsg":
        from torch_points3d_amd.pointnet2 import PointNet2_D
        net = PointNet2_D(FEAT, NUM_CLASSES, config=MODEL_CONFIG, num_categories=16, kernels=kernels, fused=FUSED)
        return PartSegStep(net, 16).to(device).train()
    from torch_points3d_amd.pointnet2 import PointNet2Unet
    net = PointNet2Unet(FEAT, output_nc=NUM_CLASSES, config=MODEL_CONFIG, kernels=kernels, fused=FUSED)
    return SegStep(net).to(device).train()


def seg_loss(logits, y):
    """Cross entropy over all points; --loss-layout rows feeds it the (B*N, classes) view instead (the form the
    reference's segmentation models use, models/segmentation/pointnet2.py:97-101)."""
    if not LOSS_ROWS:
        return F.cross_entropy(logits, y)
    return F.cross_entropy(logits.transpose(1, 2).reshape(-1, logits.shape[1]), y.reshape(-1))


LOSS_ROWS = False


def train_step(model, opt, pos, x, y):
    opt.zero_grad(set_to_none=True)
    logits = model(pos, x)
    loss = seg_loss(logits, y)
    loss.backward()
    opt.step()
    return loss


# ALGORITHMIC bytes of ONE launch (formulas: SURVEY.md 8d / DESIGN.md), from the entry point's size arguments
# in the order include/tp3d_hip.h declares them.
def algorithmic_bytes(name, a):
    if name == "tp3d_fps_f32":  # B, N, npoint
        B, N, npnt = a[:3]
        return B * (N * 12 + npnt * 8)
    if name == "tp3d_ball_query_dense_f32":  # B, N, np, nsample, sort
        B, N, npnt, ns = a[:4]
        return B * (N * 12 + npnt * 12 + npnt * ns * 12)
    if name == "tp3d_ball_query_partial_dense_f32":  # M, Nq, nsample, sort
        M, Nq, ns = a[:3]
        return M * 20 + Nq * 20 + Nq * ns * 12
    if name == "tp3d_three_nn_f32":  # B, n, m
        B, n, m = a[:3]
        return B * (n * 12 + m * 12 + n * 36)
    if name in ("tp3d_three_interpolate_fwd_f32", "tp3d_three_interpolate_bwd_f32"):  # B, C, m, n
        B, C, m, n = a[:4]
        return B * (C * m * 4 + n * 36 + C * n * 4)
    if name in ("tp3d_group_fwd_f32", "tp3d_group_bwd_f32"):  # B, C, N, np, ns
        B, C, N, npnt, ns = a[:5]
        return B * (C * N * 4 + npnt * ns * 8 + C * npnt * ns * 4)
    # ---- channel-last grouped-MLP kernels ("algorithmic bytes")
    if name == "tp3d_group_concat_fwd_f32":  # B, N, np, ns, C, ld
        B, N, npnt, ns, C, ld = a[:6]
        return B * (N * 12 + npnt * 12 + N * C * 4 + npnt * ns * 8 + npnt * ns * ld * 4)
    if name == "tp3d_rows_scatter_bwd_f32":  # B, L, div, nbins, ld, col0, C
        B, L, div, nbins, ld, col0, C = a[:7]
        return B * (L * 8 + (L // div) * C * 4 + nbins * C * 4)
    if name == "tp3d_rows_scatter_invert":  # B, L, div, nbins: the table in, its inverted form out
        B, L, div, nbins = a[:4]
        return B * (L * 8 + L * 4 + (nbins + 1) * 4)
    if name == "tp3d_rows_scatter_apply_f32":  # B, L, div, nbins, ld, col0, C, with_weights
        B, L, div, nbins, ld, col0, C = a[:7]
        return B * (L * 4 + (L // div) * C * 4 + nbins * C * 4)
    if name == "tp3d_bn_stats_f32":  # M, C, training
        M, C = a[:2]
        return M * C * 4
    if name == "tp3d_bn_act_f32":  # M, C
        M, C = a[:2]
        return 2 * M * C * 4
    if name == "tp3d_bn_act_maxpool_f32":  # G, ns, C
        G, ns, C = a[:3]
        return G * ns * C * 4 + G * C * 8
    if name == "tp3d_bn_act_bwd_f32":  # M, ns, C, training  (dA + Y read twice, dY written)
        M, ns, C = a[:3]
        return 3 * M * C * 4 + (M // ns) * C * 8
    if name == "tp3d_interp_concat_fwd_f32":  # B, m, n, C1, C2, ld
        B, m, n, C1, C2, ld = a[:6]
        return B * (m * C1 * 4 + n * 36 + n * C2 * 4 + n * ld * 4)
    if name == "tp3d_idw_weights_f32":  # rows
        return a[0] * 24
    if name in ("tp3d_gemm_tn_f32", "tp3d_gemm_tn_x3_f32", "tp3d_gemm_tn_x3_act_f32", "tp3d_gemm_rows_f32", "tp3d_gemm_rows_narrow_f32"):  # M, N, K
        M, N, K = a[:3]
        return (M * (N + K) + N * K) * 4
    if name == "tp3d_gemm_tn_bn_narrow_f32":  # M, N, K: Y and dA (M, N) in, A (M, K) in, (N, K) out
        M, N, K = a[:3]
        return (2 * M * N + M * K + N * K) * 4
    if name == "tp3d_gemm_tn_x3_act_red_f32":  # training, M, N, K: dY (M, N), Yp and dA (M, K) in, (N, K) out
        _, M, N, K = a[:4]
        return (M * (N + 2 * K) + N * K) * 4
    if name in ("tp3d_gemm_rows_bnact_sp_f32", "tp3d_gemm_rows_bnact_x3_f32"):  # M, N, K (the side output of the training launches, M * K more, not counted)
        M, N, K = a[:3]
        return (M * (N + K) + N * K) * 4
    if name == "tp3d_gemm_rows_bnbwd_sp_f32":  # M, N, K, ldc, pad_lo, pad_hi, ns: Y and dA (dense, or pooled + winners) in, dY and C out
        M, N, K, _, _, _, ns = a[:7]
        return (2 * M * K + (M // ns) * K * (1 if ns == 1 else 2) + M * N + N * K) * 4
    if name == "tp3d_bn_bwd_reduce_f32":  # M, ns, C, training: Y + dA read (pooled: the arg-max rows only)
        M, ns, C = a[:3]
        return 2 * M * C * 4 if ns == 1 else (M // ns) * C * 12
    if name == "tp3d_bn_finalize_f32":  # chunks, M, C
        chunks, M, C = a[:3]
        return chunks * 4 * C * 4
    if name == "tp3d_relation_rows_f32":  # B, N, np, ns, ld
        B, N, npnt, ns, ld = a[:5]
        return B * (N * 12 + npnt * 12 + npnt * ns * (8 + ld * 4))
    # ---- partial-dense (KPConv / RandLA) entry points
    if name == "tp3d_kpconv_weighted_f32":  # Nq, M, Mn, Cin, KP: neighbour ids + their rows once, weighted rows once
        Nq, M, Mn, Cin, KP = a[:5]
        return Nq * Mn * (8 + 4 * Cin) + Nq * KP * Cin * 4
    if name == "tp3d_knn_partial_dense_f32":  # num_clouds, max_cloud_points, M, Nq, k
        _, _, M, Nq, k = a[:5]
        return M * 12 + Nq * 20 + Nq * k * 12
    if name == "tp3d_knn_interpolate_fwd_f32":  # Nq, k, C, C2, ld
        Nq, k, C, C2, ld = a[:5]
        return Nq * (k * (12 + C * 4) + C2 * 4 + ld * 4)
    if name == "tp3d_nbr_maxpool_fwd_f32":  # Nq, M, Mn, C
        Nq, M, Mn, C = a[:4]
        return Nq * Mn * 8 + M * C * 4 + Nq * C * 8
    if name == "tp3d_voxel_cluster_f32":  # N: pos + batch in, cluster / order / start / last out
        return a[0] * (12 + 8 + 4 * 8)
    if name == "tp3d_voxel_bounds_f32":
        return a[0] * 20
    if name == "tp3d_cluster_mean_f32":  # K, C (rows read once through `order`; N unknown here: K clusters out)
        K, C = a[:2]
        return K * C * 8
    if name in ("tp3d_gemm_skinny_f32", "tp3d_gemm_skinny_bnact_f32"):  # M, N, K, lda
        M, N, K, lda = a[:4]
        return M * (lda + N) * 4
    if name == "tp3d_randla_relpos_f32":  # Nq, k, M
        Nq, k, M = a[:3]
        return Nq * 12 + Nq * k * (8 + 12 + 48)
    if name == "tp3d_attn_pool_fwd_f32":  # Nq, k, C, ldg, ldf
        Nq, k, C, ldg, ldf = a[:5]
        return Nq * k * (ldg + ldf) * 4 + Nq * C * 4
    return 0


def algorithmic_flops(name, a):
    if name in ("tp3d_gemm_tn_f32", "tp3d_gemm_tn_x3_f32", "tp3d_gemm_tn_x3_act_f32", "tp3d_gemm_rows_f32", "tp3d_gemm_rows_narrow_f32",
                "tp3d_gemm_tn_bn_narrow_f32"):  # M, N, K
        M, N, K = a[:3]
        return 2 * M * N * K
    if name == "tp3d_gemm_tn_x3_act_red_f32":  # training, M, N, K
        _, M, N, K = a[:4]
        return 2 * M * N * K
    if name in ("tp3d_gemm_rows_bnact_sp_f32", "tp3d_gemm_rows_bnact_x3_f32", "tp3d_gemm_rows_bnbwd_sp_f32"):  # M, N, K
        M, N, K = a[:3]
        return 2 * M * N * K
    return 0


def shape_table(summ, steps):
    """per (entry point, launch shape): launches, average ms, algorithmic MB and GB/s -- the `kernels` list of the side file"""
    rows = []
    for (name, a), (launches, total_ms) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
        avg_ms = total_ms / launches
        nbytes = algorithmic_bytes(name, a)
        rows.append({"entry": name, "sizes": list(a), "launches": launches, "avg_ms": round(avg_ms, 4),
                     "ms_per_step": round(total_ms / steps, 4), "algorithmic_MB": round(nbytes / 1e6, 3),
                     "GBps": round(nbytes / 1e9 / (avg_ms / 1e3), 2) if avg_ms > 0 else None})
    return rows


def entry_sums(summ):
    """{entry point: ms, launches, algorithmic bytes, flops} over all its launch shapes (KernelTimer.summary() in)."""
    per_entry = {}
    for (name, a), (launches, total_ms) in summ.items():
        e = per_entry.setdefault(name, {"ms": 0.0, "launches": 0, "bytes": 0, "flops": 0})
        e["ms"] += total_ms
        e["launches"] += launches
        e["bytes"] += launches * algorithmic_bytes(name, a)
        e["flops"] += launches * algorithmic_flops(name, a)
    return per_entry


def entry_table(per_entry, steps):
    return [{"entry": n, "ms_per_step": round(v["ms"] / steps, 4), "launches_per_step": v["launches"] / steps,
             "GBps": round(v["bytes"] / 1e9 / (v["ms"] / 1e3), 1) if v["ms"] > 0 and v["bytes"] else None,
             "TFLOPs": round(v["flops"] / 1e12 / (v["ms"] / 1e3), 2) if v["flops"] and v["ms"] > 0 else None}
            for n, v in sorted(per_entry.items(), key=lambda kv: -kv[1]["ms"])]


def load_traffic(entry):
    """HBM bytes per launch of `entry` from the committed rocprofv3 --pmc passes (the profiler cannot run inside this
    process); (bytes or None, short source tag).  profiles/traffic.json names the round and the command."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tpath))
    except (OSError, ValueError):
        return None, None
    return tj.get(entry), tj.get("_tag", "profiles/traffic.json")


def dominant_roofline(per_entry, steps):
    """`roofline` block for the entry point with the largest share of device time (all its launch shapes together,
    which is also how the rocprofv3 --stats summary groups them): a dense contraction is priced against the fp32 MFMA
    peak, everything else against the HBM peak, from ALGORITHMIC flops / bytes and live HIP-event durations."""
    if not per_entry:
        return None
    dom = max(per_entry, key=lambda n: per_entry[n]["ms"])
    e = per_entry[dom]
    traffic, source = load_traffic(dom)
    sec = e["ms"] / 1e3
    tflops, gbs = e["flops"] / 1e12 / sec, e["bytes"] / 1e9 / sec
    f_mfma, f_hbm = tflops / MFMA_F32_PEAK_TFLOPS, gbs / HBM_PEAK_GBS
    # a contraction whose operands are streamed once sits between the two roofs: the binding one is the roof it is closer
    # to (the other fraction is reported beside it)
    if e["flops"] and f_mfma >= f_hbm:
        roof = {"kernel": dom, "bound": "mfma", "achieved": round(tflops, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(f_mfma, 4), "traffic": traffic, "frac_hbm": round(f_hbm, 4),
                "algorithmic_flops_per_launch": int(e["flops"] / e["launches"])}
    else:
        roof = {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(f_hbm, 5), "traffic": traffic}
        if e["flops"]:
            roof["frac_mfma_f32"] = round(f_mfma, 4)
    roof["algorithmic_bytes_per_launch"] = int(e["bytes"] / e["launches"])
    roof.update({"avg_launch_ms": round(e["ms"] / e["launches"], 4), "launches": e["launches"],
                 "ms_per_step": round(e["ms"] / steps, 4)})
    if traffic is not None:
        roof["traffic_source"] = source
    return roof


def log(msg):
    sys.stderr.write("[bench %.1fs] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


T_START = time.perf_counter()

HEADLINE_LIMIT = 4096  # bytes: the driver parses the LAST stdout line; a 28 KB line (round 2) did not reach it


def _short(obj, limit=400):
    """strings of the headline are capped so that no block can blow the line up again"""
    if isinstance(obj, str):
        return obj if len(obj) <= limit else obj[:limit - 3] + "..."
    if isinstance(obj, dict):
        return {k: _short(v, limit) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_short(v, limit) for v in obj]
    return obj


def headline_json(line):
    """The one stdout line: compact separators, strings capped, and a hard size check (tests/test_bench_line_cpu.py)."""
    text = json.dumps(_short(line), separators=(",", ":"))
    if len(text) >= HEADLINE_LIMIT:
        # drop optional blocks, least important first, rather than print a line the driver cannot parse
        slim = dict(line)
        for key in ("north_star", "collective", "experiment_switches", "forward_only"):
            if key in slim and len(text) >= HEADLINE_LIMIT:
                slim.pop(key)
                text = json.dumps(_short(slim, 200), separators=(",", ":"))
    if len(text) >= HEADLINE_LIMIT:
        raise RuntimeError("bench headline is %d bytes (limit %d)" % (len(text), HEADLINE_LIMIT))
    json.loads(text)
    return text


def emit(line, detail, args):
    """Heavy tables (per-shape kernels, per-entry sums, north-star rows with their counters) go to a side file; the
    headline object is the LAST stdout line."""
    path = getattr(args, "details_out", None) or os.path.join(ROOT, "gpurun_out", "bench_detail_%s.json" % args.workload)
    if detail is not None:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                json.dump({"headline": line, **detail}, f, indent=1)
            line = dict(line, details=os.path.relpath(path, ROOT))
            log("details written to %s" % path)
        except OSError as exc:
            log("details not written (%s)" % exc)
    sys.stdout.flush()
    sys.stdout.write(headline_json(line) + "\n")
    sys.stdout.flush()


def cpu_share():
    """Host cores this process may actually use: affinity, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return min(n, 64)


def cpu_baseline(sample_b, iters):
    """The same module graph on the host: PyTorch-CPU conv/BN + the CPU oracle kernels ("port")."""
    from oracle import tpk_ref
    cores = cpu_share()
    log("cpu baseline on %d threads" % cores)
    torch.set_num_threads(cores)
    tpk_ref.set_num_threads(cores)
    model = build_model(tpk_ref, "cpu")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    pos, x, y = make_inputs(sample_b, N_POINTS, 1234, "cpu")
    train_step(model, opt, pos, x, y)  # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        train_step(model, opt, pos, x, y)
    dt = time.perf_counter() - t0
    model_name = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": sample_b * iters / dt,
        "unit": "point-clouds/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d train steps (fwd+bwd+Adam) of the same PointNet++ SSG on %d clouds of N=%d after 1 warm-up; "
                  "PyTorch-CPU conv/BN + oracle/tpk_ref_cpu.c kernels (OpenMP over clouds/queries)" % (
                      iters, sample_b, N_POINTS),
        "cpu_model": model_name,
        "seconds": dt,
    }


def time_forward(model, pos, x, steps, warmup, use_graph):
    """seconds for `steps` forward passes (train-mode BatchNorm, no autograd), replayed from one HIP graph if possible"""
    with torch.no_grad():
        for _ in range(max(warmup, 1)):
            model(pos, x)
        torch.cuda.synchronize()
        graphed = False
        if use_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    model(pos, x)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    model(pos, x)
                graphed = True
            except Exception as exc:  # noqa: BLE001 -- capture is an optimisation; eager launches measure the same kernels
                log("graph capture unavailable (%s: %s); eager launches" % (type(exc).__name__, exc))
                torch.cuda.synchronize()
        step = g.replay if graphed else (lambda: model(pos, x))
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, graphed


def cpu_forward_baseline(sample_b, iters):
    from oracle import tpk_ref
    cores = cpu_share()
    torch.set_num_threads(cores)
    tpk_ref.set_num_threads(cores)
    cmodel = build_model(tpk_ref, "cpu")
    cpos, cx, _ = make_inputs(sample_b, N_POINTS, 1234, "cpu")
    with torch.no_grad():
        cmodel(cpos, cx)
        t1 = time.perf_counter()
        for _ in range(max(iters, 1)):
            cmodel(cpos, cx)
        cdt = time.perf_counter() - t1
    return {"value": sample_b * max(iters, 1) / cdt, "unit": "point-clouds/s", "cores": cores, "kind": "port",
            "sample": "%d forward passes of the same PointNet++ SSG on %d clouds of N=%d after 1 warm-up; PyTorch-CPU "
                      "conv/BN + oracle/tpk_ref_cpu.c kernels (OpenMP over clouds/queries)" % (max(iters, 1), sample_b, N_POINTS),
            "seconds": cdt}


def shape_key(entry, sizes):
    """key of profiles/pmc_north_star.json: one counter row per (entry point, launch shape)"""
    n = {"tp3d_fps_f32": 3, "tp3d_three_nn_f32": 3}.get(entry, 4)  # the size arguments that define the launch shape
    return "%s|%s" % (entry, ",".join(str(int(v)) for v in list(sizes)[:n]))


def north_star_kernels(summ):
    """BASELINE.json's north-star kernels with their algorithmic-bytes rate against the HBM peak (SURVEY 8d formulas)
    and, where a committed rocprofv3 --pmc pass holds that very (entry point, shape), its VALU / occupancy counters
    (profiles/pmc_north_star.json, written by tools/pmc_north_star.py: one process per shape, so a row's counters belong
    to its shape only)."""
    wanted = ("tp3d_fps_f32", "tp3d_ball_query_dense_f32", "tp3d_three_nn_f32", "tp3d_group_concat_fwd_f32",
              "tp3d_interp_concat_fwd_f32")
    pmc = {}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_north_star.json"))).get("rows", {})
    except (OSError, ValueError):
        pmc = {}
    out = []
    for (name, a), (launches, total_ms) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
        if name not in wanted:
            continue
        nbytes = algorithmic_bytes(name, a)
        avg_ms = total_ms / launches
        gbs = nbytes / 1e9 / (avg_ms / 1e3)
        row = {"entry": name, "sizes": list(a)[:5], "avg_ms": round(avg_ms, 4), "algorithmic_MB": round(nbytes / 1e6, 3),
               "GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5)}
        if shape_key(name, a) in pmc:
            row["counters"] = pmc[shape_key(name, a)]
        out.append(row)
    return out


def north_star_summary(rows):
    """the headline's short form of north_star_kernels: the largest shape of each spatial kernel"""
    short = {"tp3d_fps_f32": "fps", "tp3d_ball_query_dense_f32": "ball_query", "tp3d_three_nn_f32": "three_nn"}
    out = {}
    for r in rows:  # rows are sorted by total time: the first row of an entry point is its largest shape
        k = short.get(r["entry"])
        if k and k not in out:
            out[k] = {"sizes": r["sizes"], "us": round(r["avg_ms"] * 1e3, 1), "hbm_frac": r["frac_of_hbm_peak"]}
    return out


def run_forward(args):
    """BASELINE configs[1] as written: PointNet++ SSG FORWARD, B=32, N=16384, one MI355X (train-mode BatchNorm, as in the
    reference's example, which never calls .eval()).  A step = one forward pass of the whole network including FPS,
    radius searches, grouping, the grouped MLPs and the decoder; replayed from one HIP graph.  The north star's
    ">= 30x the CPU path on SSG forward" is read off `gpu_over_cpu` of this line."""
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--workload forward is a single-GPU line")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    from torch_points3d_amd import _lib
    _lib.load()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model = build_model(None, device)
    pos, x, _ = make_inputs(B_PER_GPU, N_POINTS, 1234, device)
    dt, graphed = time_forward(model, pos, x, args.steps, args.warmup, not args.no_graph)
    with torch.no_grad():
        # per-entry HIP-event timing in an eager pass outside the timed region
        timer = _lib.KernelTimer()
        _lib.set_timer(timer)
        for _ in range(args.steps):
            model(pos, x)
        torch.cuda.synchronize()
        _lib.set_timer(None)
    per_entry = entry_sums(timer.summary())
    entries = entry_table(per_entry, args.steps)
    roofline = dominant_roofline(per_entry, args.steps)
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import tpk_ref
        cores = cpu_share()
        torch.set_num_threads(cores)
        tpk_ref.set_num_threads(cores)
        cmodel = build_model(tpk_ref, "cpu")
        cb = args.cpu_sample_clouds
        cpos, cx, _ = make_inputs(cb, N_POINTS, 1234, "cpu")
        with torch.no_grad():
            cmodel(cpos, cx)
            t1 = time.perf_counter()
            iters = max(args.cpu_sample_iters, 1)
            for _ in range(iters):
                cmodel(cpos, cx)
            cdt = time.perf_counter() - t1
        cpu = {"value": cb * iters / cdt, "unit": "point-clouds/s", "cores": cores, "kind": "port",
               "sample": "%d forward passes of the same PointNet++ SSG on %d clouds of N=%d after 1 warm-up; PyTorch-CPU "
                         "conv/BN + oracle/tpk_ref_cpu.c kernels (OpenMP over clouds/queries)" % (iters, cb, N_POINTS),
               "seconds": cdt}
    value = B_PER_GPU * args.steps / dt
    line = {"metric": "point-clouds/sec forward PointNet++SSG B=32 N=16384", "value": round(value, 2),
            "unit": "point-clouds/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PointNet++ SSG (%s) forward, train-mode BatchNorm, B=32, N=16384, FEAT=3, 10 classes, "
                                   "pos~U[-1,1]^3 (BASELINE configs[1])" % MODEL_CONFIG,
                       "launch": "hip-graph replay" if graphed else "eager"},
            "roofline": roofline, "cpu_baseline": cpu}
    if cpu:
        line["gpu_over_cpu"] = round(value / cpu["value"], 1)
    emit(line, {"entry_points": entries, "north_star_kernels": north_star_kernels(timer.summary())}, args)


def run_kpconv(args):
    """BASELINE configs[3]: KPConv rigid segmentation forward (radius neighbours + kernel-point convolution, unet_4,
    in_feat 64, 25 neighbours), one synthetic cloud of 65 536 points at one point per 0.02 voxel, eval mode, fp32.
    A step = one forward pass including grid sampling, radius searches and kNN up-sampling (nothing precomputed)."""
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--workload kpconv is a single-GPU line")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from bench_kpconv import synthetic_cloud
    from torch_points3d_amd import _lib
    from torch_points3d_amd.kpconv_blocks import PDData
    from torch_points3d_amd.kpconv_unet import KPConv
    _lib.load()
    device = torch.device("cuda", 0)
    torch.manual_seed(0)
    n = 65536
    model = KPConv("unet", input_nc=3, in_feat=64, in_grid_size=0.02, num_layers=4, output_nc=13).to(device).eval()
    pos, batch = synthetic_cloud(n, 1, 0.02)
    x = torch.cat([torch.ones(n, 1), torch.randn(n, 3)], 1)
    dpos, dbatch, dx = pos.to(device), batch.to(device), x.to(device)

    def step():
        with torch.no_grad():
            return model(PDData(pos=dpos, batch=dbatch, x=dx))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    timer = _lib.KernelTimer()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.set_timer(timer)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    _lib.set_timer(None)
    per_entry = entry_sums(timer.summary())
    roof = dominant_roofline(per_entry, args.steps)
    base = None
    if not args.no_cpu_baseline:
        from oracle import tpk_ref
        from oracle.kpconv_cpu import cpu_mirror
        tpk_ref.build()
        threads = cpu_share()
        torch.set_num_threads(threads)
        tpk_ref.set_num_threads(threads)
        cpu, routed = cpu_mirror(model)
        cpu.eval()
        with routed(), torch.no_grad():
            t1 = time.perf_counter()
            ref = cpu(PDData(pos=pos, batch=batch, x=x))
            sec = time.perf_counter() - t1
        err = float((out.x.cpu() - ref.x).abs().max() / ref.x.abs().max())
        base = {"value": n / sec, "unit": "points/s", "cores": threads, "kind": "port",
                "sample": "1 forward of the same model on the same cloud: reference block logic mirrored on the CPU with "
                          "oracle/tpk_ref_cpu.c radius search + kNN, oracle/voxel_ref.py grid sampling, PyTorch-CPU "
                          "KPConv_ops / Linear / BatchNorm", "seconds": sec, "max_rel_diff_vs_gpu": err}
    emit({
        "metric": "points/sec KPConv unet_4 forward N=65536", "value": round(n * args.steps / dt, 1), "unit": "points/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "KPConv rigid segmentation forward (applications/kpconv.py unet_4, in_feat 64, grid 0.02, 25 "
                               "neighbours, 13 classes), one cloud of 65536 points (BASELINE configs[3]); sampling and "
                               "searches inside the timed region", "launch": "eager"},
        "roofline": roof, "cpu_baseline": base},
        {"entry_points": entry_table(per_entry, args.steps), "kernels": shape_table(timer.summary(), args.steps)}, args)


def run_knn(args):
    """BASELINE configs[4] leg: RandLA-Net's random subsample (ratio 0.25, with replacement) + exact kNN (k = 16) of the
    sampled points in the full cloud (reference modules/RandLANet/modules.py:57-67), one room-shaped scene of 10^6
    surface points.  A step = sampler + neighbour search; the unit is queries per second."""
    if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--workload knn is a single-GPU line")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from bench_knn import room
    from torch_points3d_amd import _lib, torchpoints as tp
    from torch_points3d_amd.randla import RandomSampler
    _lib.load()
    device = torch.device("cuda", 0)
    n, k = 1000000, 16
    pos_cpu = room(n)
    pos = pos_cpu.to(device)
    batch = torch.zeros(n, dtype=torch.long, device=device)
    sampler = RandomSampler(ratio=0.25)

    def step():
        idx = sampler(pos, batch=batch)
        q = pos[idx]
        return idx, q, tp.knn(k, pos, q, batch, batch[idx])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        idx, q, (nbr, d2) = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nq = q.shape[0]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    by = batch[idx]
    a.record()
    for _ in range(args.steps):
        tp.knn(k, pos, q, batch, by)
    b.record()
    torch.cuda.synchronize()
    knn_ms = a.elapsed_time(b) / args.steps
    alg = n * 12 + nq * 12 + nq * k * 12.0  # support + queries read once, (idx, dist2) written once
    roof = {"kernel": "tp3d_knn_partial_dense_f32", "bound": "hbm", "achieved": round(alg / 1e9 / (knn_ms / 1e3), 2),
            "peak": 8000.0, "unit": "GB/s", "frac": round(alg / 1e9 / (knn_ms / 1e3) / 8000.0, 5), "traffic": None,
            "avg_launch_ms": round(knn_ms, 4),
            "note": "grid build + query kernels of one call; the search is VALU / latency shaped, not bandwidth shaped"}
    base = None
    if not args.no_cpu_baseline:
        from oracle import tpk_ref
        tpk_ref.build()
        threads = cpu_share()
        tpk_ref.set_num_threads(threads)
        sel = torch.randperm(nq)[:4096]
        qs = q.cpu()[sel].contiguous()
        t1 = time.perf_counter()
        ref_idx, ref_d2 = tpk_ref.knn(k, pos_cpu, qs)
        sec = time.perf_counter() - t1
        same = bool(torch.equal(ref_idx, nbr.cpu()[sel]) and torch.equal(ref_d2, d2.cpu()[sel]))
        base = {"value": sel.numel() / sec, "unit": "queries/s", "cores": threads, "kind": "port",
                "sample": "%d of the %d queries, brute force over the 10^6 points (oracle/tpk_ref_cpu.c, OpenMP over "
                          "queries)" % (sel.numel(), nq), "seconds": sec, "gpu_result_identical_on_sample": same}
    emit({
        "metric": "queries/sec random-subsample + exact 16-NN, N=10^6", "value": round(nq * args.steps / dt, 1),
        "unit": "queries/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "RandLA-Net sampling + neighbour search leg (BASELINE configs[4]): one room-shaped scene of "
                               "10^6 surface points, 250000 queries drawn with replacement, k = 16", "launch": "eager"},
        "roofline": roof, "cpu_baseline": base}, {}, args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults: the timed region's fixed cost -- two barriers, two device synchronisations -- is ~1 ms, 0.6 % of 20 steps)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-clouds", type=int, default=8)
    ap.add_argument("--cpu-sample-iters", type=int, default=8)
    ap.add_argument("--model", default="unet_3_ss", choices=["unet_3_ss", "unet_4_ss", "unet_3_ms"],
                    help="PointNet++ config; the BASELINE metric is unet_3_ss, the others are side measurements")
    ap.add_argument("--no-graph", action="store_true", help="single-GPU runs replay the train step from a captured "
                    "HIP graph (same kernels, no per-launch host cost); this flag keeps eager launches")
    ap.add_argument("--force-ddp", action="store_true", help="initialise the process group even with one rank "
                    "(rehearsal of the multi-process code path on a single-GPU box)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-process path on a box with fewer GPUs than ranks)")
    ap.add_argument("--workload", default="pointnet2", choices=["pointnet2", "msg_c3", "forward", "kpconv", "knn"],
                    help="pointnet2: the headline metric (BASELINE configs[1] shapes).  forward: the same network, "
                         "forward pass only (configs[1] as written), with a CPU forward baseline.  kpconv: BASELINE configs[3], "
                         "KPConv unet_4 forward on one 65 536-point cloud, with the CPU mirror of the same modules on "
                         "the oracle kernels as baseline.  knn: BASELINE configs[4] leg, random subsample + exact 16-NN on "
                         "a 10^6-point scene, brute-force oracle on a query sample as baseline.  (All three single GPU; "
                         "extra lines, not the headline)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: B=32 clouds per GPU (the metric's batch per device); strong: a global batch of 32 clouds "
                         "split over the ranks (32/16/8/4 per GPU at 1/2/4/8 GPUs, SURVEY 8e)")
    ap.add_argument("--no-geometry-prefetch", action="store_true",
                    help="compute sampling / searches inside the training pass instead of one step ahead on a second stream")
    ap.add_argument("--reference-graph", action="store_true",
                    help="what a user gets by switching ONLY the torch_points_kernels package: the reference's own "
                         "(B,C,npoint,nsample) Conv2d/BatchNorm2d module graph (MIOpen / rocBLAS) around the HIP spatial "
                         "kernels, instead of the channel-last fused modules; an extra line, not the headline")
    ap.add_argument("--adam", default="fused", choices=["fused", "foreach"],
                    help="torch.optim.Adam implementation: one fused kernel over the flat parameter buffer, or the "
                         "multi-tensor form (~15 short launches)")
    ap.add_argument("--loss-layout", default="nchw", choices=["rows", "nchw"],
                    help="cross entropy on the (B, classes, N) scores (default; PyTorch's spatial soft-max kernels) or on "
                         "their (B*N, classes) view (no transposing copies, but PyTorch's row soft-max is 0.7 ms slower on "
                         "10-wide rows -- measured, kept as a switch)")
    ap.add_argument("--details-out", default=None, metavar="PATH",
                    help="side file for the heavy tables (per-shape kernels, per-entry sums, north-star rows); default "
                         "gpurun_out/bench_detail_<workload>.json.  The headline stays the last stdout line, < 4 KB")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="experiment switch: set an attribute of torch_points3d_amd.fused (e.g. USE_MLP_CHAIN=0) before the "
                         "run; recorded in the JSON line")
    args = ap.parse_args()
    if args.reference_graph:
        global FUSED
        FUSED = False
    if args.set:
        from torch_points3d_amd import fused as _fz
        for kv in args.set:
            name, val = kv.split("=", 1)
            cur = getattr(_fz, name)
            setattr(_fz, name, type(cur)(int(val)) if isinstance(cur, (bool, int)) else type(cur)(val))
    if args.workload == "forward":
        MODEL_CONFIG_set(args.model)
        return run_forward(args)
    if args.workload == "kpconv":
        return run_kpconv(args)
    if args.workload == "knn":
        return run_knn(args)

    global MODEL_CONFIG, N_POINTS, NUM_CLASSES
    MODEL_CONFIG = args.model
    if args.workload == "msg_c3":
        # BASELINE configs[2]: PointNet++ MSG part segmentation as conf/models/segmentation/pointnet2.yaml:95-130 defines
        # it (pointnet2_charlesmsg + PointNet2_D head), ShapeNet-shaped input: N = 2048, 16 categories, 50 part classes
        MODEL_CONFIG, N_POINTS, NUM_CLASSES = "pointnet2_charlesmsg", 2048, 50
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= ndev:
        raise SystemExit("rank %d needs its own GPU (%d visible)" % (local_rank, ndev))
    device = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(device)
    multi = world > 1 or args.force_ddp
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=args.backend)

    from torch_points3d_amd import _lib
    from torch_points3d_amd.dense import Data
    from torch_points3d_amd.dp import PipelinedStep, ShardedStep
    _lib.load()  # fail loudly if the HIP extension is missing

    # whole clouds per rank; BatchNorm stays per rank; ShardedStep broadcasts rank 0's initial state
    if args.scaling == "strong" and B_PER_GPU % world:
        raise SystemExit("--scaling strong needs a rank count that divides %d" % B_PER_GPU)
    b_rank = B_PER_GPU // world if args.scaling == "strong" else B_PER_GPU
    model = build_model(None, device)
    pos, x, y = make_inputs(b_rank, N_POINTS, 1234 + rank, device)
    use_graph = not args.no_graph
    # one fused kernel over the flat parameter buffer (the foreach form is ~15 launches of a few microseconds each)
    make_opt = lambda params: torch.optim.Adam(params, lr=1e-3, fused=args.adam == "fused",  # noqa: E731
                                               foreach=args.adam == "foreach", capturable=use_graph)
    global LOSS_ROWS
    LOSS_ROWS = args.loss_layout == "rows"
    if args.no_geometry_prefetch or not hasattr(model.net, "precompute_geometry"):
        trainer = ShardedStep(model, make_opt, lambda: seg_loss(model(pos, x), y), world_size=world,
                              use_graph=use_graph, log=log, reduce_always=multi)
    else:
        # sampling / radius searches / 3-NN tables of step i+1 run on a second stream during step i (dp.PipelinedStep)
        net = model.net
        trainer = PipelinedStep(model, make_opt, lambda slot: net.precompute_geometry(pos, backward_tables=True),
                                lambda geo: seg_loss(model(pos, x, geometry=geo), y),
                                world_size=world, use_graph=use_graph, log=log, reduce_always=multi)
    log("model built; warm-up")
    graphed = trainer.warmup_and_capture(args.warmup)
    torch.cuda.synchronize()
    log("warm-up done (%s)" % ("hip-graph replay" if graphed else "eager launches"))

    timer = _lib.KernelTimer()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    if not graphed:
        _lib.set_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step()
    t_enqueued = time.perf_counter() - t0  # host time to enqueue the steps (diagnostic only)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.set_timer(None)
    if graphed and rank == 0:
        # per-kernel HIP-event timing needs eager launches: the same K steps again, outside the headline region
        # (rank 0 only; no collective is involved in this pass)
        saved = (trainer.world, trainer.reduce_always)
        trainer.world, trainer.reduce_always = 1, False
        _lib.set_timer(timer)
        eager = getattr(trainer, "serial_eager_step", trainer.eager_step)  # one stream: clean event brackets
        for _ in range(args.steps):
            eager()
        torch.cuda.synchronize()
        _lib.set_timer(None)
        trainer.world, trainer.reduce_always = saved
    log("timed region done: %.2f ms/step (host enqueue %.2f ms/step)" % (dt / args.steps * 1e3,
                                                                          t_enqueued / args.steps * 1e3))

    coll_ms = None
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the step's only collective, timed on its own (device events around K all-reduce + divide rounds)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        dist.barrier()
        e0.record()
        for _ in range(args.steps):
            trainer._reduce()
        e1.record()
        torch.cuda.synchronize()
        coll_ms = round(e0.elapsed_time(e1) / args.steps, 4)

    if rank == 0:
        summ = timer.summary()
        kernels = shape_table(summ, args.steps)
        per_entry = entry_sums(summ)
        roofline = dominant_roofline(per_entry, args.steps)
        entries = entry_table(per_entry, args.steps)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_sample_clouds, args.cpu_sample_iters)
        # the north star's ">= 30x the CPU path" is defined on the SSG FORWARD pass: measured in the same run, after the
        # headline region (BASELINE configs[1] as written; train-mode BatchNorm as the reference's example)
        forward_only = None
        if world == 1:
            fdt, fgraphed = time_forward(model, pos, x, args.steps, 2, use_graph)
            forward_only = {"ms_per_step": round(fdt / args.steps * 1e3, 3), "value": round(b_rank * args.steps / fdt, 2),
                            "unit": "point-clouds/s", "launch": "hip-graph replay" if fgraphed else "eager"}
            if not args.no_cpu_baseline:
                fcpu = cpu_forward_baseline(args.cpu_sample_clouds, max(args.cpu_sample_iters // 2, 2))
                forward_only["cpu_baseline"] = fcpu
                forward_only["gpu_over_cpu"] = round(forward_only["value"] / fcpu["value"], 1)
        value = world * b_rank * args.steps / dt
        grouping = "MSG" if (MODEL_CONFIG.endswith("_ms") or args.workload == "msg_c3") else "SSG"
        ns_rows = north_star_kernels(summ)  # the per-kernel pass runs on one stream: spatial kernels timed on their own
        line = {
            "metric": "point-clouds/sec fwd+bwd PointNet++%s B=32 N=%d" % (grouping, N_POINTS),
            "value": round(value, 2),
            "unit": "point-clouds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "PointNet++ %s (%s) train step fwd+bwd+Adam, B=%d per GPU, N=%d, FEAT=3, %d classes, "
                                   "pos~U[-1,1]^3 (BASELINE configs[%d])"
                                   % (grouping, MODEL_CONFIG, b_rank, N_POINTS, NUM_CLASSES, 2 if grouping == "MSG" else 1),
                       "launch": ("hip-graph replay" if graphed else "eager") + (
                           "" if args.no_geometry_prefetch else "; geometry of step i+1 on a 2nd stream during step i"),
                       "global_batch": world * b_rank, "points": N_POINTS,
                       "parallelism": "dp%d, whole clouds per rank, one flat gradient all-reduce (RCCL) per step" % world},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "forward_only": forward_only,
            "north_star": north_star_summary(ns_rows),
        }
        if multi:
            line["collective"] = {"what": "flat fp32 gradient all-reduce + divide per step (RCCL)",
                                  "bytes": int(trainer.flat.numel() * 4), "ms_per_step": coll_ms}
        if args.set:
            line["experiment_switches"] = args.set
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
        emit(line, {"north_star_kernels": ns_rows, "entry_points": entries, "kernels": kernels}, args)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
