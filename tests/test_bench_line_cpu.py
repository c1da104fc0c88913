"""bench.py's stdout contract: the LAST line is one compact JSON object the driver can parse (< 4 KB), whatever the
per-kernel tables hold -- those go to a side file.  Round 2's line was 28 KB and reached the driver as `parsed: null`."""
import io
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def canned_summary():
    """KernelTimer.summary() of a headline run, as shapes: 17 weight-gradient shapes, the forward / backward GEMMs, the
    BatchNorm passes, the spatial kernels (two shapes each)."""
    summ = {}
    gemm = [(524288, 128, 128), (524288, 128, 131), (262144, 128, 131), (262144, 256, 128), (1048576, 128, 64),
            (1048576, 64, 6), (1048576, 64, 64), (524288, 10, 128), (262144, 128, 128), (4096, 1024, 512),
            (4096, 512, 256), (4096, 256, 259), (16384, 256, 1280), (16384, 128, 384), (16384, 256, 256),
            (4096, 256, 256), (524288, 128, 132)]
    for i, (M, N, K) in enumerate(gemm):
        summ[("tp3d_gemm_tn_f32", (M, N, K, 0))] = (20, 2.0 + 0.1 * i)
        summ[("tp3d_gemm_rows_bnact_sp_f32", (M, N, K, 0))] = (20, 1.5)
        summ[("tp3d_gemm_rows_bnbwd_sp_f32", (M, N, K, N, 0, 0, 1, 0))] = (20, 1.9)
        summ[("tp3d_bn_bwd_reduce_f32", (M, 1, N, 1))] = (20, 0.6)
        summ[("tp3d_bn_finalize_f32", (1024, M, N, 0))] = (20, 0.2)
    summ[("tp3d_fps_f32", (32, 16384, 512))] = (20, 15.4)
    summ[("tp3d_fps_f32", (32, 512, 128))] = (20, 1.2)
    summ[("tp3d_ball_query_dense_f32", (32, 16384, 512, 64, 0, 0, 0))] = (20, 0.9)
    summ[("tp3d_ball_query_dense_f32", (32, 512, 128, 64, 0, 0, 0))] = (20, 0.25)
    summ[("tp3d_three_nn_f32", (32, 16384, 512))] = (20, 0.66)
    summ[("tp3d_three_nn_f32", (32, 512, 128))] = (20, 0.2)
    return summ


def test_entry_sums_and_dominant_roofline():
    per = bench.entry_sums(canned_summary())
    roof = bench.dominant_roofline(per, 20)
    assert roof["kernel"] == "tp3d_gemm_tn_f32" and roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s"
    flops = sum(2 * M * N * K * 20 for (n, (M, N, K, *_)), _ in canned_summary().items() if n == "tp3d_gemm_tn_f32")
    ms = sum(t for (n, _), (_, t) in canned_summary().items() if n == "tp3d_gemm_tn_f32")
    assert roof["achieved"] == pytest.approx(flops / 1e12 / (ms / 1e3), rel=1e-3)
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=1e-3)
    assert roof["launches"] == 17 * 20


def test_roofline_names_the_roof_the_kernel_is_closer_to():
    """a contraction that streams its operands once is priced against both roofs; the binding one is reported"""
    summ = {("tp3d_gemm_rows_bnbwd_sp_f32", (524288, 128, 128, 128, 0, 0, 1, 0)): (20, 20 * 0.2195)}
    roof = bench.dominant_roofline(bench.entry_sums(summ), 20)
    assert roof["bound"] == "hbm" and roof["frac"] > roof["frac_mfma_f32"] > 0.3
    summ = {("tp3d_gemm_tn_f32", (524288, 128, 128, 0)): (20, 20 * 0.170)}
    roof = bench.dominant_roofline(bench.entry_sums(summ), 20)
    assert roof["bound"] == "mfma" and roof["frac"] > roof["frac_hbm"]


def test_every_workload_roofline_has_numbers():
    """the KPConv line's roofline was {achieved: null, frac: null} whenever a GEMM dominated"""
    summ = {("tp3d_gemm_rows_f32", (65536, 64, 960, 0)): (10, 5.1),
            ("tp3d_kpconv_weighted_f32", (65536, 65536, 25, 64, 15, 0, 0)): (10, 4.2),
            ("tp3d_bn_act_f32", (65536, 64, 0)): (38, 3.2)}
    roof = bench.dominant_roofline(bench.entry_sums(summ), 10)
    assert roof["achieved"] and roof["frac"] and roof["bound"] == "mfma"
    summ[("tp3d_kpconv_weighted_f32", (65536, 65536, 25, 64, 15, 0, 0))] = (10, 9.0)
    roof = bench.dominant_roofline(bench.entry_sums(summ), 10)
    assert roof["kernel"] == "tp3d_kpconv_weighted_f32" and roof["bound"] == "hbm" and roof["achieved"] > 0


def _full_line(summ):
    per = bench.entry_sums(summ)
    cpu = {"value": 6.1, "unit": "point-clouds/s", "cores": 16, "kind": "port", "sample": "x" * 300,
           "cpu_model": "AMD EPYC 9575F 64-Core Processor", "seconds": 10.4}
    ns_rows = bench.north_star_kernels(summ)
    line = {"metric": "point-clouds/sec fwd+bwd PointNet++SSG B=32 N=16384", "value": 3850.0, "unit": "point-clouds/s",
            "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 8.31, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w" * 200, "launch": "l" * 100, "global_batch": 256, "points": 16384,
                       "parallelism": "p" * 90},
            "roofline": bench.dominant_roofline(per, 20), "cpu_baseline": cpu,
            "forward_only": {"ms_per_step": 3.6, "value": 8800.0, "unit": "point-clouds/s", "launch": "hip-graph replay",
                             "cpu_baseline": dict(cpu), "gpu_over_cpu": 650.0},
            "north_star": bench.north_star_summary(ns_rows),
            "collective": {"what": "c" * 60, "bytes": 5500000, "ms_per_step": 0.05},
            "gpu_over_cpu": 631.0}
    detail = {"north_star_kernels": ns_rows, "entry_points": bench.entry_table(per, 20),
              "kernels": [{"entry": n, "sizes": list(a), "launches": c, "avg_ms": t / c} for (n, a), (c, t) in summ.items()]}
    return line, detail


def test_headline_is_small_and_round_trips(tmp_path, monkeypatch):
    line, detail = _full_line(canned_summary())
    text = bench.headline_json(line)
    assert len(text) < 4096
    back = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline",
                "cpu_baseline", "forward_only"):
        assert key in back, key
    assert back["roofline"]["frac"] > 0 and back["cpu_baseline"]["kind"] == "port"
    assert back["north_star"]["ball_query"]["sizes"][:4] == [32, 16384, 512, 64]
    # emit(): the heavy tables land in the side file, stdout's last line is the headline
    out = io.StringIO()
    monkeypatch.setattr(sys, "stdout", out)
    args = types.SimpleNamespace(details_out=str(tmp_path / "detail.json"), workload="pointnet2")
    bench.emit(line, detail, args)
    last = out.getvalue().strip().splitlines()[-1]
    assert len(last) < 4096 and json.loads(last)["value"] == 3850.0
    side = json.load(open(tmp_path / "detail.json"))
    assert len(side["kernels"]) == len(canned_summary()) and side["headline"]["metric"] == line["metric"]


def test_headline_never_exceeds_the_limit_even_with_bloated_blocks():
    line, _ = _full_line(canned_summary())
    line["experiment_switches"] = ["X=%d" % i for i in range(400)]
    line["config"]["workload"] = "w" * 5000
    text = bench.headline_json(line)
    assert len(text) < 4096 and json.loads(text)["metric"] == line["metric"]


def test_north_star_counters_attach_per_shape_only(tmp_path, monkeypatch):
    """counters of one launch shape must not be shown on another shape of the same entry point"""
    prof = tmp_path / "profiles"
    prof.mkdir()
    key = bench.shape_key("tp3d_ball_query_dense_f32", (32, 16384, 512, 64, 0, 0, 0))
    json.dump({"source": "test", "rows": {key: {"SQ_WAVES": 1024.0}}}, open(prof / "pmc_north_star.json", "w"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rows = bench.north_star_kernels(canned_summary())
    with_counters = [r for r in rows if "counters" in r]
    assert len(with_counters) == 1 and with_counters[0]["sizes"][:4] == [32, 16384, 512, 64]
