"""Turns the rocprofv3 output of one gpurun call (gpurun_out/<src>/{stats,pmc_fetch,pmc_write,pmc_sq}) into the
committed summaries under profiles/ (prefix <tag>): kernel stats CSV, per-kernel HBM traffic CSV, traffic.json (HBM bytes
per C-ABI entry point launch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) and the VALU / occupancy
counters of the spatial kernels (<tag>_pmc_spatial.json, read back by bench.py's north_star_kernels block).

    python tools/make_profiles.py r02f r02
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src_tag = sys.argv[1] if len(sys.argv) > 1 else "r02f"
tag = sys.argv[2] if len(sys.argv) > 2 else src_tag
src = os.path.join("gpurun_out", src_tag)
os.makedirs("profiles", exist_ok=True)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern))
    return hits[0] if hits else None


stats = one("stats/*/*kernel_stats.csv")
if stats:
    shutil.copy(stats, "profiles/%s_bench_kernel_stats.csv" % tag)
if os.path.exists(os.path.join(src, "bench_stats.json")):
    shutil.copy(os.path.join(src, "bench_stats.json"), "profiles/%s_bench_under_rocprof.json" % tag)
for w in ("", "_forward", "_msg_c3", "_kpconv", "_knn", "_unet_4_ss", "_reference_graph"):
    p = os.path.join(src, "bench%s.json" % w)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, "profiles/%s_bench%s_line.json" % (tag, w))
    p = os.path.join(src, "bench%s_detail.json" % w)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, "profiles/%s_bench%s_detail.json" % (tag, w))
if os.path.exists(os.path.join(src, "trace_gaps.txt")):
    shutil.copy(os.path.join(src, "trace_gaps.txt"), "profiles/%s_trace_gaps.txt" % tag)
if os.path.exists(os.path.join(src, "parity_report.json")):
    shutil.copy(os.path.join(src, "parity_report.json"), "profiles/%s_parity_report.json" % tag)


def load(path, counters):
    acc = {c: collections.defaultdict(float) for c in counters}
    cnt = collections.defaultdict(int)
    seen = set()
    for r in csv.DictReader(open(path)):
        c = r["Counter_Name"]
        if c in acc:
            acc[c][r["Kernel_Name"]] += float(r["Counter_Value"])
            key = (r["Kernel_Name"], r.get("Dispatch_Id"))
            if key not in seen:
                seen.add(key)
                cnt[r["Kernel_Name"]] += 1
    return acc, cnt


GROUPS = {  # entry point -> (kernel name fragments, fragment that counts entry-point launches)
    "tp3d_fps_f32": (["fps_reg_kernel", "fps_generic_kernel"], "fps_"),
    "tp3d_ball_query_dense_f32": (["ball_query_dense_kernel", "grid_build", "grid_query_kernel"], "y_kernel"),
    "tp3d_three_nn_f32": (["three_nn_kernel", "three_nn_grid_kernel"], "three_nn"),
    "tp3d_bn_stats_f32": (["colreduce_partial_kernel<4, 0>", "colreduce_partial_kernel<1, 0>"], None),
    "tp3d_bn_act_f32": (["bn_act_kernel"], None),
    "tp3d_bn_act_maxpool_f32": (["bn_act_maxpool_kernel"], None),
    "tp3d_bn_act_bwd_f32": (["colreduce_partial_kernel<4, 1>", "colreduce_partial_kernel<1, 1>", "bn_bwd_finalize_kernel",
                             "bn_act_bwd_apply_kernel", "bn_pool_bwd"], "bn_bwd_finalize_kernel"),
    "tp3d_gemm_tn_f32": (["gemm_tn_partial_kernel"], "gemm_tn_partial_kernel"),
    # gemm_tn_x3_kernel<WM, WN, STRIP, BR, TERMS, RED>: the plain and activated-operand forms share the instantiations
    "tp3d_gemm_tn_x3_f32": (["6, false>", "9, false>"], "gemm_tn_x3_kernel"),
    "tp3d_gemm_tn_x3_act_f32": (["6, false>", "9, false>"], "gemm_tn_x3_kernel"),
    "tp3d_gemm_tn_x3_act_red_f32": (["6, true>"], "gemm_tn_x3_kernel"),
    "tp3d_gemm_rows_narrow_f32": (["gemm_rows_narrow_kernel"], None),
    "tp3d_gemm_tn_bn_narrow_f32": (["gemm_tn_narrow_bn_kernel"], None),
    # gemm_rows_sp_kernel<STATS, PRO, BN>: PRO 1 = forward (previous BatchNorm + activation), 2 / 3 = input gradient
    "tp3d_gemm_rows_bnbwd_sp_f32": (["gemm_rows_sp_kernel<0, 2,", "gemm_rows_sp_kernel<0, 3,"], None),
    "tp3d_gemm_rows_bnact_sp_f32": (["gemm_rows_sp_kernel<2, 1,", "gemm_rows_sp_kernel<0, 1,"], None),
    "tp3d_bn_bwd_reduce_f32": (["colreduce_partial_kernel<4, 1>", "colreduce_partial_kernel<1, 1>", "bn_pool_bwd_partial_kernel", "bn_bwd_finalize_kernel"], "bn_bwd_finalize_kernel"),
    "tp3d_gemm_rows_f32": (["gemm_rows_wide_kernel", "gemm_rows_kernel", "gemm_rows_sum_slabs_kernel"], "gemm_rows_"),
    "tp3d_rows_scatter_bwd_f32": (["csr_transpose_kernel", "rows_gather_sum_kernel"], "rows_gather_sum_kernel"),
    "tp3d_group_concat_fwd_f32": (["group_concat_fwd"], None),
    "tp3d_interp_concat_fwd_f32": (["interp_concat_fwd"], None),
}

fpath, wpath = one("pmc_fetch/*/*counter_collection.csv"), one("pmc_write/*/*counter_collection.csv")
if fpath and wpath:
    fa, fc = load(fpath, ["FETCH_SIZE"])
    wa, wc = load(wpath, ["WRITE_SIZE"])
    f, w = fa["FETCH_SIZE"], wa["WRITE_SIZE"]
    rows = sorted(((k, fc[k], f[k] / fc[k], (w[k] / wc[k]) if k in w and wc[k] else 0.0) for k in f if fc[k]),
                  key=lambda r: -(2 * r[2] + r[3]) * r[1])
    with open("profiles/%s_pmc_traffic_by_kernel.csv" % tag, "w") as out:
        out.write("kernel,launches,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch_raw,"
                  "hbm_MB_per_launch_corrected\n")
        for k, n, fk, wk in rows:
            out.write('"%s",%d,%.1f,%.1f,%.2f\n' % (k.replace('"', "'"), n, fk, wk, (2 * fk + wk) * 1024 / 1e6))
    traffic = {"_tag": "profiles/traffic.json (round %s rocprofv3 --pmc passes, not this run)" % tag, "_source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) of `python3 bench.py "
                          "--steps 3 --warmup 1 --no-graph --no-geometry-prefetch` in round %s (gpurun_out/%s); "
                          "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB summed over the entry point's kernels / its launches; "
                          "profiles/%s_pmc_traffic_by_kernel.csv holds the per-kernel rows" % (tag, src_tag, tag)}
    for entry, (pats, launchpat) in GROUPS.items():
        fb = sum(v for k, v in f.items() if any(p in k for p in pats))
        wb = sum(v for k, v in w.items() if any(p in k for p in pats))
        n = sum(c for k, c in fc.items() if (launchpat or pats[0]) in k)
        if n:
            traffic[entry] = int((2 * fb + wb) * 1024 / n)
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
    print(json.dumps(traffic, indent=1))

spath = one("pmc_sq/*/*counter_collection.csv")
if spath:
    names = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS",
             "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"]
    acc, cnt = load(spath, names)
    spatial = {"source": "rocprofv3 --pmc %s --kernel-trace over `python3 bench.py --steps 3 --warmup 1 --no-graph "
                         "--no-geometry-prefetch`, round %s (gpurun_out/%s); sums over the entry point's kernels divided by "
                         "its launches" % (" ".join(names), tag, src_tag), "kernels": {}}
    for entry in ("tp3d_fps_f32", "tp3d_ball_query_dense_f32", "tp3d_three_nn_f32", "tp3d_group_concat_fwd_f32",
                  "tp3d_interp_concat_fwd_f32"):
        pats, launchpat = GROUPS[entry]
        n = sum(c for k, c in cnt.items() if (launchpat or pats[0]) in k)
        if not n:
            continue
        row = {}
        for c in names:
            row[c] = round(sum(v for k, v in acc[c].items() if any(p in k for p in pats)) / n, 1)
        if row.get("SQ_WAVE_CYCLES"):
            # share of resident wave-cycles in which a VALU instruction was issuing / the wave waited on anything
            row["valu_active_per_wave_cycle"] = round(row["SQ_ACTIVE_INST_VALU"] / row["SQ_WAVE_CYCLES"], 4)
            row["wait_any_per_wave_cycle"] = round(row["SQ_WAIT_ANY"] / row["SQ_WAVE_CYCLES"], 4)
        if row.get("SQ_BUSY_CYCLES"):
            row["mean_waves_resident"] = round(row["SQ_WAVE_CYCLES"] / row["SQ_BUSY_CYCLES"], 2)
        row["launches_sampled"] = n
        spatial["kernels"][entry] = row
    json.dump(spatial, open("profiles/%s_pmc_spatial.json" % tag, "w"), indent=1)
    print(json.dumps(spatial, indent=1))


# ---- per-shape counters of the north-star kernels (tools/pmc_north_star.py, one rocprofv3 process per shape)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
try:
    from pmc_north_star import CASES
except Exception:  # noqa: BLE001 (torch missing where the summaries are made: the table is static)
    CASES = {}
names = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_WAIT_INST_ANY",
         "SQ_WAIT_ANY"]
rows_out = {}
for case, (entry, sizes) in CASES.items():
    path = one("pmc_ns_%s/*/*counter_collection.csv" % case)
    if not path:
        continue
    pats, launchpat = GROUPS[entry]
    acc, cnt = load(path, names)
    n = sum(c for k, c in cnt.items() if (launchpat or pats[0]) in k)
    if not n:
        continue
    row = {c: round(sum(v for k, v in acc[c].items() if any(p in k for p in pats)) / n, 1) for c in names}
    if row.get("SQ_WAVE_CYCLES"):
        row["valu_active_per_wave_cycle"] = round(row["SQ_ACTIVE_INST_VALU"] / row["SQ_WAVE_CYCLES"], 4)
        row["wait_any_per_wave_cycle"] = round(row["SQ_WAIT_ANY"] / row["SQ_WAVE_CYCLES"], 4)
    if row.get("SQ_BUSY_CYCLES"):
        row["mean_waves_resident"] = round(row["SQ_WAVE_CYCLES"] / row["SQ_BUSY_CYCLES"], 2)
    row["launches_sampled"] = n
    rows_out["%s|%s" % (entry, ",".join(str(v) for v in sizes[:4]))] = row
if rows_out:
    doc = {"source": "rocprofv3 --pmc %s --kernel-trace -- python3 tools/pmc_north_star.py <case>, one process per (entry point, "
                     "shape), round %s (gpurun_out/%s); sums over the entry point's kernels divided by its launches" % (
                         " ".join(names), tag, src_tag), "rows": rows_out}
    json.dump(doc, open("profiles/pmc_north_star.json", "w"), indent=1)
    shutil.copy("profiles/pmc_north_star.json", "profiles/%s_pmc_north_star.json" % tag)
    print(json.dumps(doc, indent=1))
