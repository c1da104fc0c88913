"""Turns the rocprofv3 output of one gpurun call (gpurun_out/<tag>/{stats,pmc_fetch,pmc_write}) into the committed
summaries under profiles/: kernel stats CSV, per-kernel HBM traffic CSV and traffic.json (bytes per C-ABI entry point
launch, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
shutil.copy(glob.glob(os.path.join(src, "stats/*/*kernel_stats.csv"))[0], "profiles/%s_bench_kernel_stats.csv" % tag)
shutil.copy(os.path.join(src, "bench_stats.json"), "profiles/%s_bench_under_rocprof.json" % tag)


def load(path, counter):
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]] += float(r["Counter_Value"])
            cnt[r["Kernel_Name"]] += 1
    return acc, cnt


f, fc = load(glob.glob(os.path.join(src, "pmc_fetch/*/*counter_collection.csv"))[0], "FETCH_SIZE")
w, wc = load(glob.glob(os.path.join(src, "pmc_write/*/*counter_collection.csv"))[0], "WRITE_SIZE")
rows = sorted(((k, fc[k], f[k] / fc[k], (w[k] / wc[k]) if k in w else 0.0) for k in f),
              key=lambda r: -(2 * r[2] + r[3]) * r[1])
with open("profiles/%s_pmc_traffic_by_kernel.csv" % tag, "w") as out:
    out.write("kernel,launches,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch_raw,hbm_MB_per_launch_corrected\n")
    for k, n, fk, wk in rows:
        out.write('"%s",%d,%.1f,%.1f,%.2f\n' % (k.replace('"', "'"), n, fk, wk, (2 * fk + wk) * 1024 / 1e6))

groups = {  # entry point -> (kernel name fragments, fragment that counts entry-point launches)
    "tp3d_fps_f32": (["fps_reg_kernel", "fps_generic_kernel"], "fps_"),
    "tp3d_ball_query_dense_f32": (["ball_query_dense_kernel", "grid_build_kernel", "grid_query_kernel"], "y_kernel"),
    "tp3d_three_nn_f32": (["three_nn_kernel"], None),
    "tp3d_bn_stats_f32": (["colreduce_partial_kernel<4, 0>", "colreduce_partial_kernel<1, 0>"], None),
    "tp3d_bn_act_f32": (["bn_act_kernel"], None),
    "tp3d_bn_act_maxpool_f32": (["bn_act_maxpool_kernel"], None),
    "tp3d_bn_act_bwd_f32": (["colreduce_partial_kernel<4, 1>", "colreduce_partial_kernel<1, 1>", "bn_bwd_finalize_kernel",
                             "bn_act_bwd_apply_kernel", "bn_pool_bwd"], "bn_bwd_finalize_kernel"),
    "tp3d_gemm_tn_f32": (["gemm_tn_partial_kernel", "gemm_tn_reduce_kernel"], "gemm_tn_reduce_kernel"),
    "tp3d_gemm_rows_f32": (["gemm_rows_kernel"], None),
    "tp3d_rows_scatter_bwd_f32": (["csr_transpose_kernel", "rows_gather_sum_kernel"], "rows_gather_sum_kernel"),
    "tp3d_group_concat_fwd_f32": (["group_concat_fwd_kernel"], None),
    "tp3d_interp_concat_fwd_f32": (["interp_concat_fwd_kernel"], None),
}
out = {}
for entry, (pats, launchpat) in groups.items():
    fb = sum(v for k, v in f.items() if any(p in k for p in pats))
    wb = sum(v for k, v in w.items() if any(p in k for p in pats))
    n = sum(c for k, c in fc.items() if (launchpat or pats[0]) in k)
    if n:
        out[entry] = int((2 * fb + wb) * 1024 / n)
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
