# round-2 evidence run on the GPU box: tests (verbose ids in the log), bench lines, rocprofv3 summaries
mkdir -p gpurun_out/r02u
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests -m gpu -q -rf > gpurun_out/r02u/tests.log 2>&1; tail -3 gpurun_out/r02u/tests.log
cp gpurun_out/parity_report.json gpurun_out/r02u/parity_report.json
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r02u/bench.json 2> gpurun_out/r02u/bench.err; tail -1 gpurun_out/r02u/bench.err
python -c "
import json;d=json.load(open('gpurun_out/r02u/bench.json'));print(d['value'],d['ms_per_step'],d['roofline'],d['forward_only']['ms_per_step'],d['forward_only'].get('gpu_over_cpu'));[print(k['entry'],k['sizes'],k['avg_ms'],k['frac_of_hbm_peak']) for k in d['north_star_kernels'][:6]]"
timeout -k 10 200 python bench.py --workload forward --steps 20 --warmup 3 > gpurun_out/r02u/bench_forward.json 2> gpurun_out/r02u/bench_forward.err
timeout -k 10 300 python bench.py --workload msg_c3 --steps 20 --warmup 3 > gpurun_out/r02u/bench_msg_c3.json 2> gpurun_out/r02u/bench_msg_c3.err
timeout -k 10 200 python bench.py --workload kpconv --steps 10 --warmup 3 > gpurun_out/r02u/bench_kpconv.json 2> gpurun_out/r02u/bench_kpconv.err
timeout -k 10 200 python bench.py --workload knn --steps 10 --warmup 3 > gpurun_out/r02u/bench_knn.json 2> gpurun_out/r02u/bench_knn.err
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02u/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02u/bench_stats.json 2> gpurun_out/r02u/bench_stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02u/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-geometry-prefetch > gpurun_out/r02u/pmc_fetch.log 2> gpurun_out/r02u/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02u/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-geometry-prefetch > gpurun_out/r02u/pmc_write.log 2> gpurun_out/r02u/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/r02u/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-geometry-prefetch > gpurun_out/r02u/pmc_sq.log 2> gpurun_out/r02u/pmc_sq.err
ls gpurun_out/r02u
