# round-3 evidence run on the GPU box (one gpurun call): tests, bench lines, rocprofv3 summaries.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/final_r03.sh'      then  python tools/make_profiles.py r03u r03
out=gpurun_out/r03u
mkdir -p $out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -rf > $out/tests.log 2>&1; tail -3 $out/tests.log
cp gpurun_out/parity_report.json $out/parity_report.json
timeout -k 10 300 python bench.py --details-out $out/bench_detail.json > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.err
timeout -k 10 200 python bench.py --workload forward --steps 20 --warmup 3 --details-out $out/bench_forward_detail.json > $out/bench_forward.json 2> $out/bench_forward.err
timeout -k 10 300 python bench.py --workload msg_c3 --steps 20 --warmup 3 --details-out $out/bench_msg_c3_detail.json > $out/bench_msg_c3.json 2> $out/bench_msg_c3.err
timeout -k 10 300 python bench.py --model unet_4_ss --steps 10 --warmup 3 --details-out $out/bench_unet_4_ss_detail.json > $out/bench_unet_4_ss.json 2> $out/bench_unet_4_ss.err
timeout -k 10 200 python bench.py --workload kpconv --steps 10 --warmup 3 --details-out $out/bench_kpconv_detail.json > $out/bench_kpconv.json 2> $out/bench_kpconv.err
timeout -k 10 200 python bench.py --workload knn --steps 10 --warmup 3 --details-out $out/bench_knn_detail.json > $out/bench_knn.json 2> $out/bench_knn.err
timeout -k 10 200 python bench.py --reference-graph --steps 10 --warmup 2 --no-cpu-baseline --details-out $out/bench_reference_graph_detail.json > $out/bench_reference_graph.json 2> $out/bench_reference_graph.err
echo "bench lines done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --details-out $out/bench_stats_detail.json > $out/bench_stats.json 2> $out/bench_stats.err
python tools/trace_gaps.py $out/stats/*/*kernel_trace.csv 8 > $out/trace_gaps.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-geometry-prefetch --details-out $out/pmc_fetch_detail.json > $out/pmc_fetch.log 2> $out/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-geometry-prefetch --details-out $out/pmc_write_detail.json > $out/pmc_write.log 2> $out/pmc_write.err
echo "traffic passes done"
for c in fps_sa1 fps_sa2 ball_sa1 ball_sa2 three_nn_fp3 three_nn_fp2; do
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/pmc_ns_$c -- python3 tools/pmc_north_star.py $c > $out/pmc_ns_$c.log 2>&1
done
rm -rf $out/stats/*/*agent_info.csv
ls $out
