mkdir -p gpurun_out/r02
timeout -k 10 400 python -m pytest tests -m gpu -q > gpurun_out/r02/t10_all.log 2>&1; tail -4 gpurun_out/r02/t10_all.log
run() { tag=$1; shift; timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r02/e2_$tag.json 2> gpurun_out/r02/e2_$tag.err; python -c "
import json;d=json.load(open('gpurun_out/r02/e2_$tag.json'));print('$tag',d['value'],d['ms_per_step'])"; }
run base
run wgrad --set OVERLAP_WGRAD=1
run wgrad_noprefetch --set OVERLAP_WGRAD=1 --no-geometry-prefetch
timeout -k 10 300 python bench.py --workload msg_c3 --steps 20 --warmup 3 > gpurun_out/r02/b4_c3.json 2> gpurun_out/r02/b4_c3.err; tail -2 gpurun_out/r02/b4_c3.err; python -c "
import json;d=json.load(open('gpurun_out/r02/b4_c3.json'));print(d['metric'],d['value'],d['ms_per_step'],d['cpu_baseline']['value'])"
