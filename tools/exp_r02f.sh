# pooled last layers of the chain on the split-role input-gradient GEMM (CHAIN_BWD_POOLED) on / off
mkdir -p gpurun_out/r02n
cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift; timeout -k 10 280 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r02n/$tag.json 2> gpurun_out/r02n/$tag.err; python -c "
import json,sys;d=json.load(open('gpurun_out/r02n/$tag.json'));print('$tag',d['value'],d['ms_per_step'],(d.get('forward_only') or {}).get('ms_per_step'))"; }
run pool0 --set CHAIN_BWD_POOLED=0 &&
run pool1 --set CHAIN_BWD_POOLED=1 &&
run c3_pool0 --workload msg_c3 --set CHAIN_BWD_POOLED=0 &&
run c3_pool1 --workload msg_c3 --set CHAIN_BWD_POOLED=1
