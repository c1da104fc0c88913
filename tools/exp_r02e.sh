# input-gradient GEMMs of the layer chain with dY formed in the loader waves (CHAIN_BWD_LOADER) against apply pass + library GEMM
mkdir -p gpurun_out/r02n
cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift; timeout -k 10 280 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r02n/$tag.json 2> gpurun_out/r02n/$tag.err; python -c "
import json,sys;d=json.load(open('gpurun_out/r02n/$tag.json'));print('$tag',d['value'],d['ms_per_step'],(d.get('forward_only') or {}).get('ms_per_step'))"; }
run bwd0 --set CHAIN_BWD_LOADER=0 &&
run bwd1 --set CHAIN_BWD_LOADER=1 &&
run c3_bwd0 --workload msg_c3 --set CHAIN_BWD_LOADER=0 &&
run c3_bwd1 --workload msg_c3 --set CHAIN_BWD_LOADER=1
