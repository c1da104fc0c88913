# loader-mode layer chain (split-role GEMM with the previous layer's BatchNorm + activation in its loader waves) against the
# layer-wise path: headline step, forward only, config 3
mkdir -p gpurun_out/r02n
cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift; timeout -k 10 280 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r02n/$tag.json 2> gpurun_out/r02n/$tag.err; python -c "
import json,sys;d=json.load(open('gpurun_out/r02n/$tag.json'));print('$tag',d['value'],d['ms_per_step'],(d.get('forward_only') or {}).get('ms_per_step'))"; }
run chain0 --set USE_MLP_CHAIN=0 &&
run chain1 --set USE_MLP_CHAIN=1 &&
run c3_chain0 --workload msg_c3 --set USE_MLP_CHAIN=0 &&
run c3_chain1 --workload msg_c3 --set USE_MLP_CHAIN=1
