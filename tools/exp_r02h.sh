# forward contractions of the layer chain as bf16 term pairs on the matrix pipe (CHAIN_BF16_TERMS), on / off
mkdir -p gpurun_out/r02n
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_gpu_fused.py -m gpu -q -k "bf16_terms or split_role_gemm_applies" 2>&1 | tail -2
run() { tag=$1; shift; timeout -k 10 280 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r02n/$tag.json 2> gpurun_out/r02n/$tag.err; python -c "
import json,sys;d=json.load(open('gpurun_out/r02n/$tag.json'));print('$tag',d['value'],d['ms_per_step'],(d.get('forward_only') or {}).get('ms_per_step'))"; }
run b3_0 --set CHAIN_BF16_TERMS=0 &&
run b3_1 --set CHAIN_BF16_TERMS=1 &&
run c3_b3_1 --workload msg_c3 --set CHAIN_BF16_TERMS=1
