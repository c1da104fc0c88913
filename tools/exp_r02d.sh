# does the split-role kernel lose its gain to the co-running geometry stream?  chain on/off without the geometry prefetch
mkdir -p gpurun_out/r02n
cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift; timeout -k 10 280 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r02n/$tag.json 2> gpurun_out/r02n/$tag.err; python -c "
import json,sys;d=json.load(open('gpurun_out/r02n/$tag.json'));print('$tag',d['value'],d['ms_per_step'],(d.get('forward_only') or {}).get('ms_per_step'))"; }
run nopf_chain0 --no-geometry-prefetch --set USE_MLP_CHAIN=0 &&
run nopf_chain1 --no-geometry-prefetch --set USE_MLP_CHAIN=1 &&
timeout -k 10 100 python -m pytest tests/test_gpu_fused.py -m gpu -q -k "constants_table" 2>&1 | tail -2
