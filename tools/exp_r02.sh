mkdir -p gpurun_out/r02
run() { tag=$1; shift; python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/r02/e_$tag.json 2> gpurun_out/r02/e_$tag.err; python -c "
import json;d=json.load(open('gpurun_out/r02/e_$tag.json'));print('$tag',d['value'],d['ms_per_step'])"; }
run chain1
run chain0 --set USE_MLP_CHAIN=0
run chain0_nonarrow --set USE_MLP_CHAIN=0 --set ROWS_GEMM_NARROW=0
run chain0_nodx --set USE_MLP_CHAIN=0 --set ROWS_GEMM_DX=0
run chain0_nonarrow_nodx --set USE_MLP_CHAIN=0 --set ROWS_GEMM_NARROW=0 --set ROWS_GEMM_DX=0
