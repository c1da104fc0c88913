run() { timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | grep "timed region" | sed 's/.*done: //'; }
build() { python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
echo "base: $(run)"
sed -i 's/constexpr int ST_ROWS = 256; /constexpr int ST_ROWS = 512; /' torch_points3d_amd/csrc/rows.hip; build; echo "ST_ROWS 512: $(run)"
sed -i 's/constexpr int ST_ROWS = 512; /constexpr int ST_ROWS = 1024;/' torch_points3d_amd/csrc/rows.hip; build; echo "ST_ROWS 1024: $(run)"
sed -i 's/constexpr int ST_ROWS = 1024;/constexpr int ST_ROWS = 256; /' torch_points3d_amd/csrc/rows.hip
sed -i 's/int64_t rps = (M + 1023) \/ 1024;/int64_t rps = (M + 2047) \/ 2048;/' torch_points3d_amd/csrc/gemm_tn_narrow.hip; build; echo "narrow splits 2048: $(run)"
sed -i 's/int64_t rps = (M + 2047) \/ 2048;/int64_t rps = (M + 1023) \/ 1024;/' torch_points3d_amd/csrc/gemm_tn_narrow.hip
sed -i 's/return (side \&\& items >= 2048) ? 1024 : 512; }/return (side \&\& items >= 2048) ? 2048 : 512; }/' torch_points3d_amd/csrc/gemm_rows_sp.hip; build; echo "sp grid 2048: $(run)"
echo "base again: $(sed -i 's/? 2048 : 512; }/? 1024 : 512; }/' torch_points3d_amd/csrc/gemm_rows_sp.hip; build; run)"
