# A/B of bench.py switches inside ONE gpurun call (boxes differ by a few per cent): tools/ab.sh OUTDIR "<args A>" "<args B>" [rounds]
out=$1; a=$2; b=$3; rounds=${4:-2}
mkdir -p $out
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = A ]; then args=$a; else args=$b; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --details-out $out/detail_${v}_$r.json $args > $out/bench_${v}_$r.out 2> $out/bench_${v}_$r.err || exit 1
    echo "$v round $r [$args]: $(grep 'timed region' $out/bench_${v}_$r.err | sed 's/.*done: //')"
  done
done
