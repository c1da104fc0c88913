out=$1; rounds=${2:-2}
mkdir -p $out
for r in $(seq 1 $rounds); do
  for v in A B; do
    if [ $v = A ]; then export TP3D_REDUCE_FORWARD=1; else unset TP3D_REDUCE_FORWARD; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --details-out $out/detail_${v}_$r.json > $out/bench_${v}_$r.out 2> $out/bench_${v}_$r.err || exit 1
    echo "$v round $r: $(grep 'timed region' $out/bench_${v}_$r.err | sed 's/.*done: //')"
  done
done
