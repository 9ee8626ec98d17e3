set -u
TAG=r02
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
for WL in poseidon s20; do
  STEPS=20; [ "$WL" = s20 ] && STEPS=5
  python3 bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_${WL}_bench.json" 2> "$OUT/${TAG}_${WL}_bench.err"
  echo "bench $WL done"
  D=/tmp/prof_${WL}_trace; rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline \
      > "$OUT/${TAG}_${WL}_bench_under_rocprof.json" 2> "$OUT/${TAG}_${WL}_rocprof.err"
  cp $(find $D -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_${WL}_kernel_stats.csv"
  echo "trace $WL done"
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/prof_${WL}_$C; rm -rf $D
    rocprofv3 --pmc $C --output-format csv -d $D -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline \
        > /dev/null 2> "$OUT/${TAG}_${WL}_pmc_${C}.err"
    lc=$(echo $C | tr A-Z a-z)
    f=$(find $D -name '*counter_collection.csv' | head -1)
    (head -1 "$f"; grep -E 'ntt_rows_kernel|blake2s_columns_kernel|merkle_subtree_kernel' "$f") > "$OUT/${TAG}_${WL}_pmc_${lc}.csv"
    echo "pmc $C $WL done"
  done
done
ls -la "$OUT"
