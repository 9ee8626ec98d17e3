#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): regenerates what profiles/r03_* and profiles/pmc_traffic.json hold for the tree as shipped --
# for the three shapes (Poseidon x64, S20, S22): the bench line of the timed workload alone, rocprofv3 kernel stats of the same
# command, and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide prescribes); then the sharded legs on
# this one GPU (RCCL at world 1: all three modes; two gloo ranks sharing the GPU) and the host-buffer probe.
# Usage: tools/collect_profiles_r03.sh [tag]
set -u
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
for WL in poseidon s20 s22; do
  STEPS=20; [ "$WL" = s20 ] && STEPS=5; [ "$WL" = s22 ] && STEPS=3
  python3 bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_${WL}_bench.json" 2> "$OUT/${TAG}_${WL}_bench.err"
  echo "bench $WL done"
  D=/tmp/prof_${WL}_trace; rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline \
      > "$OUT/${TAG}_${WL}_bench_under_rocprof.json" 2> "$OUT/${TAG}_${WL}_rocprof.err"
  cp $(find $D -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_${WL}_kernel_stats.csv"
  echo "trace $WL done"
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/prof_${WL}_$C; rm -rf $D
    rocprofv3 --pmc $C --output-format csv -d $D -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline \
        > /dev/null 2> "$OUT/${TAG}_${WL}_pmc_${C}.err"
    lc=$(echo $C | tr A-Z a-z)
    f=$(find $D -name '*counter_collection.csv' | head -1)
    (head -1 "$f"; grep -E 'ntt_rows_kernel|blake2s_columns_kernel|merkle_subtree_kernel' "$f") > "$OUT/${TAG}_${WL}_pmc_${lc}.csv"
    echo "pmc $C $WL done"
  done
done
# one proof over the "ranks" of this box: RCCL itself at world 1 (all three modes), then two gloo ranks sharing the GPU
LIGERO_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload s22 --steps 5 --warmup 2 --no-cpu-baseline \
    2> "$OUT/${TAG}_s22_rccl_world1.err" | grep '^{' > "$OUT/${TAG}_s22_rccl_world1_three_modes.json"
echo "rccl world 1 done"
LIGERO_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --workload s22 --steps 3 --warmup 1 --no-cpu-baseline \
    2> "$OUT/${TAG}_s22_sharded_gloo2.err" | grep '^{' > "$OUT/${TAG}_s22_sharded_gloo2_bench.json"
echo "gloo 2 done"
for wl in poseidon s20; do python3 tools/witness_probe.py $wl 10 2>/dev/null | tail -1; done > "$OUT/${TAG}_from_witness_probe.log"
ls -la "$OUT"
