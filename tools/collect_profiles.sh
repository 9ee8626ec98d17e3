#!/bin/bash
# Runs ON THE GPU BOX (through gpurun) and regenerates everything under profiles/ that the bench
# line's roofline refers to.  Usage: tools/collect_profiles.sh <round tag, e.g. r01>
# Trace pass and the two PMC passes are separate rocprofv3 runs (MI355X guide, HBM section).
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
for WL in poseidon s20; do
  STEPS=20; [ "$WL" = s20 ] && STEPS=5
  python3 bench.py --workload $WL --steps $STEPS --warmup 3 > "$OUT/${TAG}_${WL}_bench.json" 2> "$OUT/${TAG}_${WL}_bench.err"
  D=/tmp/prof_${WL}_trace; rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --workload $WL --steps $STEPS --warmup 3 --no-cpu-baseline \
      > "$OUT/${TAG}_${WL}_bench_under_rocprof.json" 2> "$OUT/${TAG}_${WL}_rocprof.err"
  cp $(find $D -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_${WL}_kernel_stats.csv"
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/prof_${WL}_$C; rm -rf $D
    rocprofv3 --pmc $C --output-format csv -d $D -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline \
        > /dev/null 2> "$OUT/${TAG}_${WL}_pmc_${C}.err"
    lc=$(echo $C | tr A-Z a-z)
    # keep the per-dispatch rows of our kernels only (file is otherwise MBs of torch fill kernels)
    f=$(find $D -name '*counter_collection.csv' | head -1)
    (head -1 "$f"; grep -E 'ntt_rows_kernel|blake2s_columns_kernel|merkle_subtree_kernel' "$f") > "$OUT/${TAG}_${WL}_pmc_${lc}.csv"
  done
done
# constant-operand product microbenchmark (Montgomery vs Barrett with precomputed quotient) + its correctness leg
hipcc -O3 --offload-arch=gfx950 -I ligero_amd/csrc tools/microbench4.hip -o /tmp/microbench4 2>/dev/null \
  && python3 tools/mb4.py gen && /tmp/microbench4 > "$OUT/${TAG}_microbench4_shoup.log" 2>&1 && python3 tools/mb4.py check >> "$OUT/${TAG}_microbench4_shoup.log" 2>&1
# instruction-issue microbenchmark of the non-multiplier instructions, small-commit latency with both column-hash kernels
hipcc -O2 --offload-arch=gfx950 -o /tmp/microbench5 tools/microbench5.hip 2>/dev/null && /tmp/microbench5 > "$OUT/${TAG}_microbench5_instruction_issue.log" 2>&1
python3 tools/hash_latency_probe.py > "$OUT/${TAG}_hash_latency_quad_vs_single.log" 2>&1
# BASELINE configs[3] shape on one GPU, and the coset-sharded commit as a 2-rank dry run over gloo (both ranks on this box's one GPU)
python3 bench.py --workload s22 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_s22_bench.json" 2> "$OUT/${TAG}_s22_bench.err"
LIGERO_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --workload s22 --steps 3 --warmup 1 \
    2> "$OUT/${TAG}_s22_sharded_gloo2.err" | grep '^{' > "$OUT/${TAG}_s22_sharded_gloo2_bench.json"
# PCIe-inclusive host-buffer entry point
for m in pageable registered; do for w in root coeffs; do python3 tools/pcie_probe.py poseidon $m $w; python3 tools/pcie_probe.py s20 $m $w; done; done > "$OUT/${TAG}_pcie_inclusive.log" 2>/dev/null
ls -la "$OUT"
