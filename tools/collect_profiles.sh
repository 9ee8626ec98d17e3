#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): regenerates what profiles/<tag>_* and profiles/pmc_traffic.json hold for the tree as shipped.
#   1. the three commit shapes (Poseidon x64, S20, S22): the bench line of the timed workload alone, rocprofv3 kernel stats of the
#      same command, and the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide prescribes)
#   2. the proofs/s leg (device-transcript prover, two batches of 1024 in flight) on the system HIP runtime: kernel stats + memory
#      copies of the same child command bench.py runs, and tools/device_transcript_probe.py's table
#   3. one proof over the "ranks" of this box: RCCL itself at world 1 (all four modes), two gloo ranks sharing the GPU
#   4. the default bench line (what the driver runs)
#   5. the large proofs (2^20 / 2^22 constraints: phases, kernels, three provers in flight), the commit from the assignment, the stream
#      placement A/B and the host's field-arithmetic microbenchmarks
# Usage: tools/collect_profiles.sh <round tag, e.g. r05> [commit|prover|throughput|all]      (copy gpurun_out/profiles/* into profiles/ afterwards;
#        profiles/pmc_traffic.json is rewritten in place by tools/pmc_traffic.py -- set its _source.commit when committing).
#        The whole collection outlasts one gpurun call (20 min): `commit` = steps 1 and the ISA counts, `prover` = steps 2-5,
#        `throughput` = step 2 alone.
set -u
TAG=${1:-r05}
PART=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$PART" = all ] || [ "$PART" = commit ]; then
for WL in poseidon s20 s22; do
  STEPS=200; WARM=10                        # Poseidon: bench.py's defaults, so the profiler's average is of the driver's command
  [ "$WL" = s20 ] && STEPS=5 && WARM=3; [ "$WL" = s22 ] && STEPS=3 && WARM=3
  python3 bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline > "$OUT/${TAG}_${WL}_bench.json" 2> "$OUT/${TAG}_${WL}_bench.err"
  echo "bench $WL done"
  D=/tmp/prof_${WL}_trace; rm -rf $D
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --workload $WL --steps $STEPS --warmup $WARM --no-cpu-baseline \
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
    cp "$f" /tmp/pq_${WL}_$C.csv
    echo "pmc $C $WL done"
  done
  python3 tools/pmc_traffic.py $WL /tmp/pq_${WL}_FETCH_SIZE.csv /tmp/pq_${WL}_WRITE_SIZE.csv > /dev/null
done
cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json"
# the numerator of bench.py's valu_roofline: dynamic instruction counts from the ISA of THIS tree (profiles/isa_counts.json), with the
# hardware's own count of VALU instructions per wave beside them (one PMC pass per shape; counters in a run of their own)
python3 tools/isa_counts.py > "$OUT/${TAG}_isa_counts.log" 2>&1
PMCARGS=""
for WL in poseidon s20 s22; do
  D=/tmp/prof_${WL}_insts; rm -rf $D
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $D -- python3 bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline \
      > /dev/null 2> "$OUT/${TAG}_${WL}_pmc_insts.err"
  f=$(find $D -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && (head -1 "$f"; grep -E 'ntt_rows_kernel' "$f") > "$OUT/${TAG}_${WL}_pmc_insts.csv" && PMCARGS="$PMCARGS $WL=$OUT/${TAG}_${WL}_pmc_insts.csv"
  echo "pmc insts $WL done"
done
python3 tools/isa_counts.py --pmc $PMCARGS >> "$OUT/${TAG}_isa_counts.log" 2>&1
cp profiles/isa_counts.json "$OUT/isa_counts.json"
fi
if [ "$PART" = commit ]; then ls -la "$OUT"; exit 0; fi
# the proofs/s leg: the child bench.py runs, under the profiler (the program itself after `--`, no torch in it)
D=/tmp/prof_prover; rm -rf $D
LIGERO_NO_TORCH_PRELOAD=1 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $D -- python3 bench.py --prover-child 0 device 1024 6 0 \
    > "$OUT/${TAG}_prover_child_under_rocprof.json" 2> "$OUT/${TAG}_prover_rocprof.err"
cp $(find $D -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_prover_kernel_stats.csv"
cp $(find $D -name '*memory_copy_stats.csv' | head -1) "$OUT/${TAG}_prover_memory_copy_stats.csv"
python3 tools/timeline_summary.py $(find $D -name '*kernel_trace.csv' | head -1) > "$OUT/${TAG}_prover_timeline_summary.log" 2>&1
python3 tools/copy_gaps.py $D > "$OUT/${TAG}_prover_copy_gaps.log" 2>&1
# the same child in RESIDENT mode (lg_prover_set_resident: the openings stay on the device, their digests come home): the sponge chain,
# the gathers and the digest kernels on the critical path; then how batch depth and several contexts change it
D=/tmp/prof_prover_res; rm -rf $D
LIGERO_NO_TORCH_PRELOAD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 bench.py --prover-child 0 resident 2048 4 0 \
    > "$OUT/${TAG}_prover_child_resident_under_rocprof.json" 2> "$OUT/${TAG}_prover_resident_rocprof.err"
cp $(find $D -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_prover_resident_kernel_stats.csv"
LIGERO_NO_TORCH_PRELOAD=1 python3 tools/device_transcript_probe.py pipe:1x1024 pipe:1x2048 pipe:1x4096 pipe:2x512 pipe:2x1024 --steps=6 --resident 2>&1 | grep -v amdgpu > "$OUT/${TAG}_resident_contexts_probe.log"
LIGERO_NO_TORCH_PRELOAD=1 python3 tools/device_transcript_probe.py host:4x64 1x64 1x256 1x1024 pipe:1x256 pipe:1x512 pipe:1x1024 pipe:1x2048 2x256 2x512 --steps=8 \
    2>&1 | grep -v amdgpu > "$OUT/${TAG}_device_transcript_probe.log"
LIGERO_NO_TORCH_PRELOAD=1 python3 tools/device_transcript_probe.py pipe:1x1024 --steps=8 --cpus=2 2>&1 | grep proofs/s >> "$OUT/${TAG}_device_transcript_probe.log"
echo "(copies by shader kernels instead of the SDMA engines: they stretch the HBM-bound kernels beside them, EXPERIMENTS K)" >> "$OUT/${TAG}_device_transcript_probe.log"
LIGERO_NO_TORCH_PRELOAD=1 LG_SHIP_BLOCKS=8 python3 tools/device_transcript_probe.py pipe:1x1024 --steps=8 2>&1 | grep proofs/s | sed 's/$/   [ship_kernel, 8 workgroups]/' >> "$OUT/${TAG}_device_transcript_probe.log"
LIGERO_NO_TORCH_PRELOAD=1 LG_COPY_STREAM_PRIORITY=none python3 tools/device_transcript_probe.py pipe:1x1024 pipe:1x1024 --steps=8 2>&1 | grep proofs/s | sed 's/$/   [copy stream at the encode stream'"'"'s priority: the second prover of a process may share its hardware queue, EXPERIMENTS L]/' >> "$OUT/${TAG}_device_transcript_probe.log"
echo "prover done"
if [ "$PART" = throughput ]; then ls -la "$OUT"; exit 0; fi
python3 tools/s20_prove_timing.py 20 6 2>&1 | grep -v amdgpu > "$OUT/${TAG}_s20_prove_timing.log"
timeout -k 10 600 python3 tools/s20_prove_timing.py 22 3 2>&1 | grep -v amdgpu > "$OUT/${TAG}_s22_prove_timing.log"
python3 tools/large_proofs_in_flight.py 20 3 6 2>&1 | grep -v amdgpu > "$OUT/${TAG}_large_proofs_in_flight.log"
python3 tools/poseidon_proof_latency.py 2>&1 | grep Poseidon > "$OUT/${TAG}_poseidon_proof_latency.log"
python3 tools/from_inputs_probe.py 2>&1 | grep batch > "$OUT/${TAG}_from_inputs_probe.log"
python3 tools/stream_order_probe.py 0 1 2 3 5 > "$OUT/${TAG}_stream_pick_ab.log" 2>&1
g++ -O2 -std=c++17 -I ligero_amd/csrc -o /tmp/host_mul_bench tools/host_mul_bench.cpp && /tmp/host_mul_bench > "$OUT/${TAG}_host_mul_bench.log"
(g++ -O2 -std=c++17 -I ligero_amd/csrc -o /tmp/host_sbox_bench tools/host_sbox_bench.cpp && /tmp/host_sbox_bench; g++ -O2 -std=c++17 -I ligero_amd/host -I ligero_amd/csrc -I include -o /tmp/host_sponge_bench tools/host_sponge_bench.cpp && /tmp/host_sponge_bench) > "$OUT/${TAG}_host_sbox_bench.log"
D=/tmp/prof_s20_proof; rm -rf $D      # the kernels of one large proof: trace, gathers, commit, the three sub-proofs
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/s20_prove_timing.py 20 4 > /dev/null 2>&1
cp $(find $D -name '*kernel_stats.csv' | head -1) "$OUT/${TAG}_s20_proof_kernel_stats.csv"
python3 tools/timeline_summary.py $(find $D -name '*kernel_trace.csv' | head -1) > "$OUT/${TAG}_s20_proof_timeline_summary.log" 2>&1
# one proof over the "ranks" of this box
LIGERO_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload s22 --steps 5 --warmup 2 --no-cpu-baseline \
    2> "$OUT/${TAG}_s22_rccl_world1.err" | grep '^{' > "$OUT/${TAG}_s22_rccl_world1_four_modes.json"
echo "rccl world 1 done"
LIGERO_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 50 --warmup 5 --sharded-leg s20 \
    2> "$OUT/${TAG}_poseidon_gloo2.err" | grep '^{' > "$OUT/${TAG}_poseidon_gloo2_default_with_sharded_legs.json"
echo "gloo 2 done"
python3 bench.py > "$OUT/${TAG}_poseidon_bench_default_run.json" 2> "$OUT/${TAG}_poseidon_bench_default_run.err"
rm -f "$OUT"/*.err.empty; find "$OUT" -name '*.err' -size 0 -delete
ls -la "$OUT"
