#!/bin/bash
# EXPERIMENTS.md section I: what could an evaluate + column-hash fusion save at k = 128?  Its one benefit is that
# the hash no longer reads U back from HBM (723 MB of the 1.82 GB a Poseidon step moves).  Upper bound of that benefit, without any
# of a fusion's costs: the shipped pipeline with the hash kernel's row reads ALIASED onto each column's first row
# (-DLG_AB_HASH_ALIASED_ROWS: same instruction stream, reads served from cache; digests are wrong, so roots differ).  Same box,
# alternating, two repetitions.     Build first (CPU box):  bash tools/ab_hash_reads.sh build      then on the GPU box:  bash tools/ab_hash_reads.sh
set -e
cd "$(dirname "$0")/.."
if [ "${1:-}" = build ]; then
  make -C ligero_amd/csrc -j8 BUILD=../../build/ab_alias LIB=../lib/ab_hash_aliased.so \
       HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wno-unused-value -DLG_AB_HASH_ALIASED_ROWS"
  exit 0
fi
for w in poseidon s20; do
  st=20; [ $w = poseidon ] && st=300
  bash tools/ab_bench.sh $w $st ligero_amd/lib/libligero_hip.so ligero_amd/lib/ab_hash_aliased.so
done
