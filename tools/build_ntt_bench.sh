#!/bin/bash
# usage: tools/build_ntt_bench.sh <name> <logk> [extra -D flags]; output build/ntt_bench_<name>
set -e
name=$1; logk=$2; shift 2
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ligero_amd/csrc -DLG_LOGK=$logk "$@" -o build/ntt_bench_$name tools/ntt_bench.hip
