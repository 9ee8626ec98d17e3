#!/bin/bash
# A/B timing of library builds on ONE box (boxes differ by +-10%): tools/ab_bench.sh <workload> <steps> <lib>...
WL=$1; STEPS=$2; shift 2
for rep in 1 2; do
  for L in "$@"; do
    LIGERO_HIP_LIB=$(realpath $L) python3 bench.py --no-cpu-baseline --workload $WL --steps $STEPS | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms']
print('%-60s %8.3f ms/step  interp %.3f eval %.3f hash %.3f merkle %.3f  root %s' % ('$L'[-60:], d['ms_per_step'], s['interpolate'], s['evaluate'], s['colhash'], s['merkle'], d['root0'][:8]))"
  done
done
