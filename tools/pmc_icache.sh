#!/bin/bash
# Instruction-cache counters of the commit kernels (the k = 128 evaluate kernel is 58 KB of straight-line code):
# tools/pmc_icache.sh [workload]   -- runs ON THE GPU BOX; one rocprofv3 --pmc pass per counter set
export TMPDIR=/tmp
WL=${1:-poseidon}
rocprofv3 -L 2>/dev/null | grep -i -o -E "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_CACHE|INSTS_SMEM)[A-Z0-9_]*)" | sort -u | tr '\n' ' '; echo
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  D=/tmp/pmc_ic_$$; rm -rf $D
  rocprofv3 --pmc $set --output-format csv -d $D -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pmc_ic_err.log || tail -3 /tmp/pmc_ic_err.log
  f=$(find $D -name '*counter_collection.csv' | head -1)
  [ -f "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    for kname in ("ntt_rows_kernel<7, 0, true>", "ntt_rows_kernel<7, 0, false>", "ntt_rows_kernel<12, 0, true>", "blake2s_columns_kernel"):
        if kname in r["Kernel_Name"]:
            acc[(kname, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (kn, c), v in sorted(acc.items()):
    print(f"  {kn:34s} {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
done
