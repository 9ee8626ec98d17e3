#!/bin/bash
# SQ counters per kernel: tools/pmc_ntt.sh <binary> [args]   (KERNEL=<substring> selects the kernel, default ntt_rows_kernel)
export TMPDIR=/tmp
B=$1; shift
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES SQ_INSTS_SALU"; do
  D=/tmp/pmc_$$; rm -rf $D
  rocprofv3 --pmc $set --output-format csv -d $D -- $B "$@" > /dev/null 2>&1
  f=$(find $D -name '*counter_collection.csv' | head -1)
  [ -f "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections, os
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if os.environ.get("KERNEL", "ntt_rows_kernel") in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k:24s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
done
