#!/bin/bash
# quick HBM-traffic check of one workload (two separate PMC passes): tools/pmc_quick.sh <workload>
export TMPDIR=/tmp
WL=$1
for C in FETCH_SIZE WRITE_SIZE; do
  D=/tmp/pq_$C; rm -rf $D
  rocprofv3 --pmc $C --output-format csv -d $D -- python3 bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  cp $(find $D -name '*counter_collection.csv' | head -1) /tmp/pq_$C.csv
done
python3 tools/pmc_traffic.py $WL /tmp/pq_FETCH_SIZE.csv /tmp/pq_WRITE_SIZE.csv
