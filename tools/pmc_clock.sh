#!/bin/bash
# Effective shader clock under each commit kernel (MI355X guide, DVFS: clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration):
# tools/pmc_clock.sh [workload]   -- runs ON THE GPU BOX
export TMPDIR=/tmp
WL=${1:-poseidon}
D=/tmp/pmc_clk_$$; rm -rf $D
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $D -- python3 bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/pmc_clk_err.log || tail -3 /tmp/pmc_clk_err.log
f=$(find $D -name '*counter_collection.csv' | head -1)
[ -f "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", ", ".join(rows[0].keys()))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if not any(s in name for s in ("ntt_rows_kernel", "blake2s_columns", "merkle_subtree")):
        continue
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    t0, t1 = r.get("Start_Timestamp"), r.get("End_Timestamp")
    if t0 and t1 and float(t1) > float(t0):
        dur_ns = float(t1) - float(t0)
        acc[name].append((float(r["Counter_Value"]), dur_ns))
for name, v in sorted(acc.items()):
    v = v[len(v) // 3:]                      # skip warm-up dispatches
    cyc = sum(x for x, _ in v) / len(v)
    ns = sum(d for _, d in v) / len(v)
    print(f"  {name[:60]:60s} GRBM_GUI_ACTIVE {cyc:14.0f}  duration {ns/1e3:9.1f} us  effective clock {cyc / 8 / ns:6.3f} GHz  (n={len(v)})")
PY
