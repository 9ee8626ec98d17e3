"""Summarise a rocprofv3 --kernel-trace --output-format csv run: per kernel name count / total / mean ms inside a time window,
and the busy time of every (queue, stream).    python tools/timeline_summary.py <kernel_trace.csv> [t0_ms t1_ms]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1e18)
acc, streams = defaultdict(lambda: [0, 0.0, 0.0]), defaultdict(lambda: [1e18, 0.0, 0.0])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if s < lo or s > hi:
        continue
    name = r["Kernel_Name"].split("(")[0][-60:]
    a = acc[name]
    a[0] += 1; a[1] += e - s; a[2] = max(a[2], e - s)
    st = streams[(r["Queue_Id"], r["Stream_Id"])]
    st[0] = min(st[0], s); st[1] = max(st[1], e); st[2] += e - s
print(f"window [{lo:.1f}, {min(hi, max(v[1] for v in streams.values())):.1f}] ms")
for name, (n, tot, mx) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{tot:10.3f} ms  {n:5d} x  mean {tot / n:8.3f}  max {mx:8.3f}  {name}")
for (q, s), (a, b, busy) in sorted(streams.items()):
    print(f"queue {q} stream {s}: active [{a:.1f}, {b:.1f}] ms, busy {busy:.1f} ms")
