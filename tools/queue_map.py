"""Which hardware queue did each kernel of a rocprofv3 kernel trace run on?  Counts per (queue id, kernel) over the last `tail`
dispatches (the single-commitment loop of tools/stream_order_probe.py's child).  python tools/queue_map.py <kernel_trace.csv> [tail]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-tail:]
cnt = collections.Counter()
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("lg::", "").replace("void ", "")[:48]
    cnt[(r["Queue_Id"], name)] += 1
for (q, name), n in sorted(cnt.items()):
    print(f"queue {q:>3}  {name:<50} {n}")
