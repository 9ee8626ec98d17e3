"""Where the copy engine waits in the throughput prover: `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- python3
bench.py --prover-child 0 device 1024 6 0`, then `python tools/copy_gaps.py DIR`: every large device-to-host copy with the gap in front of
it and the kernel that ended last before it, then one steady-state batch in full.  Round 5: the copies are back to back (one 1.7 ms gap
per 96 ms batch, a gather the copy engine waits for); the 9.9 k proofs/s of a 6-batch run is fill and drain -- 12 batches give 10.2 k."""
import csv, sys, glob
d = sys.argv[1]
mc = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
copies = []
for r in csv.DictReader(open(mc)):
    if "DEVICE_TO_HOST" in r.get("Direction", r.get("Kind", "")):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
copies.sort()
t0 = copies[0][0]
kern = []
for r in csv.DictReader(open(kt)):
    kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
kern.sort()
print("copies:", len(copies))
prev_end = None
for s, e in copies:
    dur = (e - s) / 1e6
    if dur < 5:
        continue
    gap = (s - prev_end) / 1e6 if prev_end else 0
    # what kernel ended last before this copy started
    last = max((k for k in kern if k[1] <= s), key=lambda k: k[1], default=None)
    print("copy start %8.2f ms dur %6.2f gap-before %6.2f  last kernel before: %s ended %.2f ms earlier" % ((s - t0) / 1e6, dur, gap, last[2] if last else "-", (s - last[1]) / 1e6 if last else 0))
    prev_end = e
# one steady-state batch in detail: everything longer than 0.25 ms between the 12th and the 16th large copy
big = [c for c in copies if (c[1] - c[0]) > 5e6]
if len(big) >= 16:
    w0, w1 = big[11][0], big[15][1]
    ev = [(s, e, "COPY D2H") for s, e in big if s >= w0 and e <= w1] + [(s, e, n) for s, e, n in kern if s >= w0 and e <= w1 and e - s > 250000]
    ev.sort()
    print("---- window")
    for s, e, n in ev:
        print("%9.2f .. %9.2f  (%6.2f ms)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
