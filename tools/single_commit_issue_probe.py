"""Host issue time vs device time of a stream of single Poseidon commitments, and what the number depends on:
warm-up length, and whether a 64-proof batch context ran (and was closed) in the same process before (as in bench.py)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import ligero_amd
from bench import synthetic_preenc
rows, k = 344, 128
pre = synthetic_preenc(7, rows * k).reshape(rows, k, 4)


def run(warm, N, label):
    one = ligero_amd.LigeroCommitter(rows=rows, k=k, batch=1, device=0)
    one.upload(pre)
    for _ in range(warm):
        one.commit_resident()
    one.sync()
    t0 = time.perf_counter()
    for _ in range(N):
        one.commit_resident()
    t1 = time.perf_counter()
    one.sync()
    t2 = time.perf_counter()
    print(f"{label}: warm-up {warm}, N={N}: issue {1e3*(t1-t0)/N:.4f} ms/commit, total {1e3*(t2-t0)/N:.4f} ms/commit", flush=True)
    one.close()


run(20, 50, "fresh process")
run(5, 50, "fresh process")
big = synthetic_preenc(8, 64 * rows * k).reshape(64 * rows, k, 4)
c = ligero_amd.LigeroCommitter(rows=rows, k=k, batch=64, device=0)
c.upload(big)
for _ in range(200):
    c.commit_resident()
c.sync()
c.close()
run(5, 50, "after a closed batch-64 context")
run(20, 200, "after a closed batch-64 context")
