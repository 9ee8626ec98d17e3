import sys, time, numpy as np
sys.path.insert(0, '.')
import ligero_amd
from bench import synthetic_preenc
rows, k = 344, 128
pre = synthetic_preenc(7, rows * k).reshape(rows, k, 4)
one = ligero_amd.LigeroCommitter(rows=rows, k=k, batch=1, device=0)
one.upload(pre)
for _ in range(20): one.commit_resident()
one.sync()
for N in (50, 200):
    t0 = time.perf_counter()
    for _ in range(N): one.commit_resident()
    t1 = time.perf_counter()
    one.sync()
    t2 = time.perf_counter()
    print(f"N={N}: issue {1e3*(t1-t0)/N:.4f} ms/commit, total {1e3*(t2-t0)/N:.4f} ms/commit")
one.close()
