#!/usr/bin/env python3
"""cost of lg_open_columns_batch on the Poseidon batch (64 x 156 columns x 344 rows = 110 MB out):
fresh output arrays / reused pageable / reused page-locked"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ligero_amd
from bench import synthetic_preenc, WORKLOADS
rows, k, batch = WORKLOADS["poseidon"]
t = 156
pre = synthetic_preenc(1000, batch * rows * k).reshape(batch * rows, k, 4)
c = ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch, device=0)
c.encode_commit(pre, want_coeffs=False)
rng = np.random.default_rng(5)
idx = np.stack([np.sort(rng.choice(8 * k, size=t, replace=False)) for _ in range(batch)]).astype(np.uint32)
def timeit(fn, n=10):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e3
print("fresh outputs     %.3f ms" % timeit(lambda: c.open_columns_batch(idx)))
out = c.open_columns_batch(idx)
print("reused pageable   %.3f ms" % timeit(lambda: c.open_columns_batch(idx, out=out)))
for a in out: c.host_register(a)
print("reused registered %.3f ms" % timeit(lambda: c.open_columns_batch(idx, out=out)))
print("bytes out: %.1f MB" % (sum(a.nbytes for a in out) / 1e6))
