#!/usr/bin/env python3
"""ms per resident commit for a range of k at about the Poseidon batch's volume (2.8 M message elements, >= 65 536 columns), for A/B builds:
    LIGERO_HIP_LIB=<lib.so> python tools/commit_time_by_k.py [logk ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ligero_amd

logks = [int(a) for a in sys.argv[1:]] or [4, 5, 6, 7, 8, 9, 10, 11, 12]
rng = np.random.default_rng(3)
out = []
for logk in logks:
    k = 1 << logk
    batch = max(1, 65536 // (8 * k))                     # at least 65 536 columns: the column hash is not a handful of serial chains
    rows = max(4, (2_800_000 // (k * batch)) // 4 * 4)
    pre = rng.integers(0, 2**62, size=(batch * rows, k, 4), dtype=np.uint64)
    pre[..., 3] &= np.uint64((1 << 60) - 1)
    with ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch) as c:
        c.upload(pre)
        for _ in range(5):
            c.commit_resident()
        c.sync()
        t0 = time.perf_counter()
        n = 40
        for _ in range(n):
            c.commit_resident()
        c.sync()
        out.append(f"k=2^{logk}: {1e3 * (time.perf_counter() - t0) / n:.3f}")
print(os.environ.get("LIGERO_HIP_LIB", "default")[-28:], " | ".join(out))
