#!/usr/bin/env python3
"""Latency of small commits with the two Blake2s column-hash kernels (one lane per column / four lanes per column):
    python tools/hash_latency_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ligero_amd  # noqa: E402
from bench import synthetic_preenc  # noqa: E402

pre = synthetic_preenc(5, 344 * 128).reshape(344, 128, 4)
for q in ("0", "1000000000"):
    os.environ["LG_HASH_QUAD_MAX_COLUMNS"] = q
    for batch in (1, 2, 4, 8, 16, 32):
        c = ligero_amd.LigeroCommitter(rows=344, k=128, batch=batch)
        c.upload(np.tile(pre, (batch, 1, 1)))
        for _ in range(10):
            c.commit_resident()
        c.sync()
        t = time.perf_counter()
        for _ in range(100):
            c.commit_resident()
        c.sync()
        dt = (time.perf_counter() - t) / 100 * 1e3
        c.profile(True)
        for _ in range(20):
            c.commit_resident()
        c.sync()
        st = c.stage_ms()
        print("quad_max=%s batch=%d columns=%d: %.3f ms/commit  hash %.3f eval %.3f interp %.3f merkle %.3f" %
              (q, batch, batch * 1024, dt, st["colhash"], st["evaluate"], st["interpolate"], st["merkle"]), flush=True)
        c.close()
