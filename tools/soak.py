#!/usr/bin/env python3
"""Determinism soak of the commit pipeline: thousands of back-to-back resident commits alternating between two inputs, the roots
read back only now and then (so the asynchronous hash / tree overlap runs unthrottled in between) and compared with the roots of
the same inputs committed alone.   python tools/soak.py [seconds] [workload: poseidon|single|s20]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ligero_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
wl = sys.argv[2] if len(sys.argv) > 2 else "poseidon"
rows, k, batch = {"poseidon": (344, 128, 64), "single": (344, 128, 1), "s20": (10036, 4096, 1)}[wl]   # "single": the three-deep ring
rng = np.random.default_rng(99)


def rand():
    a = rng.integers(0, 2**62, size=(batch * rows, k, 4), dtype=np.uint64)
    a[..., 3] &= np.uint64((1 << 60) - 1)
    return a


inputs = [rand(), rand()]
want = []
for x in inputs:
    with ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch) as c:
        want.append(c.encode_commit(x, want_coeffs=False)[1])
assert want[0] != want[1]
n = checks = 0
t_end = time.time() + budget
t_mark = time.time()
with ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch) as c, ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch) as d:
    ctxs = [c, d]
    ctxs[0].upload(inputs[0])
    ctxs[1].upload(inputs[1])
    while time.time() < t_end:
        burst = int(rng.integers(1, 40))
        which = int(rng.integers(2))
        for _ in range(burst):                     # two contexts interleaved on the same device: their streams overlap freely
            ctxs[which].commit_resident()
            ctxs[1 - which].commit_resident()
            n += 2
        for w in (0, 1):
            assert ctxs[w].root() == want[w], ("root changed", n, w)
        checks += 2
        if time.time() - t_mark > 30.0:            # keep writing: the GPU box takes 7 silent minutes for a hang
            t_mark = time.time()
            print(f"  ... {n} commits, {checks} checks", flush=True)
        if rng.integers(8) == 0:                   # swap the resident inputs: uploads race with nothing they should not
            inputs.reverse(); want.reverse()
            ctxs[0].upload(inputs[0]); ctxs[1].upload(inputs[1])
print(f"soak {wl}: {n} commits on two interleaved contexts, {checks} root checks, all equal")
