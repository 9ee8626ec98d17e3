#!/usr/bin/env python3
"""Determinism soak of the commit pipeline: thousands of back-to-back resident commits alternating between two inputs, the roots
read back only now and then (so the asynchronous hash / tree overlap runs unthrottled in between) and compared with the roots of
the same inputs committed alone.   python tools/soak.py [seconds] [workload: poseidon|single|s20|witness|witness-s20]
"witness": the matrices have the structure of preenc_u under a random wiring (X, Y, Z gathered from W) and, between the bursts of
resident commits, each context also commits from `w` alone (lg_encode_commit_from_witness: stepped upload, gathers, merged launches)
-- whose root must be the same root, and which must leave the resident matrix whole for the resident commits that follow."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import ligero_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
wl = sys.argv[2] if len(sys.argv) > 2 else "poseidon"
rows, k, batch = {"poseidon": (344, 128, 64), "single": (344, 128, 1), "s20": (10036, 4096, 1), "witness": (344, 128, 64),
                  "witness-s20": (10036, 4096, 1)}[wl]   # "single": the three-deep ring
wired = wl.startswith("witness")
rng = np.random.default_rng(99)
m = rows // 4
# a random wiring: a third of the positions are gates reading two earlier or later positions
left = np.full(m * k, 0xffffffff, dtype=np.uint32)
right = left.copy()
gates = np.flatnonzero(rng.integers(3, size=m * k) == 0)
left[gates] = rng.integers(m * k, size=gates.size)
right[gates] = rng.integers(m * k, size=gates.size)
ws = []


def rand():
    a = rng.integers(0, 2**62, size=(batch * rows, k, 4), dtype=np.uint64)
    a[..., 3] &= np.uint64((1 << 60) - 1)
    if wired:                                      # [X; Y; Z; W] of every proof from its W block (mod.rs:483-516)
        a = a.reshape(batch, 4, m * k, 4)
        a[:, :3] = 0
        w = a[:, 3]
        a[:, 0][:, gates], a[:, 1][:, gates], a[:, 2][:, gates] = w[:, left[gates]], w[:, right[gates]], w[:, gates]
        ws.append(np.ascontiguousarray(w).reshape(batch * m, k, 4))
        a = a.reshape(batch * rows, k, 4)
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
    if wired:
        for x in ctxs:
            x.upload_gate_map(left, right, np.zeros((0, 4), dtype=np.uint64))
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
        if wired and rng.integers(2) == 0:         # the same matrices from w alone, then resident commits again
            for w in (0, 1):
                assert ctxs[w].encode_commit_from_witness(ws[w])[1] == want[w], ("root from w differs", n, w)
            n += 2
            checks += 2
        if time.time() - t_mark > 30.0:            # keep writing: the GPU box takes 7 silent minutes for a hang
            t_mark = time.time()
            print(f"  ... {n} commits, {checks} checks", flush=True)
        if rng.integers(8) == 0:                   # swap the resident inputs: uploads race with nothing they should not
            inputs.reverse(); want.reverse(); ws.reverse()
            ctxs[0].upload(inputs[0]); ctxs[1].upload(inputs[1])
print(f"soak {wl}: {n} commits on two interleaved contexts, {checks} root checks, all equal")
