#!/usr/bin/env python3
"""lg_encode_commit_from_witness timing probe: tools/witness_probe.py poseidon|s20 [reps]
synthetic backward-only gate map (half of the positions are Mul gates, as in compiled R1CS circuits), random w, page-locked.
LG_WITNESS_STEPS / LG_WITNESS_TAIL change the step plan (read once per process)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import ligero_amd  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "poseidon"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows, k, batch = bench.WORKLOADS[wl]
m = rows // 4
mk = m * k
rng = np.random.default_rng(3)
left = np.full(mk, 0xffffffff, dtype=np.uint32)
right = left.copy()
gates = np.arange(1, mk, 2)
left[gates] = (rng.random(gates.shape[0]) * gates).astype(np.uint32)
right[gates] = (rng.random(gates.shape[0]) * gates).astype(np.uint32)
w = bench.synthetic_preenc(7, batch * mk).reshape(batch * m, k, 4)
c = ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch)
c.upload_gate_map(left, right, np.zeros((0, 4), dtype=np.uint64))
c.host_register(w)
c.encode_commit_from_witness(w)
t0 = time.perf_counter()
for _ in range(reps):
    _, root = c.encode_commit_from_witness(w)
dt = (time.perf_counter() - t0) / reps * 1e3
# the same matrix resident: what the device alone needs
c.commit_resident()
c.sync()
t0 = time.perf_counter()
for _ in range(reps):
    c.commit_resident()
    r2 = c.root()
dr = (time.perf_counter() - t0) / reps * 1e3
print(f"{wl}: from_witness {dt:.3f} ms | resident commit + root read {dr:.3f} ms | roots equal {root == r2} | "
      f"steps={os.environ.get('LG_WITNESS_STEPS', 'auto')} tail={os.environ.get('LG_WITNESS_TAIL', 'auto')}")
c.host_unregister(w)
c.close()
