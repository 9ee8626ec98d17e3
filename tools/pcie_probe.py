#!/usr/bin/env python3
"""PCIe-inclusive cost of the host-buffer entry point lg_encode_commit for one workload:
    tools/pcie_probe.py <workload> <pageable|pinned|registered> <root|coeffs>
(one mode per process; the coefficient buffer is allocated and touched once, outside the timing)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ligero_amd
from bench import synthetic_preenc, WORKLOADS

wl, mem, what = sys.argv[1], sys.argv[2], sys.argv[3]
rows, k, batch = WORKLOADS[wl]
pre = synthetic_preenc(1000, batch * rows * k).reshape(batch * rows, k, 4)
c = ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch, device=0)


def alloc():
    if mem == "pinned":
        t = torch.empty(pre.shape, dtype=torch.int64).pin_memory()
        return t, t.numpy().view(np.uint64)
    a = np.zeros(pre.shape, dtype=np.uint64)
    if mem == "registered":
        c.host_register(a)
    return a, a


keep_in, buf = alloc()
buf[...] = pre
keep_out, out = alloc() if what == "coeffs" else (None, None)
for _ in range(2):
    c.encode_commit(buf, want_coeffs=out is not None, coeffs_out=out)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    _, root = c.encode_commit(buf, want_coeffs=out is not None, coeffs_out=out)
dt = (time.perf_counter() - t0) / n
print(f"{wl} {mem:8s} {what:6s} {dt*1e3:8.3f} ms per lg_encode_commit ({pre.nbytes/1e6:.0f} MB in{', same out' if out is not None else ''})  root {root[:4].hex()}")
