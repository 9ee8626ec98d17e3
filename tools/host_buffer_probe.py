import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, bench, ligero_amd
out = {}
for wl, reps in (("poseidon", 20), ("s20", 5)):
    rows, k, batch = bench.WORKLOADS[wl][0], bench.WORKLOADS[wl][1], (64 if wl == "poseidon" else 1)
    pre = bench.synthetic_preenc(3, batch * rows * k).reshape(-1, k, 4)
    r = bench.host_buffer_commit_ms(ligero_amd, pre, rows, k, batch, 0, reps)
    out[wl] = {a: round(b, 3) for a, b in r.items() if isinstance(b, float)}
print(json.dumps(out))
