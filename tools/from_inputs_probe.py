import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, bench, ligero_amd
inst, idx, vals = bench.poseidon_batch_inputs()
left, right, consts = inst.gate_map()
prog = inst.trace_program(); pos = inst.input_positions(idx)
for batch in (1, 64, 1024):
    v = np.ascontiguousarray(vals[np.arange(batch) % 64])
    w = np.concatenate([inst.build_w(idx, v[b])[0] for b in range(min(batch, 64))])
    with ligero_amd.LigeroCommitter(rows=inst.rows, k=inst.k, batch=batch) as c:
        c.upload_gate_map(left, right, consts); c.upload_trace_program(prog)
        _, r0, ok = c.encode_commit_from_inputs(pos, v)
        t = time.perf_counter()
        for _ in range(20): c.encode_commit_from_inputs(pos, v)
        dt = (time.perf_counter() - t) / 20 * 1e3
        msg = f"batch {batch}: from_inputs {dt:.3f} ms"
        if batch <= 64:
            _, r1 = c.encode_commit_from_witness(w)
            t = time.perf_counter()
            for _ in range(20): c.encode_commit_from_witness(w)
            msg += f", from_witness {(time.perf_counter() - t) / 20 * 1e3:.3f} ms, roots equal {r0 == r1}"
        print(msg, "ok", bool(ok.all()))
