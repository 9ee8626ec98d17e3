"""Latency of the sub-proof polynomial calls (host buffers in, coefficients out) on one proof."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import ligero_amd, bench
for name, (rows, k, batch) in {"poseidon": (344, 128, 1), "poseidon x64": (344, 128, 64), "s20": (10036, 4096, 1)}.items():
    c = ligero_amd.LigeroCommitter(rows=rows, k=k, batch=batch)
    pre = bench.synthetic_preenc(1, batch * rows * k).reshape(batch * rows, k, 4)
    c.encode_commit(pre, want_coeffs=False)
    r_int = bench.synthetic_preenc(2, batch * rows)
    r_a = bench.synthetic_preenc(3, batch * rows * k).reshape(batch * rows, k, 4)
    r_q = bench.synthetic_preenc(4, batch * (rows // 4))
    for fn, arg in (("interleaved_row_mul", r_int), ("linear_constraint_poly", r_a), ("quadratic_constraint_poly", r_q)):
        getattr(c, fn)(arg)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            getattr(c, fn)(arg)
        print(f"{name:12s} {fn:28s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms per call (incl. H2D of the challenge and D2H of the result)")
    idx = list(range(0, 8 * k, 8 * k // 156))[:156]
    c.open_columns(idx)
    t0 = time.perf_counter()
    for _ in range(5):
        c.open_columns(idx)
    print(f"{name:12s} {'open_columns(t=156), 1 proof':28s} {(time.perf_counter() - t0) / 5 * 1e3:8.3f} ms per call")
    if batch > 1:
        bidx = np.tile(np.asarray(idx, dtype=np.uint32), (batch, 1))
        c.open_columns_batch(bidx)
        t0 = time.perf_counter()
        for _ in range(5):
            c.open_columns_batch(bidx)
        print(f"{name:12s} {'open_columns_batch(t=156)':28s} {(time.perf_counter() - t0) / 5 * 1e3:8.3f} ms per call (all {batch} proofs)")
    c.close()
