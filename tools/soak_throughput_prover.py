import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ["LIGERO_NO_TORCH_PRELOAD"] = "1"
import numpy as np, bench, resource
from ligero_amd.prover import LigeroBatchProver, LigeroProver, proofs_equal
inst, idx, vals = bench.poseidon_batch_inputs()
batch = 1024
bp = LigeroBatchProver(inst, batch, device=0, threads=2, device_transcript=True)
single = LigeroProver(inst)
rng = np.random.default_rng(3)
n = 0; t0 = time.perf_counter(); mark = t0
sel = rng.integers(0, 64, size=batch); bp.submit(idx, np.ascontiguousarray(vals[sel])); prev = sel
while time.perf_counter() - t0 < float(sys.argv[1]):
    sel = rng.integers(0, 64, size=batch)
    bp.submit(idx, np.ascontiguousarray(vals[sel]))
    got = bp.collect(); n += batch
    if time.perf_counter() - mark > 20:
        b = int(rng.integers(batch))
        assert proofs_equal(single.prove(idx, vals[prev[b]]), got[b]) and single.verify(got[b])
        print(f"{n} proofs, {n / (time.perf_counter() - t0):.0f}/s, max rss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.2f} GB, spot check ok", flush=True)
        mark = time.perf_counter()
    prev = sel
bp.collect(); bp.close(); single.close()
print("done", n)
