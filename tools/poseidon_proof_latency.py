"""One Poseidon-R1CS proof at a time on the single prover (BASELINE configs[1] as a latency): ms per prove(), ms per verify()."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ligero_amd.prover import LigeroProver  # noqa: E402

inst, idx, vals = bench.poseidon_batch_inputs()
with LigeroProver(inst) as p:
    for i in range(3):
        proof = p.prove(idx, vals[i])
    t = time.perf_counter()
    for i in range(50):
        proof = p.prove(idx, vals[i % 64])
    dt = (time.perf_counter() - t) / 50
    t = time.perf_counter()
    for i in range(20):
        ok = p.verify(proof)
    dv = (time.perf_counter() - t) / 20
    print(f"Poseidon proof: {dt * 1e3:.3f} ms per prove() ({1 / dt:.0f} proofs/s one at a time), {dv * 1e3:.3f} ms per verify() -> {ok}")
