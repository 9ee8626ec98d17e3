#!/usr/bin/env python3
"""Per-phase wall time of the 64-proof batch prover on the Poseidon fixtures (LG_PROVER_TIMING=1 -> stderr).
    python tools/batch_prove_timing.py [batch] [threads] [rounds]"""
import os, sys, time
os.environ["LG_PROVER_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ligero_amd import host_pipeline as hp
from ligero_amd.prover import LigeroBatchProver

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
G = os.path.join(ROOT, "tests", "golden")
inst = hp.LigeroInstance(hp.ArithmeticCircuit.from_r1cs(os.path.join(G, "poseidon.r1cs")))
w = hp.read_witness(os.path.join(G, "poseidon_witness.json"))
idx = list(range(1, w.shape[0]))
vals = np.broadcast_to(w[1:], (batch,) + w[1:].shape).copy()
with LigeroBatchProver(inst, batch, threads=threads) as bp:
    print("threads", bp.threads, file=sys.stderr)
    for i in range(rounds):
        t0 = time.perf_counter()
        bp.prove(idx, vals, copy=False)
        print(f"batch {i}: {1e3 * (time.perf_counter() - t0):.2f} ms for {batch} proofs", file=sys.stderr)
