#!/usr/bin/env python3
"""proofs/s of the batched prover on the Poseidon fixture (64 distinct witnesses per call)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from ligero_amd import host_pipeline as hp
from ligero_amd.prover import LigeroBatchProver
G = os.path.join(ROOT, "tests", "golden")
inst = hp.LigeroInstance(hp.ArithmeticCircuit.from_r1cs(os.path.join(G, "poseidon.r1cs")))
blob = open(os.path.join(G, "poseidon_witness_batch64.bin"), "rb").read()
P, MASK = 21888242871839275222246405745257275088548364400416034343698204186575808495617, (1 << 64) - 1
allv = np.empty((64, 264, 4), dtype=np.uint64)
for i in range(64):
    for j in range(1, 265):
        v = (int.from_bytes(blob[(i * 265 + j) * 32:(i * 265 + j + 1) * 32], "little") << 256) % P      # Montgomery form
        allv[i, j - 1] = [(v >> (64 * l)) & MASK for l in range(4)]
idx = list(range(1, 265))
for threads in ([int(a) for a in sys.argv[1:]] or [1, 8, 16, 32, 64]):
    with LigeroBatchProver(inst, 64, threads=threads) as bp:
        bp.prove(idx, allv, copy=False)
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            proofs = bp.prove(idx, allv, copy=False)
        dt = (time.perf_counter() - t0) / n
        print(f"threads {bp.threads:3d}: {dt*1e3:8.2f} ms per 64 proofs = {64/dt:8.1f} proofs/s  (host cores {os.cpu_count()})")
        del proofs
