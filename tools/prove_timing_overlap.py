#!/usr/bin/env python3
"""proofs/s when several batch provers run concurrently (host phases of one overlap device phases of another)"""
import os, sys, time, threading
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
for nprov, batch, threads in ((2, 64, 16), (3, 64, 16), (4, 64, 16), (2, 128, 16), (3, 64, 8), (4, 64, 8)):
    provers = [LigeroBatchProver(inst, batch, threads=threads) for _ in range(nprov)]
    vals = [np.ascontiguousarray(np.concatenate([allv] * 8)[(i * batch) % 64:(i * batch) % 64 + batch]) for i in range(nprov)]
    steps = 5
    def work(i):
        for _ in range(steps):
            provers[i].prove(idx, vals[i], copy=False)
    for i in range(nprov):
        provers[i].prove(idx, vals[i], copy=False)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(nprov)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    n = nprov * batch * steps
    print(f"{nprov} provers x batch {batch} x {threads} threads: {n/dt:8.1f} proofs/s")
    for p in provers: p.close()
