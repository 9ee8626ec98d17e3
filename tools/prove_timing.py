#!/usr/bin/env python3
"""wall time of the full prove() / verify() of the Poseidon fixture (C++ host + device), and where it goes"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from ligero_amd import host_pipeline as hp
from ligero_amd.prover import LigeroProver

G = os.path.join(ROOT, "tests", "golden")
t0 = time.perf_counter()
circ = hp.ArithmeticCircuit.from_r1cs(os.path.join(G, "poseidon.r1cs"))
inst = hp.LigeroInstance(circ)
t1 = time.perf_counter()
w = hp.read_witness(os.path.join(G, "poseidon_witness.json"))
vals = w[1:]
idx = list(range(1, len(w)))
prover = LigeroProver(inst)
t2 = time.perf_counter()
print(f"setup: r1cs -> circuit -> LigeroCircuit::new {1e3*(t1-t0):.1f} ms; device context {1e3*(t2-t1):.1f} ms")
for _ in range(2):
    p = prover.prove(idx, vals)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    p = prover.prove(idx, vals)
tp = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    ok = prover.verify(p)
tv = (time.perf_counter() - t0) / n
print(f"prove {tp*1e3:.2f} ms   verify {tv*1e3:.2f} ms   accepted {ok}")
# pieces of prove on the host
t0 = time.perf_counter(); pre, _ = inst.build_preenc_u(idx, vals); a = time.perf_counter() - t0
t0 = time.perf_counter(); r = hp.field_elements_from_seed(bytes(32), 4 * inst.m * inst.k); b = time.perf_counter() - t0
t0 = time.perf_counter(); inst.a_row_mul(r); c = time.perf_counter() - t0
t0 = time.perf_counter(); s = hp.PoseidonSponge(); s.absorb_elements(r[:255]); s.squeeze_bytes(32); d = time.perf_counter() - t0
print(f"host pieces: trace+preenc_u {a*1e3:.2f} ms, ChaCha20 r_linear ({4*inst.m*inst.k} elems) {b*1e3:.2f} ms, A.row_mul {c*1e3:.2f} ms, sponge setup+absorb 255+squeeze {d*1e3:.2f} ms")
