#!/usr/bin/env python3
"""Per-phase wall time of ONE proof of the 2^log_n-constraint repeated-squaring R1CS on the single-GPU prover
(LG_PROVER_TIMING=1 makes the C++ prover print its phases on stderr).   python tools/s20_prove_timing.py [log_n] [proofs]"""
import os, sys, time
os.environ["LG_PROVER_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # repeated_squaring_instance
from ligero_amd.prover import LigeroProver

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
inst, idx, vals, setup = bench.repeated_squaring_instance(log_n)
print("setup", setup, "dims", (inst.m, inst.k, inst.n, inst.t), file=sys.stderr)
with LigeroProver(inst) as p:
    for i in range(proofs):
        t0 = time.perf_counter()
        proof = p.prove(idx, vals)
        print(f"proof {i}: {1e3 * (time.perf_counter() - t0):.1f} ms in all (incl. the ctypes call and the proof handle)", file=sys.stderr)
    t0 = time.perf_counter()
    ok = p.verify(proof)
    print(f"verify: {1e3 * (time.perf_counter() - t0):.1f} ms -> {ok}", file=sys.stderr)
