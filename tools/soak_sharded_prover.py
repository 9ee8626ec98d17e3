#!/usr/bin/env python3
"""Determinism soak of the sharded PROVER on one GPU: `world` ShardedLigeroProver instances (coset mode and row relay) on threads
of this process, the Poseidon fixture proved over and over, alternating between the valid witness and one with a flipped bit (two
different matrices, commitments and proofs); every proof must equal the first proof of the same witness field for field, on every
rank, and the first ones must equal the single-GPU prover's.

    python tools/soak_sharded_prover.py <seconds> [world=4] [poseidon|s18|s20]

s18 / s20: the synthetic 2^18- / 2^20-constraint repeated-squaring R1CS (BASELINE configs[2]) instead of the Poseidon fixture.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ligero_amd import host_pipeline as hp  # noqa: E402
from ligero_amd.prover import LigeroProver, ShardedLigeroProver, proofs_equal  # noqa: E402
from thread_dist import run_ranks  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    case = sys.argv[3] if len(sys.argv) > 3 else "poseidon"
    if case == "poseidon":
        circ = hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "poseidon.r1cs"))
        inst = hp.LigeroInstance(circ)
        w = hp.read_witness(os.path.join(GOLDEN, "poseidon_witness.json"))
        idx, good = list(range(1, w.shape[0])), w[1:]
    else:
        import bench
        inst, idx, good, _ = bench.repeated_squaring_instance(int(case[1:]))
        good = np.ascontiguousarray(good)
    bad = good.copy()
    bad[0, 0] ^= np.uint64(1)
    wit = [good, bad]
    with LigeroProver(inst) as single:
        ref = [single.prove(idx, v) for v in wit]
        assert single.verify(ref[0]) and not single.verify(ref[1])

    def body(rank, dist):
        counts = {}
        for mode in ("coset", "relay"):
            with ShardedLigeroProver(inst, dist, device=0, mode=mode) as sp:
                first = [sp.prove(idx, v) for v in wit]
                assert proofs_equal(first[0], ref[0]) and proofs_equal(first[1], ref[1]), (mode, rank)
                t_end = time.time() + seconds / 2
                n = 0
                while True:
                    go = torch.tensor([1 if time.time() < t_end else 0], dtype=torch.int64)     # the ranks agree on when to stop
                    flags = torch.zeros(world, dtype=torch.int64)
                    dist.all_gather_into_tensor(flags, go)
                    if int(flags.min()) == 0:
                        break
                    which = n % 2 if (n // 7) % 2 == 0 else 0          # runs of alternating and of repeated witnesses
                    assert proofs_equal(sp.prove(idx, wit[which]), first[which]), (mode, rank, n)
                    n += 1
                counts[mode] = n
        return counts

    out = run_ranks(world, body, timeout=max(300, int(seconds) + 180))
    for mode, n in out[0].items():
        print(f"soak sharded prover, {mode}: world {world}, {case}: {n} proofs per rank after the first two, all equal to the single-GPU proofs")


if __name__ == "__main__":
    main()
