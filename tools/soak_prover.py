#!/usr/bin/env python3
"""Determinism soak of the provers above the device library on one GPU: the 64 committed Poseidon witnesses
(tests/golden/poseidon_witness_batch64.bin) proved over and over by ONE long-lived LigeroProver (one at a time, in random order,
valid and bit-flipped) and by long-lived LigeroBatchProvers of several batch sizes (random selections and orders of the witnesses
per call, borrowed and copied proofs) -- every proof compared field for field with the proof a fresh single prover made of the same
witness at the start, and verified (or rejected, for the flipped ones) now and then.  What it is after: state carried between
proofs -- reused device contexts and page-locked buffers, the from-`w` commit's gate map, the batch prover's pipelined contexts and
worker threads.

    python tools/soak_prover.py <seconds> [seed=1]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ligero_amd.prover import LigeroBatchProver, LigeroProver, proofs_equal  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    inst, idx, vals = bench.poseidon_batch_inputs()
    bad = vals.copy()
    bad[:, 0, 0] ^= np.uint64(1)                              # the same 64 witnesses with one bit flipped: proofs that must not verify
    with LigeroProver(inst) as fresh:
        ref = [fresh.prove(idx, vals[i]) for i in range(64)]
        ref_bad = [fresh.prove(idx, bad[i]) for i in range(8)]
        assert all(fresh.verify(p) for p in ref[:4]) and not any(fresh.verify(p) for p in ref_bad[:2])
    n_single = n_batch = n_verify = 0
    t_end = time.time() + seconds
    t_mark = time.time()
    sizes = (1, 3, 16, 64)
    with LigeroProver(inst) as single:
        provers = {b: LigeroBatchProver(inst, b) for b in sizes}
        try:
            while time.time() < t_end:
                what = int(rng.integers(3))
                if what == 0:                                 # one at a time
                    for _ in range(int(rng.integers(1, 9))):
                        if rng.integers(4) == 0:
                            i = int(rng.integers(8))
                            p = single.prove(idx, bad[i])
                            assert proofs_equal(p, ref_bad[i]), ("single, flipped", i)
                            if rng.integers(4) == 0:
                                assert not single.verify(p)
                                n_verify += 1
                        else:
                            i = int(rng.integers(64))
                            p = single.prove(idx, vals[i])
                            assert proofs_equal(p, ref[i]), ("single", i)
                            if rng.integers(8) == 0:
                                assert single.verify(p)
                                n_verify += 1
                        n_single += 1
                else:                                         # a batch: any selection, any order
                    b = sizes[int(rng.integers(len(sizes)))]
                    pick = rng.integers(64, size=b)
                    copy = bool(rng.integers(2))
                    proofs = provers[b].prove(idx, vals[pick], copy=copy)
                    for j, i in enumerate(pick):
                        assert proofs_equal(proofs[j], ref[int(i)]), ("batch", b, j, int(i), copy)
                    if rng.integers(4) == 0:
                        assert single.verify(proofs[int(rng.integers(b))])
                        n_verify += 1
                    del proofs
                    n_batch += b
                if time.time() - t_mark > 30:
                    t_mark = time.time()
                    print(f"  ... {n_single} single, {n_batch} batched proofs", flush=True)
        finally:
            for p in provers.values():
                p.close()
    print(f"soak prover (seed {seed}): {n_single} proofs one at a time and {n_batch} in batches of {sizes} on long-lived provers, "
          f"{n_verify} verified, every proof equal to a fresh prover's")


if __name__ == "__main__":
    main()
