#!/usr/bin/env python3
"""Large proofs as a stream: P single provers (a context each) on P host threads proving the 2^log_n-constraint R1CS side by side --
while one proof's transcript runs on its host core (half of a proof's time, the device idle), another proof's commit has the device.
    python tools/large_proofs_in_flight.py [log_n=20] [P=2] [proofs per prover=6]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ligero_amd.prover import LigeroProver, proofs_equal  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
proofs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
inst, idx, vals, _ = bench.repeated_squaring_instance(log_n)
provers = [LigeroProver(inst) for _ in range(P)]
try:
    last = [None] * P
    for p in provers:
        for _ in range(2):
            p.prove(idx, vals)

    def work(i):
        for _ in range(proofs):
            last[i] = provers[i].prove(idx, vals)
    for n in sorted({1, P}):
        ts = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = time.perf_counter() - t0
        print(f"{n} prover(s) in flight: {n * proofs / dt:.1f} proofs/s ({dt / proofs * 1e3:.1f} ms per proof per prover)", flush=True)
    assert all(proofs_equal(last[0], x) for x in last[1:])
    print("proofs equal across provers:", provers[0].verify(last[-1]))
finally:
    for p in provers:
        p.close()
