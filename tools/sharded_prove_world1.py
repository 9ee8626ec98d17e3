#!/usr/bin/env python3
"""One 2^log_n-constraint proof through the SHARDED provers at world 1 over RCCL (every collective issued, as an identity): both modes,
ms per proof -- a single process without a launcher, so that it can run under rocprofv3.
    python tools/sharded_prove_world1.py [log_n=20] [proofs=3] [modes=coset,relay]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
from ligero_amd.prover import LigeroProver, ShardedLigeroProver, proofs_equal  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
modes = (sys.argv[3] if len(sys.argv) > 3 else "coset,relay").split(",")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    inst, idx, vals, _ = bench.repeated_squaring_instance(log_n)
    with LigeroProver(inst) as single:
        ref = single.prove(idx, vals)
    for mode in modes:
        with ShardedLigeroProver(inst, dist, device=0, collectives_at_world_1=True, mode=mode) as sp:
            sp.prove(idx, vals)
            t0 = time.perf_counter()
            for _ in range(proofs):
                p = sp.prove(idx, vals)
            dt = (time.perf_counter() - t0) / proofs
            print(f"{mode}: {dt * 1e3:.1f} ms per proof, trace on the {'device' if sp.device_trace else 'host'}, equal to the single prover's: {proofs_equal(ref, p)}", flush=True)
finally:
    dist.destroy_process_group()
