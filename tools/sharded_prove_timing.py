#!/usr/bin/env python3
"""Per-phase wall time (LG_PROVER_TIMING=1) of the SHARDED prover on the 2^log_n-constraint R1CS, one rank over RCCL with the
collectives forced (identities): what a rank of a multi-GPU proof spends outside the exchanges.
    python tools/sharded_prove_timing.py [log_n] [proofs]"""
import os, sys, time
os.environ["LG_PROVER_TIMING"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import bench
from ligero_amd.prover import ShardedLigeroProver

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
proofs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    inst, idx, vals, setup = bench.repeated_squaring_instance(log_n)
    with ShardedLigeroProver(inst, dist, device=0, collectives_at_world_1=True) as sp:
        for i in range(proofs):
            t0 = time.perf_counter()
            proof = sp.prove(idx, vals)
            print(f"proof {i}: {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr)
finally:
    dist.destroy_process_group()
