#!/usr/bin/env python3
"""Seconds per sharded Poseidon proof on ONE GPU, `world` ranks as threads of this process or as gloo processes, either mode
(LG_PROVER_TIMING=1 adds the per-phase times) -- the probe that found torch's thread pool exhausting the container's CPU quota
(ligero_amd/sharded.py cap_host_threads).

    python tools/sharded_prove_probe.py <world> <coset|relay> [threads|procs]
"""
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
G = os.path.join(ROOT, "tests", "golden")

def body(rank, dist, mode):
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import ShardedLigeroProver
    circ = hp.ArithmeticCircuit.from_r1cs(G+"/poseidon.r1cs"); inst = hp.LigeroInstance(circ)
    w = hp.read_witness(G+"/poseidon_witness.json"); idx, good = list(range(1, w.shape[0])), w[1:]
    with ShardedLigeroProver(inst, dist, device=0, mode=mode) as sp:
        sp.prove(idx, good)
        ts=[]
        for i in range(6):
            t0=time.perf_counter(); sp.prove(idx, good); ts.append(time.perf_counter()-t0)
        return ts

def worker(rank, world, port, mode, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]=str(port); os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank]=body(rank, dist, mode)
    finally:
        dist.destroy_process_group()

if __name__ == "__main__":
    world=int(sys.argv[1]); mode=sys.argv[2]; how=sys.argv[3] if len(sys.argv)>3 else "threads"
    if how=="threads":
        from thread_dist import run_ranks
        out=run_ranks(world, lambda r,d: body(r,d,mode))
    else:
        import torch.multiprocessing as mp
        s=socket.socket(); s.bind(("127.0.0.1",0)); port=s.getsockname()[1]; s.close()
        mgr=mp.Manager(); o=mgr.dict()
        mp.spawn(worker, args=(world, port, mode, o), nprocs=world, join=True)
        out=dict(o)
    print(world, mode, how, [round(x*1e3,1) for x in out[0]])
