"""GPU test of the coset-sharded single-proof commit (ligero_amd/sharded.py) with the REAL
device backend at world_size 2: two processes share the one GPU of the test box, each owning its row shard and half
of the coset planes; the collectives run over gloo (RCCL needs one GPU per rank, which this box does not have -- the
exchange pattern and every device call are the ones the 8-GPU layout uses).  The root must equal the oracle's."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, random_mont

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, rows, k, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ligero_amd.sharded import CosetShardedCommitter, HipStageBackend
        pre = random_mont(515, rows * k).reshape(rows, k, 4)               # same seed on every rank
        be = HipStageBackend(rows, k, device=0)
        sc = CosetShardedCommitter(be, dist)
        r0, r1 = sc.row_range()
        root = sc.commit(pre[r0:r1])
        opened = sc.open_columns([0, 5, 8 * k - 1])
        out[rank] = (root, {j: (c.tobytes(), s.tobytes(), p.tobytes()) for j, (c, s, p) in opened.items()})
        be.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows,k", [(20, 128), (21, 128), (6, 4096), (4, 8192)])   # even / ragged shards, one row per wg, folded k
def test_world2_on_one_gpu_matches_oracle(oracle, rows, k):
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), rows, k, out), nprocs=world, join=True)
    pre = random_mont(515, rows * k).reshape(rows, k, 4)
    ref = oracle.encode_commit(pre, k, 8 * k)
    ecols, esib, epaths = oracle.open_columns(ref["u"], ref["leaves"], ref["nodes"], [0, 5, 8 * k - 1])
    want = {j: (ecols[i].tobytes(), esib[i].tobytes(), epaths[i].tobytes()) for i, j in enumerate([0, 5, 8 * k - 1])}
    assert set(out.keys()) == {0, 1}
    got = {}
    for rank in range(world):
        root, opened = out[rank]
        assert root == ref["root"], rank
        got.update(opened)
    assert got == want                                                    # every opened column came from its owner, bit-exact
