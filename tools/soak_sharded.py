#!/usr/bin/env python3
"""Determinism soak of the two one-proof-over-several-GPUs commits on ONE GPU: `world` sharded contexts of the real device backend on
threads of this process (tests/thread_dist.py: the box admits six processes on its card), commits QUEUED back to back the way
bench.py's sharded leg issues them (lg_commit_sharded / lg_commit_row_relay, no host wait in between), the input swapped between two
matrices now and then, the root read after every burst and compared with the root of the same matrix committed by one ordinary
context.  What it would catch: a buffer of commit i reused by commit i + 1 before its reader is done (column states, leaves, the
coefficient rows of an exchange piece), on either stream.

    python tools/soak_sharded.py <seconds> [world=4] [rows=1501] [k=1024]
"""
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ligero_amd  # noqa: E402
from ligero_amd.sharded import CosetShardedCommitter, HipRelayBackend, HipStageBackend, RowRelayCommitter  # noqa: E402
from thread_dist import run_ranks  # noqa: E402


def matrix(seed, rows, k):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 2**62, size=(rows, k, 4), dtype=np.uint64)
    a[..., 3] &= np.uint64((1 << 60) - 1)        # < 2^252 < p: valid Montgomery-form words
    return a


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1501
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    mats = [matrix(11, rows, k), matrix(12, rows, k)]
    want = []
    with ligero_amd.LigeroCommitter(rows=rows, k=k) as c:
        for m in mats:
            want.append(c.encode_commit(m)[1])
    assert want[0] != want[1]

    def take(m, ranges):
        return np.concatenate([m[a:a + n] for a, n in ranges]) if ranges else None

    def body(rank, dist):
        rnd = random.Random(99)                     # the same schedule on every rank
        modes = []
        be = HipStageBackend(rows, k, device=0, world=world, rank=rank, pieces=4)
        modes.append(("coset, one all-gather", CosetShardedCommitter(be, dist)))
        modes.append(("coset, 4 pieces", CosetShardedCommitter(be, dist, exchange_pieces=4)))
        modes.append(("row relay", RowRelayCommitter(lambda local: HipRelayBackend(local, k, device=0), rows, dist)))
        modes.append(("row relay, 2 plane groups", RowRelayCommitter(lambda local: HipRelayBackend(local, k, device=0), rows, dist, plane_groups=2)))
        counts = {name: [0, 0] for name, _ in modes}
        t_end = time.time() + seconds
        try:
            while True:
                go = torch.tensor([1 if time.time() < t_end else 0], dtype=torch.int64)     # the ranks agree on when to stop
                flags = torch.zeros(world, dtype=torch.int64)
                dist.all_gather_into_tensor(flags, go)
                if int(flags.min()) == 0:
                    break
                for name, cm in modes:
                    which = rnd.randrange(2)
                    assert cm.commit(take(mats[which], cm.row_ranges())) == want[which], (name, "upload", rank)
                    burst = rnd.randrange(1, 12)
                    for _ in range(burst):
                        cm.commit_queued(None)      # resident rows, nothing waited for
                    assert cm.be.root() == want[which], (name, "burst", rank)
                    counts[name][0] += burst + 1
                    counts[name][1] += 2
        finally:
            for name, cm in modes[2:]:
                cm.be.close()
            be.close()
        return counts

    out = run_ranks(world, body, timeout=max(300, int(seconds) + 120))
    for name, (commits, checks) in out[0].items():
        print(f"soak {name}: world {world}, {rows} x {k}: {commits} commits per rank, {checks} root checks per rank, all equal")


if __name__ == "__main__":
    main()
