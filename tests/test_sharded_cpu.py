"""CPU tests of the multi-GPU host layer (ligero_amd/sharded.py) with torch.distributed over
gloo at world_size 2: the orchestration (row shards, all-gather of coefficient rows, plane
ownership, all-gather of leaf digests, replicated tree; and proof sharding in throughput mode)
is the product's; the device work is replaced by an oracle-backed stand-in that implements the
backend protocol on numpy arrays.  The result must equal the single-process oracle commit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, random_mont

sys.path.insert(0, ROOT)


class OracleStageBackend:
    """test double for ligero_amd.sharded.HipStageBackend (same methods), CPU + oracle"""

    def __init__(self, rows, k, world=1, pieces=1):
        from oracle import binding as orc
        from ligero_amd.sharded import padded_shard_rows, shard_piece_rows
        self.orc = orc
        self.rows, self.k, self.n, self.nplanes = rows, k, 8 * k, 8
        sub, npieces = shard_piece_rows(rows, world, pieces)
        self.coeff_rows = max(world * padded_shard_rows(rows, world), npieces * world * sub)   # whole exchange pieces (padding rows stay 0xff)
        self.preenc = np.zeros((rows, k, 4), dtype=np.uint64)
        self.coeffs = np.full((self.coeff_rows, k, 4), np.uint64(2**64 - 1), dtype=np.uint64)
        self.leaves = np.zeros((self.n, 32), dtype=np.uint8)
        self.nodes = None
        self.u = {}
        self._fresh = True

    def stage_interpolate(self, preenc_rows, row0, nrows):
        if preenc_rows is not None:
            self.preenc[row0:row0 + nrows] = np.asarray(preenc_rows).reshape(nrows, self.k, 4)
        for r in range(row0, row0 + nrows):
            self.coeffs[r] = self.orc.reed_solomon_interpolate(self.preenc[r], self.k)

    def stage_evaluate_hash(self, planes):
        u = np.stack([self.orc.reed_solomon_evaluate(self.coeffs[r], self.n) for r in range(self.rows)])
        for s in planes:
            for q in range(self.k):
                j = 8 * q + s
                self.leaves[j] = np.frombuffer(self.orc.col_hash(u[:, j]), dtype=np.uint8)
        self.u = u

    def stage_evaluate_rows(self, planes, row0, nrows):
        """split form (lg_stage_evaluate_rows): any rows, any order, each exactly once"""
        if not isinstance(self.u, np.ndarray) or self._fresh:
            self.u = np.zeros((self.rows, self.n, 4), dtype=np.uint64)
            self._seen = np.zeros(self.rows, dtype=bool)
            self._fresh = False
        assert not self._seen[row0:row0 + nrows].any(), "a row evaluated twice"
        for r in range(row0, row0 + nrows):
            self.u[r] = self.orc.reed_solomon_evaluate(self.coeffs[r], self.n)
        self._seen[row0:row0 + nrows] = True

    def stage_hash(self, planes):
        assert self._seen.all(), "lg_stage_hash before every row was evaluated"
        for s in planes:
            for q in range(self.k):
                j = 8 * q + s
                self.leaves[j] = np.frombuffer(self.orc.col_hash(self.u[:, j]), dtype=np.uint8)
        self._fresh = True

    def stage_merkle(self):
        self.nodes = self.orc.merkle_tree(self.leaves)

    def sync(self):
        pass

    def coeffs_bytes(self):
        return torch.from_numpy(self.coeffs.view(np.uint8).reshape(self.coeff_rows, self.k * 32))

    def leaves_bytes(self):
        return torch.from_numpy(self.leaves)

    def root(self):
        return self.nodes[0].tobytes()

    def open_columns(self, indices):
        return self.orc.open_columns(self.u, self.leaves, self.nodes, indices)


class OracleBatchCommitter:
    """test double for LigeroCommitter in throughput mode"""

    def __init__(self, rows, k, batch):
        from oracle import binding as orc
        self.orc, self.rows, self.k, self.batch = orc, rows, k, batch

    def encode_commit(self, pre, want_coeffs=False):
        pre = np.asarray(pre).reshape(self.batch, self.rows, self.k, 4)
        return None, b"".join(self.orc.encode_commit(pre[b], self.k, 8 * self.k, want_u=False)["root"] for b in range(self.batch))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, rows, k, batch, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))      # the ranks share this box's CPU quota (cap_host_threads)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ligero_amd.sharded import CosetShardedCommitter, ShardedBatchCommitter, shard_range
        pre = random_mont(4242, rows * k).reshape(rows, k, 4)           # same seed on every rank
        sc = CosetShardedCommitter(OracleStageBackend(rows, k, world), dist)
        r0, r1 = sc.row_range()
        root = sc.commit(pre[r0:r1])
        opened = sc.open_columns([0, 1, 9, 8 * k - 1])
        # throughput mode: `batch` independent proofs dealt to ranks
        preb = random_mont(777, batch * rows * k).reshape(batch, rows, k, 4)
        sb = ShardedBatchCommitter(lambda b: OracleBatchCommitter(rows, k, b), batch, dist)
        roots = sb.commit(preb[sb.b0:sb.b1])
        out[rank] = (root, sorted(opened), roots, (r0, r1), (sb.b0, sb.b1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rows,k", [(6, 8), (7, 16), (1, 8)])     # even and ragged row shards; a rank with no rows at all
def test_world2_gloo_matches_single_process(oracle, rows, k):
    world, batch = 2, 3
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), rows, k, batch, out), nprocs=world, join=True)
    pre = random_mont(4242, rows * k).reshape(rows, k, 4)
    ref = oracle.encode_commit(pre, k, 8 * k, want_u=False)
    preb = random_mont(777, batch * rows * k).reshape(batch, rows, k, 4)
    ref_roots = b"".join(oracle.encode_commit(preb[b], k, 8 * k, want_u=False)["root"] for b in range(batch))
    assert set(out.keys()) == {0, 1}
    for rank in range(world):
        root, opened, roots, rr, br = out[rank]
        assert root == ref["root"]
        assert roots == ref_roots
    # plane ownership: rank 0 owns planes 0-3 (columns 0, 1), rank 1 owns 4-7 (column 8k-1 = plane 7)
    assert out[0][1] == [0, 1, 9] and out[1][1] == [8 * k - 1]
    assert out[0][3][1] == out[1][3][0] and out[1][3][1] == rows
    assert out[0][3] == (0, min(rows, (rows + 1) // 2))                  # padded equal shards: ceil(rows / 2) rows first


def _worker_n(rank, world, port, rows, k, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))      # the ranks share this box's CPU quota (cap_host_threads)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ligero_amd.sharded import CosetShardedCommitter
        pre = random_mont(99, rows * k).reshape(rows, k, 4)
        sc = CosetShardedCommitter(OracleStageBackend(rows, k, world), dist)
        r0, r1 = sc.row_range()
        root = sc.commit(pre[r0:r1])
        opened = sc.open_columns(list(range(8 * k)))
        out[rank] = (root, sorted(opened), (r0, r1), list(sc.planes))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,k", [(4, 10, 4), (8, 12, 4), (8, 5, 2)])   # the ranks the driver's scaling run uses; fewer rows than ranks
def test_world4_and_world8_gloo(oracle, world, rows, k):
    """the orchestration at the world sizes of the scaling run (1 / 2 / 4 / 8): padded row shards (some ranks may own no row at
    all), one or two planes per rank, every column opened by exactly one rank"""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_n, args=(world, _free_port(), rows, k, out), nprocs=world, join=True)
    pre = random_mont(99, rows * k).reshape(rows, k, 4)
    ref = oracle.encode_commit(pre, k, 8 * k, want_u=False)
    assert set(out.keys()) == set(range(world))
    seen, covered = [], 0
    for rank in range(world):
        root, opened, (r0, r1), planes = out[rank]
        assert root == ref["root"], rank
        assert planes == list(range(rank * 8 // world, (rank + 1) * 8 // world))
        assert all(j % 8 in planes for j in opened)
        seen += opened
        assert r0 == covered or r0 == r1 == rows
        covered = max(covered, r1)
    assert sorted(seen) == list(range(8 * k)) and covered == rows


def test_thread_ranks_equal_the_gloo_group(oracle):
    """tests/thread_dist.py (eight ranks as threads of one process -- what the GPU tests use at world 8, where the GPU box's
    limit of six processes per card rules out a process per rank) gives the commitment the gloo process group gives"""
    from thread_dist import run_ranks
    from ligero_amd.sharded import CosetShardedCommitter
    world, rows, k = 8, 12, 4
    pre = random_mont(99, rows * k).reshape(rows, k, 4)

    def body(rank, tdist):
        sc = CosetShardedCommitter(OracleStageBackend(rows, k, world), tdist)
        r0, r1 = sc.row_range()
        return sc.commit(pre[r0:r1]), sorted(sc.open_columns(list(range(8 * k))))

    res = run_ranks(world, body)
    ref = oracle.encode_commit(pre, k, 8 * k, want_u=False)
    assert all(root == ref["root"] for root, _ in res)
    assert sorted(j for _, opened in res for j in opened) == list(range(8 * k))
    with pytest.raises(ZeroDivisionError):                                  # a failing rank surfaces; nobody hangs in a collective
        run_ranks(2, lambda rank, d: 1 // rank if rank == 0 else d.all_gather_into_tensor(torch.zeros(2), torch.zeros(1)), timeout=20)


def _worker_pieces(rank, world, port, rows, k, pieces, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))      # the ranks share this box's CPU quota (cap_host_threads)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ligero_amd.sharded import CosetShardedCommitter
        pre = random_mont(31, rows * k).reshape(rows, k, 4)
        sc = CosetShardedCommitter(OracleStageBackend(rows, k, world, pieces), dist, exchange_pieces=pieces)
        mine = sc.row_ranges()
        local = np.concatenate([pre[a:a + n] for a, n in mine]) if mine else None
        root = sc.commit(local)
        again = sc.commit(None)                                             # resident rows
        out[rank] = (root, again, sc.piece_plan(), sorted(sc.stage_ms), mine)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,rows,k,pieces", [(2, 12, 4, 2), (2, 13, 4, 3), (4, 9, 2, 2), (2, 3, 4, 8)])
def test_pipelined_exchange_matches_single_process(oracle, world, rows, k, pieces):
    """exchange_pieces > 1: every rank owns a sub-block of every piece, so each piece of the coefficient all-gather is one
    in-place collective on whole rows and complete row prefixes arrive in order -- even, ragged and short last pieces, more
    pieces than rows per rank"""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_pieces, args=(world, _free_port(), rows, k, pieces, out), nprocs=world, join=True)
    pre = random_mont(31, rows * k).reshape(rows, k, 4)
    ref = oracle.encode_commit(pre, k, 8 * k, want_u=False)
    owned = []
    for rank in range(world):
        root, again, plan, stages, mine = out[rank]
        assert root == ref["root"] and again == ref["root"], rank
        assert sum(n for _, n in plan) == rows and [o for o, _ in plan] == sorted(o for o, _ in plan) and len(plan) <= pieces
        assert stages == ["allgather_coeffs", "allgather_digests", "evaluate_hash", "interpolate", "merkle"]
        assert len(mine) <= len(plan)                                    # at most one sub-block per piece
        owned += [r for a, n in mine for r in range(a, a + n)]
    assert sorted(owned) == list(range(rows))                            # every row has exactly one owner


def test_pipelined_exchange_on_eight_thread_ranks(oracle):
    from thread_dist import run_ranks
    from ligero_amd.sharded import CosetShardedCommitter
    world, rows, k = 8, 21, 4
    pre = random_mont(32, rows * k).reshape(rows, k, 4)

    def body(rank, tdist):
        sc = CosetShardedCommitter(OracleStageBackend(rows, k, world, 2), tdist, exchange_pieces=2)
        mine = sc.row_ranges()
        return sc.commit(np.concatenate([pre[a:a + n] for a, n in mine]) if mine else None)

    ref = oracle.encode_commit(pre, k, 8 * k, want_u=False)
    assert all(root == ref["root"] for root in run_ranks(world, body))


def test_single_process_degenerate(oracle):
    from ligero_amd.sharded import CosetShardedCommitter, owned_planes, padded_shard_range, padded_shard_rows, shard_range
    rows, k = 5, 8
    pre = random_mont(99, rows * k).reshape(rows, k, 4)
    sc = CosetShardedCommitter(OracleStageBackend(rows, k), None)
    assert sc.commit(pre) == oracle.encode_commit(pre, k, 8 * k, want_u=False)["root"]
    assert owned_planes(8, 4, 3) == [6, 7] and owned_planes(16, 8, 1) == [2, 3]
    assert [shard_range(10, 4, r) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    # the shapes BASELINE configs[2] / [3] name, on 8 GPUs: equal shards of ceil(rows / 8), the last one short
    assert padded_shard_rows(20068, 8) == 2509 and padded_shard_range(20068, 8, 7) == (17563, 20068)
    assert padded_shard_rows(10036, 8) == 1255 and padded_shard_range(10036, 8, 7) == (8785, 10036)
    assert [padded_shard_range(3, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert [padded_shard_range(5, 4, r) for r in range(4)] == [(0, 2), (2, 4), (4, 5), (5, 5)]
    with pytest.raises(ValueError):
        owned_planes(8, 3, 0)
    # exchange pieces: rank g owns sub-block g of every piece (the rule lg_shard_row_ranges implements in the library)
    from ligero_amd.sharded import shard_row_ranges
    assert shard_row_ranges(20068, 8, 0, 1) == [(0, 2509)] and shard_row_ranges(20068, 8, 7, 1) == [(17563, 2505)]
    assert shard_row_ranges(20068, 8, 3, 4) == [(1884 + p * 5024, 628) for p in range(3)] + [(1884 + 3 * 5024, 628)]
    assert shard_row_ranges(20068, 8, 7, 4)[-1] == (4396 + 3 * 5024, 20068 - (4396 + 3 * 5024))
    assert shard_row_ranges(3, 4, 3, 8) == [] and shard_row_ranges(3, 4, 1, 8) == [(2, 1)]      # sub-blocks are even when there are pieces
