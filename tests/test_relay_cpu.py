"""CPU tests of the row-relay commit (ligero_amd/sharded.py RowRelayCommitter): the orchestration -- who
keeps which rows, the order in which the ranks take their turn on the columns, what is sent where -- is the product's; the
device work is replaced by an oracle-backed double that parks the Blake2s states in the device library's LG_BUF_HSTATE
layout (oracle/model_relay.py).  The root must equal the single-process oracle commit (src/ligero/mod.rs:521-551)."""
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


class OracleRelayBackend:
    """test double for ligero_amd.sharded.HipRelayBackend (same methods), CPU + oracle"""

    def __init__(self, local_rows, k):
        from oracle import binding as orc
        from oracle import model_relay
        self.orc, self.mr = orc, model_relay
        self.local_rows, self.k, self.n, self.nplanes = local_rows, k, 8 * k, 8
        self.ki = k
        self.pre = np.zeros((local_rows, k, 4), dtype=np.uint64)
        self.coeffs = np.zeros((local_rows, k, 4), dtype=np.uint64)
        self.u = np.zeros((local_rows, self.n, 4), dtype=np.uint64)
        self.hstate = torch.zeros((self.nplanes, self.ki * model_relay.HSTATE_BYTES), dtype=torch.uint8)
        self.leaves = torch.zeros((self.n, 32), dtype=torch.uint8)
        self.nodes = None
        self.hash_calls = []

    def stage_interpolate(self, preenc_rows, row0, nrows):
        if preenc_rows is not None:
            self.pre[row0:row0 + nrows] = np.asarray(preenc_rows).reshape(nrows, self.k, 4)
        for r in range(row0, row0 + nrows):
            self.coeffs[r] = self.orc.reed_solomon_interpolate(self.pre[r], self.k)
        self._evaluated = np.zeros(self.local_rows, dtype=bool)

    def stage_evaluate_rows(self, row0, nrows):
        assert not self._evaluated[row0:row0 + nrows].any(), "a row evaluated twice"
        for r in range(row0, row0 + nrows):
            self.u[r] = self.orc.reed_solomon_evaluate(self.coeffs[r], self.n)
        self._evaluated[row0:row0 + nrows] = True

    def stage_hash_rows(self, plane0, nplanes, row0, nrows, col_pos, col_rows):
        assert self._evaluated[row0:row0 + nrows].all(), "hashed before evaluated"
        self.hash_calls.append((plane0, nplanes, row0, nrows, col_pos))
        hb = self.mr.HSTATE_BYTES
        canon = self.orc.from_mont(self.u[row0:row0 + nrows]).view(np.uint8).reshape(nrows, self.n, 32)
        for s in range(plane0, plane0 + nplanes):
            cols = np.arange(self.k) * 8 + s                                     # column j = 8 q + s <-> state record [s][q]
            if col_pos == 0:
                h = self.mr.ColumnRelayHasher(self.k, col_rows)
            else:
                h = self.mr.ColumnRelayHasher.import_state(self.hstate[s].numpy().reshape(self.k, hb), col_pos, col_rows)
            h.absorb(canon[:, cols])
            if col_pos + nrows == col_rows:
                self.leaves[cols] = torch.from_numpy(h.digests())
            else:
                self.hstate[s] = torch.from_numpy(h.export_state().reshape(-1))

    def hstate_bytes(self):
        return self.hstate

    def leaves_bytes(self):
        return self.leaves

    def stage_merkle(self):
        self.nodes = self.orc.merkle_tree(self.leaves.numpy())

    def sync(self):
        pass

    def root(self):
        return self.nodes[0].tobytes()

    def open_columns(self, indices):
        cols, sib, paths = self.orc.open_columns(np.ascontiguousarray(self.u) if self.local_rows else np.zeros((1, self.n, 4), dtype=np.uint64),
                                                 self.leaves.numpy(), self.nodes, indices)
        return cols[:, :self.local_rows], sib, paths


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _local_rows(pre, ranges):
    return np.concatenate([pre[a:a + n] for a, n in ranges]) if ranges else None


def _rank_body(rank, d, rows, k, layout, groups, seed=4243):
    from ligero_amd.sharded import RowRelayCommitter
    pre = random_mont(seed, rows * k).reshape(rows, k, 4)                 # same seed on every rank
    rc = RowRelayCommitter(lambda local: OracleRelayBackend(local, k), rows, d, plane_groups=groups, layout=layout)
    root = rc.commit(_local_rows(pre, rc.row_ranges()))
    again = rc.commit(None)                                               # resident rows
    idx = [0, 1, 9, 8 * k - 1]
    cols, sib, paths = rc.open_columns(idx)
    return root, again, cols, sib.tobytes(), paths.tobytes(), rc.row_ranges(), sorted(rc.stage_ms), list(rc.be.hash_calls)


def _worker(rank, world, port, rows, k, layout, groups, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))      # the ranks share this box's CPU quota (cap_host_threads)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = _rank_body(rank, dist, rows, k, layout, groups)
    finally:
        dist.destroy_process_group()


def _subgroup_worker(rank, world, port, rows, k, out):
    """a relay over the sub-group {1, 2} of a three-process world: group ranks 0, 1 are world ranks 1, 2"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))      # the ranks share this box's CPU quota (cap_host_threads)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grp = dist.new_group([1, 2])                                      # (every process calls new_group)
        if rank == 0:
            return
        from ligero_amd.sharded import RowRelayCommitter
        pre = random_mont(4243, rows * k).reshape(rows, k, 4)
        rc = RowRelayCommitter(lambda local: OracleRelayBackend(local, k), rows, dist, group=grp)
        out[rank] = (rc.commit(_local_rows(pre, rc.row_ranges())), rc.rank, rc.world)
    finally:
        dist.destroy_process_group()


def test_relay_over_a_subgroup_names_its_peers_by_world_rank(oracle):
    rows, k = 10, 4
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_subgroup_worker, args=(3, _free_port(), rows, k, out), nprocs=3, join=True)
    want = oracle.encode_commit(random_mont(4243, rows * k).reshape(rows, k, 4), k, 8 * k, want_u=False)["root"]
    assert dict(out) == {1: (want, 0, 2), 2: (want, 1, 2)}


def _check(oracle, out, world, rows, k, layout):
    from ligero_amd.sharded import RowRelayCommitter, relay_chain
    pre = random_mont(4243, rows * k).reshape(rows, k, 4)
    ref = oracle.encode_commit(pre, k, 8 * k)
    idx = [0, 1, 9, 8 * k - 1]
    ecols, esib, epaths = oracle.open_columns(ref["u"], ref["leaves"], ref["nodes"], idx)
    assert set(out.keys()) == set(range(world))
    pieces = []
    for rank in range(world):
        root, again, cols, sib, paths, ranges, stages, calls = out[rank]
        assert root == ref["root"] and again == ref["root"], rank
        assert sib == esib.tobytes() and paths == epaths.tobytes(), rank      # the tree is replicated: every rank serves whole paths
        assert stages == ["digests", "encode", "merkle", "relay"]
        pieces.append(cols)
    # the ranks' row pieces put together are the reference's columns
    merged = np.empty((len(idx), rows, 4), dtype=np.uint64)
    for pos, n, owner, local in relay_chain(rows, world, layout):
        merged[:, pos:pos + n] = pieces[owner][:, local:local + n]
    assert np.array_equal(merged, ecols)


# even / odd row boundaries (a Blake2s block holds two rows: an odd boundary parks 40 bytes), a rank without rows, plane groups
@pytest.mark.parametrize("rows,k,layout,groups", [(6, 8, "contiguous", 1), (7, 16, "contiguous", 1), (10, 8, "contiguous", 2), (1, 8, "contiguous", 1),
                                                  (12, 8, "blocks", 1), (20, 8, "blocks", 4), (4, 8, "blocks", 1),
                                                  (21, 8, "round_robin:2", 1), (9, 16, "round_robin:3", 1), (3, 8, "round_robin:4", 1)])
def test_world2_gloo_matches_single_process(oracle, rows, k, layout, groups):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), rows, k, layout, groups, out), nprocs=world, join=True)
    _check(oracle, out, world, rows, k, layout)


@pytest.mark.parametrize("world,rows,k,layout,groups", [(4, 10, 4, "contiguous", 1), (8, 12, 4, "contiguous", 2), (8, 5, 2, "contiguous", 1),
                                                        (4, 20, 4, "blocks", 1), (8, 36, 4, "blocks", 8), (8, 12, 2, "blocks", 1),
                                                        (4, 37, 4, "round_robin:2", 2), (8, 70, 4, "round_robin:4", 4), (8, 19, 2, "round_robin:8", 1)])
def test_world4_and_world8_on_thread_ranks(oracle, world, rows, k, layout, groups):
    """the world sizes of the scaling run; ragged and empty shards, the four-block layout (4 G hops), one hop per plane"""
    from thread_dist import run_ranks
    out = dict(enumerate(run_ranks(world, lambda rank, d: _rank_body(rank, d, rows, k, layout, groups))))
    _check(oracle, out, world, rows, k, layout)


def test_row_ownership_and_chain():
    from ligero_amd.sharded import relay_chain, relay_row_ranges
    # BASELINE configs[3]: 20 068 rows on 8 GPUs
    r = [relay_row_ranges(20068, 8, g) for g in range(8)]
    assert r[0] == [(0, 2508)] and r[7] == [(17558, 2510)] and sum(n for x in r for _, n in x) == 20068
    assert all(a % 2 == 0 for x in r for a, _ in x)                      # hand-overs at Blake2s block boundaries
    assert max(n for x in r for _, n in x) - min(n for x in r for _, n in x) <= 2
    assert relay_row_ranges(20067, 4, 3) == [(15048, 5019)]              # the last rank takes the odd row
    b = [relay_row_ranges(20068, 8, g, "blocks") for g in range(8)]
    assert b[0] == [(0, 627), (5017, 627), (10034, 627), (15051, 627)] and b[7][3] == (15051 + 4389, 628)
    chain = relay_chain(20068, 8, "blocks")
    assert len(chain) == 32 and [c[2] for c in chain[:9]] == [0, 1, 2, 3, 4, 5, 6, 7, 0]
    assert relay_row_ranges(3, 4, 3) == [(0, 3)] and relay_row_ranges(3, 4, 0) == []     # fewer row pairs than ranks: the first ranks keep none
    with pytest.raises(ValueError):
        relay_row_ranges(10, 2, 0, "blocks")


def test_single_process_degenerate(oracle):
    from ligero_amd.sharded import RowRelayCommitter
    rows, k = 5, 8
    pre = random_mont(99, rows * k).reshape(rows, k, 4)
    assert RowRelayCommitter(lambda local: OracleRelayBackend(local, k), 8, None, plane_groups=4, layout="blocks").groups == 1   # wrap-around chain
    rc = RowRelayCommitter(lambda local: OracleRelayBackend(local, k), rows, None)
    assert rc.commit(pre) == oracle.encode_commit(pre, k, 8 * k, want_u=False)["root"]
    assert rc.be.hash_calls == [(0, 8, 0, 5, 0)]
