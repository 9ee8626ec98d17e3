"""GPU tests of the row-relay commit (ligero_amd/sharded.py RowRelayCommitter over lg_stage_hash_rows):
rows sharded end to end, the Blake2s state of every column handed from rank to rank.  Bit-exact with the oracle's
single-process commit of src/ligero/mod.rs:521-551 -- root, digests, opened columns, and the parked states themselves
(oracle/model_relay.py states the LG_BUF_HSTATE record) -- with the REAL device backend: worlds 2 and 4 as gloo processes
sharing this box's GPU, world 8 as eight contexts on threads of this process (the box admits six processes on its card)."""
import json
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, random_mont

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("rows,k,cuts", [(7, 128, [1, 2, 3, 6]), (10, 128, [5, 4]), (1, 16, []), (5, 8192, [2, 3]), (4, 4096, [1])])
def test_parked_states_equal_the_model(oracle, rows, k, cuts):
    """lg_stage_hash_rows over [0, cut) then [cut, rows): the exported LG_BUF_HSTATE record (chaining value + the 8 or 40
    carried bytes of the block in progress) equals the model's, the digests and the root equal the oracle's -- even and odd
    cut positions, one row, folded k = 8192 (16 planes)"""
    from ligero_amd.sharded import HipRelayBackend
    from oracle import model_relay as mr
    n = 8 * k
    pre = random_mont(8181, rows * k).reshape(rows, k, 4)
    ref = oracle.encode_commit(pre, k, n)
    canon = oracle.from_mont(ref["u"]).view(np.uint8).reshape(rows, n, 32)
    be = HipRelayBackend(rows, k)
    try:
        np_, ki = be.nplanes, be.ki
        for cut in cuts + [rows]:
            be.stage_interpolate(pre, 0, rows)
            be.stage_evaluate_rows(0, rows)
            be.stage_hash_rows(0, np_, 0, cut, 0, rows)
            if cut < rows:
                be.sync()                      # (the copy below runs on torch's stream, not the library's)
                got = be.hstate_bytes().cpu().numpy().reshape(np_, ki, mr.HSTATE_BYTES)
                h = mr.ColumnRelayHasher(n, rows)
                h.absorb(canon[:cut])
                want = h.export_state().reshape(ki, np_, mr.HSTATE_BYTES).transpose(1, 0, 2)      # column j = np q + s -> record [s][q]
                used = 32 + (40 if cut & 1 else 8)
                assert np.array_equal(got[:, :, :used], want[:, :, :used]), cut
                be.stage_hash_rows(0, np_, cut, rows - cut, cut, rows)
            be.stage_merkle()
            assert be.root() == ref["root"], cut
            assert np.array_equal(be.c.leaves()[0], ref["leaves"]), cut
    finally:
        be.close()


def _local_rows(pre, ranges):
    return np.concatenate([pre[a:a + n] for a, n in ranges]) if ranges else None


def _rank_body(rank, d, rows, k, layout, groups, seed=616):
    from ligero_amd.sharded import HipRelayBackend, RowRelayCommitter
    pre = random_mont(seed, rows * k).reshape(rows, k, 4)                 # same seed on every rank
    rc = RowRelayCommitter(lambda local: HipRelayBackend(local, k, device=0), rows, d, plane_groups=groups, layout=layout)
    try:
        root = rc.commit(_local_rows(pre, rc.row_ranges()))
        again = rc.commit(None)                                           # resident rows
        idx = [0, 5, 8 * k - 1]
        cols, sib, paths = rc.open_columns(idx)
        return root, again, cols, sib.tobytes(), paths.tobytes(), dict(rc.stage_ms)
    finally:
        rc.be.close()


def _worker(rank, world, port, rows, k, layout, groups, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))      # the ranks share this box's CPU quota (cap_host_threads)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = _rank_body(rank, dist, rows, k, layout, groups)
    finally:
        dist.destroy_process_group()


def _check(oracle, out, world, rows, k, layout):
    from ligero_amd.sharded import relay_chain
    pre = random_mont(616, rows * k).reshape(rows, k, 4)
    ref = oracle.encode_commit(pre, k, 8 * k)
    idx = [0, 5, 8 * k - 1]
    ecols, esib, epaths = oracle.open_columns(ref["u"], ref["leaves"], ref["nodes"], idx)
    pieces = []
    for rank in range(world):
        root, again, cols, sib, paths, stage_ms = out[rank]
        assert root == ref["root"] and again == ref["root"], rank
        assert sib == esib.tobytes() and paths == epaths.tobytes(), rank
        assert set(stage_ms) == {"encode", "relay", "digests", "merkle"}
        pieces.append(cols)
    merged = np.empty((len(idx), rows, 4), dtype=np.uint64)
    for pos, n, owner, local in relay_chain(rows, world, layout):
        merged[:, pos:pos + n] = pieces[owner][:, local:local + n]
    assert np.array_equal(merged, ecols)


# even / odd boundaries, a rank without rows, one row per workgroup at k = 4096, folded k = 8192, the four-block layout, plane groups
@pytest.mark.parametrize("rows,k,layout,groups", [(20, 128, "contiguous", 1), (21, 128, "contiguous", 2), (1, 128, "contiguous", 1), (7, 4096, "contiguous", 1),
                                                  (5, 8192, "contiguous", 4), (20, 128, "blocks", 1), (12, 8192, "blocks", 1),
                                                  (43, 128, "round_robin:4", 1), (21, 8192, "round_robin:2", 1)])
def test_world2_on_one_gpu_matches_oracle(oracle, rows, k, layout, groups):
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), rows, k, layout, groups, out), nprocs=world, join=True)
    _check(oracle, out, world, rows, k, layout)


@pytest.mark.parametrize("world,rows,k,layout,groups", [(4, 21, 128, "contiguous", 1), (8, 12, 128, "contiguous", 2), (8, 7, 8192, "contiguous", 1),
                                                        (4, 44, 128, "blocks", 1), (8, 36, 128, "blocks", 1),
                                                        (4, 45, 128, "round_robin:3", 2), (8, 70, 128, "round_robin:4", 4), (8, 37, 8192, "round_robin:2", 0)])
def test_world4_and_world8_on_one_gpu(oracle, world, rows, k, layout, groups):
    if world <= 4:
        import torch.multiprocessing as mp
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), rows, k, layout, groups, out), nprocs=world, join=True)
    else:
        from thread_dist import run_ranks
        out = dict(enumerate(run_ranks(world, lambda rank, d: _rank_body(rank, d, rows, k, layout, groups))))
    _check(oracle, out, world, rows, k, layout)


def test_forced_chunks_overlap_evaluation_and_hash(oracle, monkeypatch):
    """the rank that holds the first rows hashes each evaluated chunk on the library's second stream beside the evaluation
    of the next (lg_stage_hash_rows is queued, not waited for): forced at a small size, odd chunk boundaries"""
    from ligero_amd.sharded import HipRelayBackend, RowRelayCommitter
    monkeypatch.setenv("LG_FORCE_CHUNKS", "3")
    rows, k = 23, 128
    pre = random_mont(616, rows * k).reshape(rows, k, 4)
    rc = RowRelayCommitter(lambda local: HipRelayBackend(local, k), rows, None)
    try:
        assert rc.be.pipeline_chunks() == 3
        for _ in range(3):
            assert rc.commit(pre) == oracle.encode_commit(pre, k, 8 * k, want_u=False)["root"]
    finally:
        rc.be.close()


def _s22_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))      # the ranks share this box's CPU quota (cap_host_threads)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from ligero_amd.sharded import HipRelayBackend, RowRelayCommitter
        rows, k = 20068, 8192
        # (four plane groups: 16 384 columns per hop and launch -- the four-lanes-per-column kernel with its state parked and resumed)
        rc = RowRelayCommitter(lambda local: HipRelayBackend(local, k, device=0), rows, dist, plane_groups=4)
        try:
            (a, n), = rc.row_ranges()
            root = rc.commit(bench.shard_rows_of_seeded_matrix(bench.LARGE_SEED, k, a, a + n))
            out[rank] = (root.hex(), dict(rc.stage_ms))
        finally:
            rc.be.close()
    finally:
        dist.destroy_process_group()


def test_full_size_s22_at_world2_equals_the_golden_root():
    """BASELINE configs[3] (20 068 x 8192 -> 65 536, U = 42 GB) row-sharded over two ranks that share this box's GPU: each keeps
    half of the rows end to end (21 GB of U), 5.2 MB of Blake2s states cross once, the root is the committed golden one"""
    import torch.multiprocessing as mp
    gold = json.load(open(os.path.join(GOLDEN, "large_roots.json")))["s22"]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_s22_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in range(2):
        assert out[rank][0] == gold["root"], rank
    print("s22 row-relay at world 2 (one GPU, gloo):", out[0][1], out[1][1])


def test_error_behaviour_of_the_relay_entry_points():
    """bad ranges, a wrong row count for the rank, missing callbacks: a status, never a launch (include/ligero_hip.h)"""
    import ctypes
    from ligero_amd import _ffi
    from ligero_amd.sharded import HipRelayBackend, _LgComm
    L = _ffi.lib()
    be = HipRelayBackend(6, 128)
    try:
        ctx = be.c._ctx
        allp = (1 << be.nplanes) - 1
        be.stage_interpolate(random_mont(1, 6 * 128).reshape(6, 128, 4), 0, 6)
        be.stage_evaluate_rows(0, 6)
        B = _ffi.LG_ERR_BAD_ARG
        assert L.lg_stage_hash_rows(ctx, allp, 0, 0, 0, 6) == B                      # no rows
        assert L.lg_stage_hash_rows(ctx, allp, 4, 3, 0, 6) == B                      # rows [4, 7) of 6
        assert L.lg_stage_hash_rows(ctx, allp, 0, 6, 1, 6) == B                      # positions [1, 7) of a 6-row column
        assert L.lg_stage_hash_rows(ctx, allp, 0, 6, 2**64 - 3, 6) == B              # (no wrap-around of col_pos + nrows)
        assert L.lg_stage_hash_rows(ctx, allp, 0, 6, 0, 2**59) == B                  # longer than Blake2s' byte counter
        assert L.lg_stage_hash_rows(ctx, 1 << be.nplanes, 0, 6, 0, 6) == B           # a plane that does not exist
        assert L.lg_stage_hash_rows(None, allp, 0, 6, 0, 6) == B
        # the one-call commit: this context has 6 rows, rank 0 of 2 over 20 rows keeps 10
        comm = _LgComm(world=2, rank=0, flags=0, user=None)
        assert L.lg_commit_row_relay(ctx, ctypes.byref(comm), 20, _ffi.LG_RELAY_CONTIGUOUS, 0, None) == B      # world 2 without callbacks
        one = _LgComm(world=1, rank=0, flags=0, user=None)
        assert L.lg_commit_row_relay(ctx, ctypes.byref(one), 20, _ffi.LG_RELAY_CONTIGUOUS, 0, None) == _ffi.LG_ERR_STATE
        assert b"keeps 20 of the 20 rows" in L.lg_last_error(ctx)
        assert L.lg_commit_row_relay(ctx, ctypes.byref(one), 6, 7, 0, None) == B                                  # no such layout
        bad = _LgComm(world=2, rank=2, flags=0, user=None)
        assert L.lg_commit_row_relay(ctx, ctypes.byref(bad), 6, _ffi.LG_RELAY_CONTIGUOUS, 0, None) == B
        assert L.lg_commit_row_relay(ctx, None, 6, _ffi.LG_RELAY_CONTIGUOUS, 0, None) == B
        # ... and the context still works afterwards
        be.stage_hash_rows(0, be.nplanes, 0, 6, 0, 6)
        be.stage_merkle()
        assert len(be.root()) == 32
    finally:
        be.close()
