"""The 8-GPU configuration of BASELINE configs[3] in front of the driver: the FULL-SIZE 2^22-constraint
shape (20 068 x 8192 -> 65 536, U = 42 GB) over EIGHT rank contexts of the real device backend -- each holding what its GPU would
hold: 2 of 16 coset planes, or 2508 / 2510 of the rows -- on threads of the test process (tests/thread_dist.py: the GPU box admits
six processes per card; the collectives are the thread harness', everything else is the code an 8-GPU node runs).  Both multi-GPU
commits in both of their variants, every root against the golden the oracle's streamed restatement produced
(tests/golden/large_roots.json), and one opened column per owner re-hashed against its leaf and walked up its path.  Follows
src/ligero/mod.rs:521-551 (commit) and 935-955 (openings)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
GOLDEN = os.path.join(HERE, "golden")

pytestmark = pytest.mark.gpu

WORLD, ROWS, K = 8, 20068, 8192
N = 8 * K


@pytest.fixture(scope="module")
def s22_matrix():
    import bench
    return bench.shard_rows_of_seeded_matrix(bench.LARGE_SEED, K, 0, ROWS)      # 5.3 GB, made once for the four variants


def _take(m, ranges):
    return np.ascontiguousarray(np.concatenate([m[a:a + n] for a, n in ranges])) if ranges else None


@pytest.mark.parametrize("pieces", [1, 4])
def test_full_size_s22_coset_sharded_over_eight_rank_contexts(oracle, model, s22_matrix, pieces):
    """lg_commit_sharded: rank g interpolates the rows it owns (one shard, or sub-block g of each of 4 exchange pieces), the
    coefficient rows are all-gathered in place, rank g evaluates and hashes planes 2g, 2g + 1 for ALL rows, the digests are
    all-gathered, every rank builds the tree.  Root = golden on every rank; rank g opens a column of each of its planes:
    Blake2s of the column = its leaf, the path leads to the root; planes 0 and 8 hold the message itself (systematic code)."""
    from ligero_amd.sharded import CosetShardedCommitter, HipStageBackend
    from thread_dist import run_ranks
    gold = json.load(open(os.path.join(GOLDEN, "large_roots.json")))["s22"]

    def body(rank, dist):
        be = HipStageBackend(ROWS, K, device=0, world=WORLD, rank=rank, pieces=pieces)
        try:
            cm = CosetShardedCommitter(be, dist, exchange_pieces=pieces)
            assert cm.native and len(cm.row_ranges()) == pieces
            root = cm.commit(_take(s22_matrix, cm.row_ranges()))
            idx = [16 * (1000 + 37 * rank) + 2 * rank, 16 * 4095 + 2 * rank + 1]       # one column of each owned plane
            got = cm.open_columns(idx)
            leaves = be.leaves_bytes().cpu().numpy()
            for j in idx:
                col, sib, path = got[j]
                leaf = oracle.col_hash(col)
                assert leaf == leaves[j].tobytes(), (rank, j)
                assert model.merkle_verify(root, leaf, j, sib.tobytes(), [x.tobytes() for x in path]), (rank, j)
                if j % 8 == 0:
                    assert np.array_equal(col, s22_matrix[:, j // 8]), (rank, j)     # systematic: U[i][8 q] = preenc_u[i][q] (ranks 0 and 4)
            return root.hex()
        finally:
            be.close()

    roots = run_ranks(WORLD, body, timeout=600)
    assert roots == [gold["root"]] * WORLD


@pytest.mark.parametrize("layout,groups", [("contiguous", 4), ("blocks", 1), ("round_robin:4", 4)])
def test_full_size_s22_row_relay_over_eight_rank_contexts(oracle, model, s22_matrix, layout, groups):
    """lg_commit_row_relay: rank g keeps its rows END TO END (all 16 planes of 2508 / 2510 rows, or of its share of each of the
    X, Y, Z, W blocks, or four ranges dealt round robin: 32 hops of four plane groups each, the evaluation of a rank's next
    range beside the hops of its current one), the columns' Blake2s states travel from rank to rank (four plane groups in flight on the contiguous layout),
    the last rank broadcasts the digests.  Root = golden on every rank; an opened column is the ranks' row pieces put
    together in row order: Blake2s of it = its leaf, the path leads to the root."""
    from ligero_amd.sharded import HipRelayBackend, RowRelayCommitter
    from thread_dist import run_ranks
    gold = json.load(open(os.path.join(GOLDEN, "large_roots.json")))["s22"]
    idx = [0, 16 * 77 + 5, N - 1]

    def body(rank, dist):
        rc = RowRelayCommitter(lambda local: HipRelayBackend(local, K, device=0), ROWS, dist, plane_groups=groups, layout=layout)
        try:
            root = rc.commit(_take(s22_matrix, rc.row_ranges()))
            cols, sib, paths = rc.open_columns(idx)
            leaves = rc.be.leaves_bytes().cpu().numpy()
            return root.hex(), cols, sib, paths, leaves, rc
        finally:
            rc.be.close()

    out = run_ranks(WORLD, body, timeout=600)
    assert [o[0] for o in out] == [gold["root"]] * WORLD
    whole = out[0][5].assemble_columns([o[1] for o in out])
    root = bytes.fromhex(gold["root"])
    for i, j in enumerate(idx):
        leaf = oracle.col_hash(whole[i])
        for o in out:                                              # the tree is replicated: every rank serves the same leaf and path
            assert leaf == o[4][j].tobytes()
            assert model.merkle_verify(root, leaf, j, o[2][i].tobytes(), [x.tobytes() for x in o[3][i]])
        if j % 8 == 0:
            assert np.array_equal(whole[i], s22_matrix[:, j // 8])     # systematic: U[i][8 q] = preenc_u[i][q]


@pytest.mark.parametrize("mode", ["coset", "relay"])
def test_s20_proof_sharded_over_eight_rank_contexts_equals_the_single_gpu_proof(tmp_path, mode):
    """BASELINE configs[2]'s circuit (the 2^20-constraint repeated-squaring R1CS: 5.2 M nodes, m = 2509, k = 4096) proved by EIGHT
    sharded provers on threads -- the coset mode (planes dealt to ranks, sub-proof points from the ranks that hold them) and the
    row relay (rows of each of the X, Y, Z, W blocks dealt to ranks, partial sums added) -- every rank's proof equal field for
    field to the proof one ordinary prover makes, and accepted by it (src/ligero/mod.rs:457-578)."""
    import importlib.util
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import LigeroProver, ShardedLigeroProver, proofs_equal
    from thread_dist import run_ranks
    spec = importlib.util.spec_from_file_location("gen_rs", os.path.join(ROOT, "tools", "gen_repeated_squaring_r1cs.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    r1cs, wtns = str(tmp_path / "rs20.r1cs"), str(tmp_path / "rs20.wtns")
    gen.write_r1cs(r1cs, 20)
    gen.write_wtns(wtns, gen.witness(20, 1))
    inst = hp.LigeroInstance(hp.ArithmeticCircuit.from_r1cs(r1cs))
    assert (inst.m, inst.k, inst.n, inst.t) == (2509, 4096, 32768, 156)
    w = hp.read_witness(wtns)
    idx = np.arange(1, w.shape[0], dtype=np.uint64)
    with LigeroProver(inst) as single:
        ref = single.prove(idx, w[1:])
        assert single.verify(ref)

        def body(rank, dist):
            with ShardedLigeroProver(inst, dist, device=0, mode=mode) as sp:
                return sp.prove(idx, w[1:])

        proofs = run_ranks(WORLD, body, timeout=600)
        for rank, p in enumerate(proofs):
            assert proofs_equal(ref, p), rank
        assert single.verify(proofs[WORLD - 1])
        # round 5: ... and it is the ORACLE's proof of this statement, byte for byte (tests/golden/proofs_large.json, made by
        # oracle/model_prover.py + oracle/ligero_oracle.c without the product): the eight-rank proof of BASELINE configs[2] is pinned by it
        import json
        import proof_fp
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs_large.json")))["s20"]
        for p in (proofs[0], proofs[WORLD - 1]):
            fp = proof_fp.fingerprint(p)
            assert proof_fp.same(fp, gold), proof_fp.diff(fp, gold)
