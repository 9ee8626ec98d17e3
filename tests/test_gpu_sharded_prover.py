"""GPU test of ONE proof made by several ranks (ligero_amd.prover.ShardedLigeroProver over lgp_sharded_prover_create;
DESIGN.md section 7): `world` processes share the one GPU of the test box, each with its row shard of preenc_u and its coset
planes of U; the collectives run over gloo (RCCL needs one GPU per rank -- every device call and both exchanges are the ones
the 8-GPU layout uses).  Every rank must end with the complete proof, field for field the proof the ordinary single-GPU prover
makes from the reference's Poseidon fixtures (golden u_root included), and the ordinary verifier must accept it."""
import json
import os
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))      # spawned ranks import tests/proof_fp.py


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _poseidon_case():
    """the reference's fixtures through the C++ host pipeline (no oracle in the workers: product code only)"""
    from ligero_amd import host_pipeline as hp
    circ = hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "poseidon.r1cs"))
    inst = hp.LigeroInstance(circ)
    w = hp.read_witness(os.path.join(GOLDEN, "poseidon_witness.json"))
    return inst, list(range(1, w.shape[0])), w[1:]


def _small_case():
    """src/ligero/tests.rs:245-266 by label: 16 rows of k = 4 -- row shards of 2 rows at world 8, t = n = 32 (every column opened)"""
    from ligero_amd import host_pipeline as hp
    c = hp.ArithmeticCircuit()
    x, y = c.new_variable_with_label("x"), c.new_variable_with_label("y")
    c1, c2, c3 = (c.constant(hp.fr_mont(v)) for v in (-8, -63, -6))
    x2, y3, xy = c.mul(x, x), c.pow(y, 3), c.add(x, y)          # the reference's order of construction: the golden "multioutput" case
    outs = [c.add(x2, c1), c.add(y3, c2), c.add(xy, c3)]
    return hp.LigeroInstance(c, outs), ["x", "y"], np.stack([hp.fr_mont(3), hp.fr_mont(4)])


def _rank_body(rank, world, dist, which, mode="coset"):
    from ligero_amd.prover import LigeroProver, ShardedLigeroProver, proofs_equal
    from proof_fp import fingerprint
    inst, names, vals = _poseidon_case() if which == "poseidon" else _small_case()
    by_label = isinstance(names[0], str)
    with ShardedLigeroProver(inst, dist, device=0, mode=mode) as sp:
        proof = sp.prove_with_labels(names, vals) if by_label else sp.prove(names, vals)
        again = sp.prove_with_labels(names, vals) if by_label else sp.prove(names, vals)        # the context is reused
        accepted_by_sharded = sp.verify(proof)
        res = {"root": proof.info()["u_root"], "info": {k: v for k, v in proof.info().items() if k != "u_root"},
               "again": proofs_equal(proof, again), "accepted_by_sharded": accepted_by_sharded, "fingerprint": fingerprint(proof)}
        if rank == world - 1:                                          # one rank compares with the unsharded prover
            with LigeroProver(inst) as single:
                ref = single.prove_with_labels(names, vals) if by_label else single.prove(names, vals)
                res["equal"] = proofs_equal(proof, ref)
                res["accepted"] = single.verify(proof)
            res["ref_root"] = ref.info()["u_root"]
        # a wrong witness: every rank proves it, nobody may accept it
        bad = vals.copy()
        bad[0, 0] ^= np.uint64(1)
        wrong = sp.prove_with_labels(names, bad) if by_label else sp.prove(names, bad)
        res["wrong_rejected"] = not sp.verify(wrong)
    return res


def _worker(rank, world, port, which, out, mode="coset"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))      # the ranks share this box's CPU quota (cap_host_threads)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out[rank] = _rank_body(rank, world, dist, which, mode)
    finally:
        dist.destroy_process_group()


# mode "relay" (round 3): rows end to end in the blocks layout, hash states handed on, sub-proof points = sums of per-rank partial
# sums, opened columns as row pieces -- the SAME proof, field for field; m = 86 over 4 ranks and m = 4 over 8 (four ranks without rows)
# trace "device" (round 4): every rank's rows of preenc_u come from a tracer on its device (lg_tracer_rows) instead of the host's
# evaluation of the whole circuit -- forced here (LG_DEVICE_TRACE=1: the estimate keeps circuits this small on the host; the
# 2^20-constraint proofs of tests/test_gpu_world8.py take it by themselves); ranks without rows included
@pytest.mark.parametrize("which,world,mode,trace", [("poseidon", 2, "coset", "host"), ("poseidon", 4, "coset", "host"), ("small", 8, "coset", "host"),
                                                    ("poseidon", 2, "relay", "host"), ("poseidon", 4, "relay", "host"), ("small", 8, "relay", "host"),
                                                    ("poseidon", 2, "coset", "device"), ("poseidon", 4, "relay", "device"),
                                                    ("small", 8, "coset", "device"), ("small", 8, "relay", "device")])
def test_sharded_proof_equals_the_single_gpu_proof(which, world, mode, trace, monkeypatch):
    monkeypatch.setenv("LG_DEVICE_TRACE", "1" if trace == "device" else "0")
    if world <= 4:                         # real gloo process groups; the GPU box admits at most six processes on its card,
        import torch.multiprocessing as mp
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), which, out, mode), nprocs=world, join=True)
    else:                                  # so world 8 is eight sharded provers on eight threads of this process (tests/thread_dist.py)
        from thread_dist import run_ranks
        out = dict(enumerate(run_ranks(world, lambda rank, dist: _rank_body(rank, world, dist, which, mode))))
    assert set(out.keys()) == set(range(world))
    last = out[world - 1]
    assert last["equal"], "the sharded proof differs from the single-GPU proof"
    assert last["accepted"]
    if which == "poseidon":
        assert last["ref_root"].hex() == json.load(open(os.path.join(GOLDEN, "vectors.json")))["poseidon"]["root"]
    # round 5: every rank's proof is the ORACLE's proof of this statement, byte for byte (tests/golden/proofs.json, made by
    # oracle/model_prover.py from the reference's source) -- not only the proof the single-GPU product prover makes
    import proof_fp
    want = proof_fp.golden()["cases"]["poseidon" if which == "poseidon" else "multioutput"]
    for rank in range(world):
        r = out[rank]
        assert proof_fp.same(r["fingerprint"], want), (rank, proof_fp.diff(r["fingerprint"], want))
        assert r["root"] == last["ref_root"], rank                        # every rank holds the complete, same proof
        assert r["info"] == last["info"], rank
        assert r["again"] and r["accepted_by_sharded"] and r["wrong_rejected"], (rank, r)


def _rccl_worker(out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from ligero_amd.prover import LigeroProver, ShardedLigeroProver, proofs_equal
        inst, idx, vals = _poseidon_case()
        with ShardedLigeroProver(inst, dist, device=0, collectives_at_world_1=True) as sp, LigeroProver(inst) as single:
            proof = sp.prove(idx, vals)
            ref = single.prove(idx, vals)
            out["equal"] = proofs_equal(proof, ref)
            out["accepted"] = single.verify(proof)
            out["error"] = sp.comm_error
            with ShardedLigeroProver(inst, dist, device=0, collectives_at_world_1=True, mode="relay") as rp:   # the relay's identity broadcast
                out["relay_equal"] = proofs_equal(rp.prove(idx, vals), ref)
                out["relay_error"] = rp.comm_error
    finally:
        dist.destroy_process_group()


def test_sharded_prover_over_rccl_at_world_1():
    """backend "nccl" = RCCL: the device all-gathers on the library's buffers and the host all-gathers staged through the GPU, as
    the multi-GPU run issues them (identities in a one-rank group)"""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    p = mp.get_context("spawn").Process(target=_rccl_worker, args=(out,))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert out["error"] is None and out["equal"] and out["accepted"]
    assert out["relay_error"] is None and out["relay_equal"]


def _failing_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))      # the ranks share this box's CPU quota (cap_host_threads)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ligero_amd.prover import ShardedLigeroProver
        inst, names, vals = _small_case()
        with ShardedLigeroProver(inst, dist, device=0) as sp:
            try:
                if rank == 1:
                    sp.prove_with_labels(names[:1], vals[:1])               # this rank "forgets" y: its trace fails
                else:
                    sp.prove_with_labels(names, vals)
                out[rank] = "no error"
            except RuntimeError as e:
                out[rank] = str(e)
            # the group is still usable: the next proof goes through on every rank
            out[f"next{rank}"] = sp.verify(sp.prove_with_labels(names, vals))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_does_not_leave_the_others_inside_a_collective():
    """before every exchange the ranks trade one status word: when one rank's step throws, every rank throws instead of
    waiting for it forever"""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_failing_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert "Uninitialised variable" in out[1]
    assert "failed on rank 1" in out[0]
    assert out["next0"] and out["next1"]
