"""GPU tests of prove / verify (ligero_amd/host/prover.hpp through libligero_prover.so): the complete
LigeroCircuit::prove()/verify() flow of src/ligero/mod.rs:435-455, 613-644 from the reference's
fixtures, with every device sub-proof checked by the verifier's algebra and every opened column by
its Merkle path.  Mirrors the reference's test_poseidon (src/ligero/tests.rs:380-416: prove, then
assert verify) and adds the negative cases that test does not have.  The transcript is UNPINNED
against the Rust crates (transcript.hpp); the commitment root IS the golden one."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def poseidon(model, oracle):
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import LigeroProver
    circ = hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "poseidon.r1cs"))
    inst = hp.LigeroInstance(circ)
    w = model.load_witness_json(os.path.join(GOLDEN, "poseidon_witness.json"))
    vals = oracle.to_mont(oracle.ints_to_limbs(w[1:]))
    prover = LigeroProver(inst)
    yield inst, prover, list(range(1, len(w))), vals
    prover.close()


def test_poseidon_prove_then_verify(poseidon, vectors):
    inst, prover, idx, vals = poseidon
    proof = prover.prove(idx, vals)
    info = proof.info()
    assert info["u_root"].hex() == vectors["poseidon"]["root"]          # the committed golden root
    assert info["preenc_u_lc"] == inst.k
    assert 0 < info["linear_poly"] <= 2 * inst.k - 1 and 0 < info["quadratic_poly"] <= 2 * inst.k - 1
    assert info["opened_columns"] == inst.t == 156 and info["column_len"] == 4 * inst.m and info["auth_path_len"] == 9
    assert prover.verify(proof)
    # deterministic: the same statement gives the same proof and it still verifies
    again = prover.prove(idx, vals)
    assert again.info() == info and prover.verify(again)


@pytest.mark.parametrize("what,index", [(0, 5), (1, 17), (2, 0), (2, 100), (3, 3), (4, 0), (4, 1000), (5, 77), (6, 4242), (7, 0), (7, 333), (8, 9)])
def test_tampered_proofs_are_rejected(poseidon, what, index):
    """one flipped item anywhere -- root, a polynomial coefficient, an opened column element, a path
    digest, a leaf index -- and verify() must say no"""
    _, prover, idx, vals = poseidon
    proof = prover.prove(idx, vals)
    proof.tamper(what, index)
    assert not prover.verify(proof)


def test_unsatisfied_circuit_is_rejected(poseidon):
    """a wrong witness still yields a proof object (the reference's prove() does not check outputs), but the
    linear / quadratic tests fail"""
    inst, prover, idx, vals = poseidon
    bad = vals.copy()
    bad[10] = bad[11]                       # some private wire gets another wire's value
    pre, ok = inst.build_preenc_u(idx, bad)
    assert not ok
    assert not prover.verify(prover.prove(idx, bad))


def test_small_handbuilt_circuit(oracle):
    """x^3 + x + 5 = 35 in the reference's builder API (src/arithmetic_circuit/tests.rs style): output x^3 + x + 5 - 34,
    which is 1 for x = 3"""
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import LigeroProver

    def mont(v):
        return oracle.to_mont(oracle.ints_to_limbs([v % oracle_p()]))[0]

    def oracle_p():
        return 21888242871839275222246405745257275088548364400416034343698204186575808495617

    c = hp.ArithmeticCircuit()
    x = c.new_variable()
    x3 = c.mul(c.mul(x, x), x)
    s = c.add(c.add(x3, x), c.constant(mont(5)))
    out = c.add(s, c.constant(mont(-34)))
    inst = hp.LigeroInstance(c, outputs=[out])
    with LigeroProver(inst) as prover:
        good = prover.prove([x], np.stack([mont(3)]))
        assert prover.verify(good)
        assert good.info()["opened_columns"] == inst.t
        assert not prover.verify(prover.prove([x], np.stack([mont(4)])))


def test_batch_prover_matches_single_prover(poseidon, oracle, vectors):
    """throughput mode (BASELINE configs[4]): 64 Poseidon proofs per call through batch-wide device calls and
    host threads; every proof carries its golden root, verifies, and equals what the single prover gives"""
    from ligero_amd.prover import LigeroBatchProver
    inst, prover, idx, vals = poseidon
    blob = open(os.path.join(GOLDEN, "poseidon_witness_batch64.bin"), "rb").read()
    B = 64
    ws = [[int.from_bytes(blob[(i * 265 + j) * 32:(i * 265 + j + 1) * 32], "little") for j in range(265)] for i in range(B)]
    allv = np.stack([oracle.to_mont(oracle.ints_to_limbs(w[1:])) for w in ws])
    with LigeroBatchProver(inst, B) as bp:
        assert bp.threads >= 1
        proofs = bp.prove(idx, allv)
        assert len(proofs) == B
        assert [p.info()["u_root"].hex() for p in proofs] == vectors["poseidon_batch64_roots"]
        for b in (0, 1, 31, 63):
            assert prover.verify(proofs[b]), b
            single = prover.prove(idx, allv[b])
            assert single.info() == proofs[b].info()
        # tampering one proof of the batch does not go unnoticed
        proofs[5].tamper(6, 99)
        assert not prover.verify(proofs[5])
    with LigeroBatchProver(inst, 3, threads=1) as bp3:           # odd batch, single-threaded host side
        for p in bp3.prove(idx, allv[:3]):
            assert prover.verify(p)
        # borrowed views of the prover's reused storage: same proofs, read-only
        views = bp3.prove(idx, allv[3:6], copy=False)
        assert [v.info()["u_root"].hex() for v in views] == vectors["poseidon_batch64_roots"][3:6]
        assert all(prover.verify(v) for v in views)
        with pytest.raises(RuntimeError):
            views[0].tamper(1, 0)
        views2 = bp3.prove(idx, allv[:3], copy=False)               # storage is reused: the earlier views now show these
        assert [v.info()["u_root"].hex() for v in views2] == vectors["poseidon_batch64_roots"][:3]
