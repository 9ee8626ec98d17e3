"""GPU tests of prove / verify AGAINST THE ORACLE: every prover of the product -- single, host-transcript batch,
device-transcript batch (the sharded prover: tests/test_gpu_sharded_prover.py) -- must make the proof oracle/model_prover.py
makes from the same statement, byte for byte in all ten fields (tests/golden/proofs.json holds the oracle's fingerprints;
tests/golden/make_golden_proofs.py wrote them); the oracle's verify must accept GPU-made proofs and reject tampered ones; the
product's verify must accept oracle-made proofs.  The oracle is a big-int restatement of /root/reference/src/ligero/mod.rs
457-578, 613-996 and src/utils.rs:23-55 that shares no code with the product, so a slip shared by the product's host and device
transcripts can no longer hide.  (Against bytes of the Rust crate itself parity stays UNPINNED: no cargo here.)"""
import os

import numpy as np
import pytest

import proof_fp
from conftest import GOLDEN
from prover_hooks import tamper

pytestmark = pytest.mark.gpu
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def hp_circuit(hp, mc):
    """the product's ArithmeticCircuit with the node list of a model circuit (same builders, same order)"""
    c = hp.ArithmeticCircuit()
    for i, nd in enumerate(mc.nodes):
        if nd[0] == "V":
            got = c.new_variable_with_label(nd[1])
        elif nd[0] == "C":
            got = c.constant(hp.fr_mont(nd[1]))
        elif nd[0] == "A":
            got = c.add(nd[1], nd[2])
        else:
            got = c.mul(nd[1], nd[2])
        assert got == i
    return c


def mont_rows(hp, values):
    return np.stack([hp.fr_mont(v) for v in values])


def model_case(name):
    from oracle import model as M
    from oracle import model_prover as MP
    if name == "multiplication":
        return MP.r1cs_circuit(os.path.join(GOLDEN, "multiplication.r1cs"), [1, 33, 3, 11])
    if name == "poseidon":
        return MP.r1cs_circuit(os.path.join(GOLDEN, "poseidon.r1cs"), M.load_witness_json(os.path.join(GOLDEN, "poseidon_witness.json")))
    base = name.replace("_invalid", "")
    c, o, va = {"lemniscate": MP.lemniscate_circuit, "determinant": MP.determinant_circuit, "multioutput": MP.multioutput_circuit}[base]()
    if name.endswith("_invalid"):
        va = [(va[0][0], (va[0][1] + 1) % P)] + va[1:]
    return c, o, va


def product_case(hp, name):
    """-> (instance, prove(prover) -> Proof) of a named case for the product's provers"""
    mc, outs, va = model_case(name)
    inst = hp.LigeroInstance(hp_circuit(hp, mc), outputs=outs)
    vals = mont_rows(hp, [v for _, v in va])
    if isinstance(va[0][0], str):
        return inst, lambda prover: prover.prove_with_labels([s for s, _ in va], vals)
    return inst, lambda prover: prover.prove([i for i, _ in va], vals)


@pytest.mark.parametrize("name", ["lemniscate", "lemniscate_invalid", "determinant", "determinant_invalid", "multioutput", "multiplication", "poseidon"])
def test_single_prover_makes_the_oracles_proof(name):
    """the reference's own prove-and-verify cases (src/ligero/tests.rs:186-415): byte for byte the oracle's proof, and the product's
    verifier says what the oracle's said"""
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import LigeroProver
    want = proof_fp.golden()["cases"][name]
    inst, prove = product_case(hp, name)
    with LigeroProver(inst) as prover:
        proof = prove(prover)
        assert (inst.m, inst.k, inst.n, inst.t) == tuple(want["dims"][d] for d in "mknt")
        fp = proof_fp.fingerprint(proof)
        assert proof_fp.same(fp, want), proof_fp.diff(fp, want)
        assert prover.verify(proof) == want["accepted"]


@pytest.fixture(scope="module")
def poseidon_batch(oracle):
    from ligero_amd import host_pipeline as hp
    inst = hp.LigeroInstance(hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "poseidon.r1cs")))
    blob = open(os.path.join(GOLDEN, "poseidon_witness_batch64.bin"), "rb").read()
    ws = [[int.from_bytes(blob[(i * 265 + j) * 32:(i * 265 + j + 1) * 32], "little") for j in range(265)] for i in range(64)]
    allv = np.stack([oracle.to_mont(oracle.ints_to_limbs(w[1:])) for w in ws])
    return inst, list(range(1, 265)), allv


def test_host_transcript_batch_makes_the_oracles_proofs(poseidon_batch):
    """throughput mode, transcript on host threads: all 64 proofs of BASELINE configs[4] equal the oracle's"""
    from ligero_amd.prover import LigeroBatchProver
    inst, idx, allv = poseidon_batch
    want = proof_fp.golden()["poseidon_batch64"]
    with LigeroBatchProver(inst, 64) as bp:
        proofs = bp.prove(idx, allv)
        for b in range(64):
            fp = proof_fp.fingerprint(proofs[b])
            assert proof_fp.same(fp, want[b]), (b, proof_fp.diff(fp, want[b]))


@pytest.mark.parametrize("B", [3, 64, 70])
def test_device_transcript_batch_makes_the_oracles_proofs(poseidon_batch, B):
    """throughput mode with Fiat-Shamir on the device (lg_prove_batch_queue): every proof of a batch that does not fill a wave, of
    one wave and of two equals the oracle's -- sponge, challenge draws, index sampling, polynomials, openings, paths"""
    from ligero_amd.prover import LigeroBatchProver
    inst, idx, allv = poseidon_batch
    want = proof_fp.golden()["poseidon_batch64"]
    sel = np.arange(B) % 64
    with LigeroBatchProver(inst, B, device_transcript=True) as bp:
        proofs = bp.prove(idx, allv[sel])
        for b in range(B):
            fp = proof_fp.fingerprint(proofs[b])
            assert proof_fp.same(fp, want[sel[b]]), (b, proof_fp.diff(fp, want[sel[b]]))
        # two batches in flight, read out of the arenas
        bp.submit(idx, allv[sel])
        bp.submit(idx, allv[sel[::-1]])
        first = bp.collect()
        assert proof_fp.same(proof_fp.fingerprint(first[B - 1]), want[sel[B - 1]])
        second = bp.collect()
        assert proof_fp.same(proof_fp.fingerprint(second[0]), want[sel[::-1][0]])


def _to_model(proof):
    from oracle import model_prover as MP
    info = proof.info()
    return MP.proof_from_field_bytes(proof.field_bytes(), info["column_len"], info["auth_path_len"])


@pytest.mark.parametrize("name", ["determinant", "poseidon"])
def test_oracle_verifier_on_gpu_made_proofs(name):
    """the oracle's verify (mod.rs:613-996 restated) accepts the GPU-made proof and rejects each of the twelve corruptions the
    product's own verifier is tested with"""
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import LigeroProver
    from oracle import model_prover as MP
    mc, outs, _ = model_case(name)
    lc = MP.LigeroCircuit(mc, outs)
    inst, prove = product_case(hp, name)
    with LigeroProver(inst) as prover:
        assert lc.verify(_to_model(prove(prover)), MP.test_sponge())
        for what, index in [(0, 5), (1, 17), (2, 0), (2, 100), (3, 3), (4, 0), (4, 1000), (5, 77), (6, 4242), (7, 0), (7, 333), (8, 9)]:
            bad = prove(prover)
            tamper(bad, what, index)
            assert not lc.verify(_to_model(bad), MP.test_sponge()), (what, index)
            assert not prover.verify(bad), (what, index)


@pytest.mark.parametrize("name", ["lemniscate", "lemniscate_invalid", "multioutput", "multiplication", "poseidon"])
def test_product_verifier_on_oracle_made_proofs(name):
    """verify() of the product takes a proof the ORACLE made (through lgp_proof_from_fields, both byte forms) and says what the
    oracle's verifier says; the proof survives the round trip through the handle"""
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import BYTES_MONTGOMERY, LigeroProver, Proof
    from oracle import model_prover as MP
    mc, outs, va = model_case(name)
    lc = MP.LigeroCircuit(mc, outs)
    mproof = lc.prove_with_labels(va, MP.test_sponge()) if isinstance(va[0][0], str) else lc.prove(va, MP.test_sponge())
    fb = MP.proof_field_bytes(mproof)
    column_len, path_len = 4 * lc.m, len(mproof["interleaved"]["paths"][0][2])
    handle = Proof.from_fields(fb, column_len, path_len)
    assert handle.field_bytes() == fb
    inst = hp.LigeroInstance(hp_circuit(hp, mc), outputs=outs)
    with LigeroProver(inst) as prover:
        assert prover.verify(handle) == (not name.endswith("_invalid"))
        again = Proof.from_fields(handle.field_bytes(BYTES_MONTGOMERY), column_len, path_len, form=BYTES_MONTGOMERY)
        assert again.field_bytes() == fb and prover.verify(again) == (not name.endswith("_invalid"))
    # malformed fields are refused, not read
    short = dict(fb)
    short["linear.columns"] = fb["linear.columns"][:-32]
    with pytest.raises(RuntimeError):
        Proof.from_fields(short, column_len, path_len)
    big = dict(fb)
    big["interleaved.preenc_u_lc"] = P.to_bytes(32, "little") + fb["interleaved.preenc_u_lc"][32:]
    with pytest.raises(RuntimeError):
        Proof.from_fields(big, column_len, path_len)


@pytest.mark.parametrize("block", range(4))
def test_random_circuits_whole_proofs_equal_the_c_oracle(block):
    """48 random circuits (oracle/model_prover.py random_circuit: the constant 1 first / elsewhere / absent, satisfied and not, 4 to
    ~1000 gates: k = 4 ... 64, i.e. every column opened and t = 155 of n = 256 / 512) through LigeroCircuit::new, the trace, the commit,
    the three sub-proofs and the transcript of the PRODUCT -- every proof equals, in all ten fields, the one the C oracle's
    reference-shaped prover makes (itself equal to the big-int model on such circuits, tests/test_oracle_prover.py), and both
    verifiers agree on it; every fourth circuit also goes through the two batch provers"""
    from ligero_amd import host_pipeline as hp
    from ligero_amd.prover import LigeroBatchProver, LigeroProver
    from oracle import binding as orc
    from oracle import model_prover as MP
    for seed in range(12 * block, 12 * block + 12):
        one = ("first", "middle", "absent")[seed % 3]
        sat = seed % 4 != 3
        mc, outs, va = MP.random_circuit(5000 + seed, nvars=1 + seed % 9, ngates=3 + (seed * 97) % 1000, one=one, satisfied=sat)
        lc = MP.LigeroCircuit(mc, outs)
        st = orc.Statement(lc)
        want = st.prove(va)
        inst = hp.LigeroInstance(hp_circuit(hp, mc), outputs=outs)
        assert (inst.m, inst.k, inst.n, inst.t) == (lc.m, lc.k, lc.n, lc.t), seed
        idx, vals = [i for i, _ in va], mont_rows(hp, [v for _, v in va])
        with LigeroProver(inst) as prover:
            proof = prover.prove(idx, vals)
            got = proof.field_bytes()
            assert got == want, (seed, [f for f in got if got[f] != want[f]])
            assert prover.verify(proof) == sat == st.verify(got), seed
        if seed % 4 == 0:
            other = [(i, (v * 3 + 1) % P) for i, v in va]                    # a second statement of the same circuit (unsatisfied)
            want2 = st.prove(other)
            allv = np.stack([vals, mont_rows(hp, [v for _, v in other]), vals])
            for device_transcript in (False, True):
                with LigeroBatchProver(inst, 3, device_transcript=device_transcript) as bp:
                    proofs = bp.prove(idx, allv)
                    assert proofs[0].field_bytes() == want and proofs[2].field_bytes() == want, (seed, device_transcript)
                    assert proofs[1].field_bytes() == want2, (seed, device_transcript)
