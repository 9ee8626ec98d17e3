"""CPU tests of the C oracle's whole prover / verifier (oracle/ligero_oracle.c orc_prove / orc_verify: the serial reference-shaped
CPU baseline bench.py times beside proofs/sec) against the big-int model (oracle/model_prover.py) and the committed fingerprints:
two independent restatements of /root/reference/src/ligero/mod.rs:457-996 + src/utils.rs:23-55 must agree byte for byte."""
import copy
import hashlib
import json
import os
import random

import pytest

from conftest import GOLDEN
from oracle import binding as orc
from oracle import model as M
from oracle import model_prover as MP
from oracle import transcript_model as T


@pytest.mark.parametrize("seed", [bytes(32), bytes(range(32)), bytes([0xFF] * 32)])
def test_prng_helpers(seed):
    assert orc.limbs_to_ints(orc.field_elements_from_seed(seed, 500)) == T.field_elements_from_seed(seed, 500)
    for n, t in ((1024, 156), (32, 32), (32768, 156), (64, 40), (8, 0)):
        assert orc.distinct_indices_from_seed(seed, n, t) == T.distinct_indices_from_seed(seed, n, t)


def test_sponge_scripts():
    rng = random.Random(5)
    for _ in range(20):
        ops, sp, want = [], MP.test_sponge(), []
        for _ in range(rng.randrange(1, 12)):
            kind = rng.choice(("bytes", "elems", "seed"))
            if kind == "bytes":
                data = bytes(rng.randrange(256) for _ in range(rng.choice((0, 1, 23, 31, 32, 54, 55, 100))))
                ops.append(("bytes", data))
                sp.absorb_bytes(data)
            elif kind == "elems":
                e = [rng.randrange(M.P) for _ in range(rng.choice((0, 1, 2, 3, 7, 128)))]
                ops.append(("elems", e))
                sp.absorb_elements(e)
            else:
                ops.append(("seed",))
                want.append(sp.squeeze_bytes(32))
        assert orc.sponge_script(ops) == want


def _model_proof(lc, va):
    return lc.prove_with_labels(va, MP.test_sponge()) if isinstance(va[0][0], str) else lc.prove(va, MP.test_sponge())


@pytest.mark.parametrize("which", ["lemniscate", "determinant", "multioutput", "multiplication"])
def test_c_prover_makes_the_models_proof(which):
    if which == "multiplication":
        c, outs, va = MP.r1cs_circuit(os.path.join(GOLDEN, "multiplication.r1cs"), [1, 33, 3, 11])
    else:
        c, outs, va = {"lemniscate": MP.lemniscate_circuit, "determinant": MP.determinant_circuit, "multioutput": MP.multioutput_circuit}[which]()
    lc = MP.LigeroCircuit(c, outs)
    st = orc.Statement(lc)
    proof = _model_proof(lc, va)
    fb = st.prove(va)
    assert fb == MP.proof_field_bytes(proof)
    assert st.verify(fb) and lc.verify(proof, MP.test_sponge())
    # an unsatisfying assignment: the same (rejected) proof from both
    bad_va = [(va[0][0], va[0][1] + 1)] + list(va[1:])
    bad = st.prove(bad_va)
    assert bad == MP.proof_field_bytes(_model_proof(lc, bad_va)) and not st.verify(bad)
    # every tampered field is rejected by the C verifier as by the model's
    from test_model_prover import _tampers
    for what, tp in _tampers(proof):
        assert not st.verify(MP.proof_field_bytes(tp)), what
        assert not lc.verify(tp, MP.test_sponge()), what
    if which != "multioutput":
        with pytest.raises(RuntimeError, match="Uninitialised variable"):
            st.prove(va[:1])


def test_c_prover_on_poseidon_equals_the_golden_fingerprint():
    w = M.load_witness_json(os.path.join(GOLDEN, "poseidon_witness.json"))
    c, outs, va = MP.r1cs_circuit(os.path.join(GOLDEN, "poseidon.r1cs"), w)
    st = orc.Statement(MP.LigeroCircuit(c, outs))
    fb = st.prove(va)
    want = json.load(open(os.path.join(GOLDEN, "proofs.json")))["cases"]["poseidon"]
    assert all(hashlib.sha256(fb[f]).hexdigest() == want[f] for f in orc.FIELDS)
    assert st.verify(fb)
    flipped = dict(fb)
    flipped["quadratic.columns"] = fb["quadratic.columns"][:64] + bytes([fb["quadratic.columns"][64] ^ 1]) + fb["quadratic.columns"][65:]
    assert not st.verify(flipped)


@pytest.mark.parametrize("log_n", [10, 14])
def test_large_family_goldens_reproduce(log_n):
    """tests/golden/proofs_large.json (the repeated-squaring family of BASELINE configs[2]; 2^20 itself takes ten minutes to make and is
    checked on the GPU box against the product): the small members regenerate from the oracle, serial and threaded alike"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgl", os.path.join(GOLDEN, "make_golden_proofs_large.py"))
    mgl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mgl)
    gold = json.load(open(os.path.join(GOLDEN, "proofs_large.json")))
    assert gold[f"s{log_n}"] == mgl.fingerprint(log_n, threads=1 if log_n == 10 else 4)
    if "s20" in gold:
        assert gold["s20"]["dims"] == {"m": 2509, "k": 4096, "n": 32768, "t": 156} and gold["s20"]["accepted"]


@pytest.mark.parametrize("seed", range(24))
def test_random_circuits_model_and_c_prover_agree(seed):
    """random circuits LigeroCircuit::new accepts (the constant 1 first, elsewhere or absent: the three paths of mod.rs:160-169;
    satisfied and unsatisfied outputs; k = 4 ... 64, so t = n and t = 155 < n): the two restatements make the same proof bytes and
    reach the same verdicts"""
    one = ("first", "middle", "absent")[seed % 3]
    sat = seed % 4 != 3
    c, outs, va = MP.random_circuit(1000 + seed, nvars=1 + seed % 7, ngates=3 + (seed * 53) % 900, one=one, satisfied=sat)
    lc = MP.LigeroCircuit(c, outs)
    proof = lc.prove(va, MP.test_sponge())
    assert lc.verify(proof, MP.test_sponge()) == sat
    st = orc.Statement(lc)
    fb = st.prove(va)
    assert fb == MP.proof_field_bytes(proof)
    assert st.verify(fb) == sat
