#!/usr/bin/env python3
"""Writes tests/golden/proofs.json: fingerprints of WHOLE LigeroProofs made by the oracle (oracle/model_prover.py, a big-int
restatement of /root/reference/src/ligero/mod.rs:457-578 and its sub-protocols), one per case the reference's own tests prove
(src/ligero/tests.rs:186-415) plus the 64 Poseidon statements of BASELINE configs[4].  Reads only files under tests/golden/:

    python tests/golden/make_golden_proofs.py        (about two minutes on 8 cores)

A fingerprint is the SHA-256 and the byte length of each of the ten proof fields (oracle/model_prover.py proof_field_bytes: the
layout include/ligero_prover.h lgp_proof_field_bytes exports), so a GPU-made proof is compared byte for byte without storing
5 MB per proof.  MODEL-DERIVED, like vectors.json: the Rust crate cannot be run here and its tests hold no proof bytes (PARITY
UNPINNED); what this pins is the product against an independent restatement of the reference's source."""
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import model as M  # noqa: E402
from oracle import model_prover as MP  # noqa: E402


def batch_witness(i):
    blob = open(os.path.join(HERE, "poseidon_witness_batch64.bin"), "rb").read()
    return [int.from_bytes(blob[(i * 265 + j) * 32:(i * 265 + j + 1) * 32], "little") for j in range(265)]


def case(name):
    """-> (circuit, outputs, assignment) of a named case"""
    if name == "lemniscate":
        return MP.lemniscate_circuit()
    if name == "lemniscate_invalid":                     # invalid_assignment[0].1 += F::ONE (tests.rs:161-162)
        c, o, va = MP.lemniscate_circuit()
        return c, o, [(va[0][0], (va[0][1] + 1) % M.P)] + va[1:]
    if name == "determinant":
        return MP.determinant_circuit()
    if name == "determinant_invalid":
        c, o, va = MP.determinant_circuit()
        return c, o, [(va[0][0], (va[0][1] + 1) % M.P)] + va[1:]
    if name == "multioutput":
        return MP.multioutput_circuit()
    if name == "multiplication":
        return MP.r1cs_circuit(os.path.join(HERE, "multiplication.r1cs"), [1, 33, 3, 11])
    if name == "poseidon":
        return MP.r1cs_circuit(os.path.join(HERE, "poseidon.r1cs"), M.load_witness_json(os.path.join(HERE, "poseidon_witness.json")))
    if name.startswith("poseidon_batch64/"):
        return MP.r1cs_circuit(os.path.join(HERE, "poseidon.r1cs"), batch_witness(int(name.split("/")[1])))
    raise KeyError(name)


def prove_case(name):
    circ, outs, va = case(name)
    lc = MP.LigeroCircuit(circ, outs)
    by_label = isinstance(va[0][0], str)
    proof = lc.prove_with_labels(va, MP.test_sponge()) if by_label else lc.prove(va, MP.test_sponge())
    fp = MP.proof_fingerprint(proof)
    fp["dims"] = {"m": lc.m, "k": lc.k, "n": lc.n, "t": lc.t}
    fp["accepted"] = lc.verify(proof, MP.test_sponge())
    return name, fp


def main():
    names = ["lemniscate", "lemniscate_invalid", "determinant", "determinant_invalid", "multioutput", "multiplication", "poseidon"]
    names += [f"poseidon_batch64/{i}" for i in range(64)]
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        res = dict(pool.map(prove_case, names, chunksize=1))
    assert res["poseidon"] == res["poseidon_batch64/0"], "witness 0 of the batch is the fixture witness"
    for n in names:
        assert res[n]["accepted"] == (not n.endswith("_invalid")), n
    out = {"provenance": "MODEL-DERIVED by tests/golden/make_golden_proofs.py from oracle/model_prover.py (restatement of "
                         "src/ligero/mod.rs:457-578, 613-996, src/utils.rs:23-55); not produced by the Rust crate (PARITY UNPINNED)",
           "fields": list(MP.FIELDS),
           "cases": {n: res[n] for n in names if "/" not in n},
           "poseidon_batch64": [res[f"poseidon_batch64/{i}"] for i in range(64)]}
    with open(os.path.join(HERE, "proofs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote proofs.json:", len(names), "proofs")


if __name__ == "__main__":
    main()
