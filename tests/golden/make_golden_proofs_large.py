#!/usr/bin/env python3
"""Writes tests/golden/proofs_large.json: fingerprints of the WHOLE LigeroProof of the synthetic repeated-squaring R1CS of BASELINE
configs[2] (2^20 constraints: m 2509, k 4096, n 32768, t 156) and of two small members of the family (2^10, 2^14: reproduced by the CPU
suite), made by the oracle alone:

  tools/gen_repeated_squaring_r1cs.py   the .r1cs and the witness (test tooling, not product)
  oracle/model.py                       from_constraint_system (src/arithmetic_circuit/mod.rs:455-520)
  oracle/model_prover.py                LigeroCircuit::new (src/ligero/mod.rs:147-433): dimensions, the matrix A
  oracle/ligero_oracle.c orc_prove      prove_inner with the test_sponge() transcript (equal to the big-int model byte for byte on
                                        every case of tests/test_oracle_prover.py; here with its row loops on several threads --
                                        the same bytes, orc_prover_set_threads) and orc_verify on the result

    python tests/golden/make_golden_proofs_large.py [log_n ...]        (default 10 14 20; 2^20 takes about ten minutes and 25 GB)

MODEL-DERIVED like the other goldens (PARITY UNPINNED against the Rust crate)."""
import hashlib
import importlib.util
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import binding as orc  # noqa: E402
from oracle import model as M  # noqa: E402
from oracle import model_prover as MP  # noqa: E402


def generator():
    spec = importlib.util.spec_from_file_location("gen_rs", os.path.join(ROOT, "tools", "gen_repeated_squaring_r1cs.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


def statement(log_n: int, seed: int = 1):
    """-> (orc.Statement, assignment) of the 2^log_n-constraint circuit"""
    gen = generator()
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "rs.r1cs")
        gen.write_r1cs(path, log_n)
        wit = gen.witness(log_n, seed)
        circ, outs, va = MP.r1cs_circuit(path, wit)
    lc = MP.LigeroCircuit(circ, outs)
    return orc.Statement(lc), va, (lc.m, lc.k, lc.n, lc.t)


def fingerprint(log_n: int, threads: int):
    t0 = time.time()
    st, va, dims = statement(log_n)
    t1 = time.time()
    orc.lib().orc_prover_set_threads(threads)
    try:
        fb = st.prove(va)
        t2 = time.time()
        ok = st.verify(fb)
    finally:
        orc.lib().orc_prover_set_threads(1)
    fp = {name: hashlib.sha256(fb[name]).hexdigest() for name in orc.FIELDS}
    fp["lens"] = {name: len(fb[name]) for name in orc.FIELDS}
    fp["dims"] = dict(zip("mknt", dims))
    fp["accepted"] = bool(ok)
    fp["u_root_hex"] = fb["u_root"].hex()
    print(f"2^{log_n}: statement {t1 - t0:.1f} s, prove {t2 - t1:.1f} s, verify {time.time() - t2:.1f} s, accepted {ok}, dims {dims}", flush=True)
    return fp


def main():
    logs = [int(a) for a in sys.argv[1:]] or [10, 14, 20]
    path = os.path.join(HERE, "proofs_large.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out["provenance"] = ("MODEL-DERIVED by tests/golden/make_golden_proofs_large.py: oracle/model_prover.py (LigeroCircuit::new) + oracle/ligero_oracle.c "
                         "orc_prove on tools/gen_repeated_squaring_r1cs.py's circuit, seed 1; not produced by the Rust crate (PARITY UNPINNED)")
    for log_n in logs:
        out[f"s{log_n}"] = fingerprint(log_n, threads=min(8, os.cpu_count() or 1))
        assert out[f"s{log_n}"]["accepted"]
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
