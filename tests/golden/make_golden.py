#!/usr/bin/env python3
"""Regenerates the fixtures under tests/golden/.  Run in the BUILD container only (it reads
/root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

What it writes
  cube.r1cs, multiplication.r1cs, poseidon.r1cs, poseidon_witness.json, poseidon_witness.wtns
      DATA files copied from the reference's own test fixtures (circom/cube.r1cs,
      circom/poseidon/poseidon.r1cs, circom/poseidon/witness.json; used by
      src/arithmetic_circuit/tests.rs:189-241 and src/ligero/tests.rs:364-415).
  poseidon_witness_batch64.bin
      64 witnesses (265 x 32-byte LE values each) for public inputs [10+i, 1+i, 42+i],
      produced with the reference's circom witness calculator
      (node circom/poseidon/poseidon_js/generate_witness.js); i = 0 reproduces witness.json.
  vectors.json
      Known answers.  There are no golden bytes in the reference for this path and the Rust
      crate cannot be built here, so these are MODEL-DERIVED (oracle/model.py, Python big
      ints + hashlib), labelled as such, plus RFC 7693 / FIPS 180-4 hash vectors.
"""
import hashlib
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import model as M  # noqa: E402

REF = "/root/reference"


def read_wtns(path):
    d = open(path, "rb").read()
    assert d[:4] == b"wtns"
    off = 12
    secs = {}
    for _ in range(struct.unpack_from("<I", d, 8)[0]):
        typ, ln = struct.unpack_from("<IQ", d, off)
        off += 12
        secs[typ] = (off, ln)
        off += ln
    o, _ = secs[1]
    fs = struct.unpack_from("<I", d, o)[0]
    nw = struct.unpack_from("<I", d, o + 4 + fs)[0]
    o, _ = secs[2]
    return [int.from_bytes(d[o + i * fs:o + (i + 1) * fs], "little") for i in range(nw)]


def main():
    shutil.copy(f"{REF}/circom/cube.r1cs", f"{HERE}/cube.r1cs")
    shutil.copy(f"{REF}/circom/poseidon/poseidon.r1cs", f"{HERE}/poseidon.r1cs")
    shutil.copy(f"{REF}/circom/poseidon/witness.json", f"{HERE}/poseidon_witness.json")
    shutil.copy(f"{REF}/circom/poseidon/witness.wtns", f"{HERE}/poseidon_witness.wtns")   # the same witness, snarkjs binary format
    shutil.copy(f"{REF}/circom/multiplication.r1cs", f"{HERE}/multiplication.r1cs")

    # ---- batch of 64 Poseidon witnesses via the reference's wasm witness calculator
    blob = bytearray()
    with tempfile.TemporaryDirectory() as td:
        for i in range(64):
            json.dump({"inputs": [10 + i, 1 + i, 42 + i]}, open(f"{td}/in.json", "w"))
            subprocess.check_call(["node", f"{REF}/circom/poseidon/poseidon_js/generate_witness.js",
                                   f"{REF}/circom/poseidon/poseidon_js/poseidon.wasm", f"{td}/in.json", f"{td}/out.wtns"])
            w = read_wtns(f"{td}/out.wtns")
            assert len(w) == 265
            if i == 0:
                assert w == M.load_witness_json(f"{HERE}/poseidon_witness.json")
            for v in w:
                blob += v.to_bytes(32, "little")
    open(f"{HERE}/poseidon_witness_batch64.bin", "wb").write(bytes(blob))

    vec = {"provenance": "model-derived (oracle/model.py: Python big ints + hashlib), NOT arkworks-derived; "
                         "hash vectors from RFC 7693 appendix B and FIPS 180-4"}
    # ---- constants
    vec["field"] = {"modulus": str(M.P), "R": str(M.R), "R2": str(M.R2), "inv64": hex(M.INV64),
                    "two_adic_root": str(M.TWO_ADIC_ROOT),
                    "omega": {str(s): str(M.domain_generator(s)) for s in (4, 8, 32, 128, 1024, 4096, 32768, 65536)}}
    # ---- hash KATs
    vec["blake2s_abc"] = "508c5e8c327c14e2e1a72ba34eeb452f37458b209ed63a294d999b4c86675982"      # RFC 7693 app. B
    vec["sha256_abc"] = "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"       # FIPS 180-4
    assert hashlib.blake2s(b"abc").hexdigest() == vec["blake2s_abc"]
    assert hashlib.sha256(b"abc").hexdigest() == vec["sha256_abc"]
    # ---- RS known answers
    u = M.reed_solomon([1, 2, 3, 4], 4, 32)
    vec["rs_k4"] = {"msg": [1, 2, 3, 4], "codeword": [str(x) for x in u],
                    "coeffs": [str(x) for x in M.reed_solomon_interpolate([1, 2, 3, 4], 4)]}
    assert [u[8 * q] for q in range(4)] == [1, 2, 3, 4]
    vec["col_hash_1_2_3"] = M.col_hash([1, 2, 3]).hex()
    lv = [M.col_hash([j]) for j in range(4)]
    vec["merkle_4"] = {"leaves": [x.hex() for x in lv], "root": M.merkle_tree(lv)[0].hex()}
    vec["calculate_t"] = {str(k): M.reed_solomon_parameters(k, k, 128)[1] for k in (2, 4, 8, 16, 32, 64, 128, 4096, 8192)}
    # ---- cube (config 1): worked preenc_u of SURVEY appendix A7
    m, k, n, t, pre, circ, outs = M.preenc_from_r1cs(f"{HERE}/cube.r1cs", [1, 3, 9])
    co, uu, leaves, nodes, root = M.encode_commit(pre, k, n)
    vec["cube"] = {"m": m, "k": k, "n": n, "t": t, "nodes": len(circ.nodes), "outputs": outs,
                   "preenc_u": [[str(v) for v in r] for r in pre],
                   "coeffs": [[str(v) for v in r] for r in co],
                   "u_row0": [str(v) for v in uu[0]], "u_row15": [str(v) for v in uu[15]],
                   "leaves": [x.hex() for x in leaves], "nodes_heap": [x.hex() for x in nodes], "root": root.hex()}
    # ---- Poseidon (config 2)
    w = M.load_witness_json(f"{HERE}/poseidon_witness.json")
    m, k, n, t, pre, circ, outs = M.preenc_from_r1cs(f"{HERE}/poseidon.r1cs", w)
    co, uu, leaves, nodes, root = M.encode_commit(pre, k, n)
    idx = [0, 1, 7, 8, 511, 512, 1022, 1023]
    cols, paths = M.open_columns(uu, leaves, nodes, idx)
    vec["poseidon"] = {"m": m, "k": k, "n": n, "t": t, "nodes": len(circ.nodes), "constants": len(circ.constants),
                       "num_outputs": len(outs), "root": root.hex(),
                       "leaves_sha256": hashlib.sha256(b"".join(leaves)).hexdigest(),
                       "nodes_sha256": hashlib.sha256(b"".join(nodes)).hexdigest(),
                       "coeffs_sha256": hashlib.sha256(b"".join(M.fr_to_bytes(v) for r in co for v in r)).hexdigest(),
                       "u_sha256": hashlib.sha256(b"".join(M.fr_to_bytes(v) for r in uu for v in r)).hexdigest(),
                       "leaf_0": leaves[0].hex(), "leaf_1023": leaves[1023].hex(),
                       "open_idx": idx,
                       "open_cols_sha256": hashlib.sha256(b"".join(M.fr_to_bytes(v) for c in cols for v in c)).hexdigest(),
                       "open_sib": [p[0].hex() for p in paths],
                       "open_paths": [[x.hex() for x in p[1]] for p in paths]}
    # ---- batch-64 roots
    roots = []
    for i in range(64):
        wi = [int.from_bytes(blob[(i * 265 + j) * 32:(i * 265 + j + 1) * 32], "little") for j in range(265)]
        _, _, _, _, pre_i, _, _ = M.preenc_from_r1cs(f"{HERE}/poseidon.r1cs", wi)
        roots.append(M.encode_commit(pre_i, k, n)[4].hex())
    assert roots[0] == root.hex()
    vec["poseidon_batch64_roots"] = roots
    json.dump(vec, open(f"{HERE}/vectors.json", "w"), indent=1)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
