"""Pin against the Rust reference.  tests/golden/rust_dump.json is produced by rust-shim/run.sh on a box with cargo (this
image has none): the reference's own prove / verify on multiplication.r1cs and the Poseidon fixture, with recording hash and
sponge wrappers.  Absent: skipped, and parity stays "model-derived".  Present: every recorded quantity must equal this
repository's (tests/golden/compare_rust_dump.py) -- no tolerance."""
import os

import pytest

from conftest import GOLDEN

DUMP = os.path.join(GOLDEN, "rust_dump.json")


def test_expected_side_is_self_consistent():
    """the comparison's own side (C++ host pipeline + oracle + transcript) builds for both cases; the commitment roots are the
    committed goldens -- so that the script is known to run before a maintainer ever has a dump"""
    import json
    from golden.compare_rust_dump import expected_case
    vec = json.load(open(os.path.join(GOLDEN, "vectors.json")))
    e = expected_case("poseidon")
    assert e["dims"] == (86, 128, 1024, 156) and e["root"] == vec["poseidon"]["root"]
    assert len(e["events"]) == 10 and [len(x) for x in e["opened"]] == [156, 156, 156]
    assert len(e["two_to_one"]) == 1023 and len(e["commit_cols"]) == 1024
    from golden.compare_rust_dump import expected_bls12_377_case
    b = expected_bls12_377_case()                                          # the reference's second field, through the generic model
    assert b["dims"] == (4, 4, 32, 32) and len(b["commit_cols"]) == 32 and len(b["two_to_one"]) == 31
    assert all(len(bytes.fromhex(c[2][0])) == 48 for c in b["commit_cols"])
    s = expected_case("multiplication")
    assert s["dims"][1] == 4 and s["dims"][2] == 32 and s["dims"][3] == 32          # t = n: every column opened
    assert s["opened"] == [list(range(32))] * 3


@pytest.mark.skipif(not os.path.exists(DUMP), reason="no Rust dump (run rust-shim/run.sh on a box with cargo)")
def test_rust_dump_matches():
    from golden.compare_rust_dump import compare
    for line in compare(DUMP):
        print(line)


def _self_dump(tmp_path, mutate=None):
    """a dump in the Rust shim's format made from this repository's own expectations (plumbing check of compare())"""
    import json
    from golden.compare_rust_dump import expected_case
    cases = []
    for name in ("multiplication", "poseidon"):
        e = expected_case(name)
        n = e["dims"][2]
        log = {"col_hash_input_sha256": [c[0] for c in e["commit_cols"]], "col_hash_input_len": [c[1] for c in e["commit_cols"]],
               "col_hash_first_elems": [c[2] for c in e["commit_cols"]], "col_hash_output": [c[3] for c in e["commit_cols"]],
               "two_to_one": [list(x) for x in sorted(e["two_to_one"])],
               "sponge": [{"op": op, "bytes": "" if op == "absorb" else v, "field_elements": v if op == "absorb" else []} for op, v in e["events"]]}
        vlog = {"col_hash_output": [e["commit_cols"][j][3] for ind in e["opened"] for j in ind]}
        cases.append({"name": name, "witness": e["witness"], "num_nodes": e["num_nodes"], "prove": log, "verify": vlog, "verified": True})
    from golden.compare_rust_dump import expected_bls12_377_case
    b = expected_bls12_377_case()
    cases.append({"name": "bls12_377_curve", "witness": b["witness"], "num_nodes": b["num_nodes"], "verified": True, "verify": {},
                  "prove": {"col_hash_input_sha256": [c[0] for c in b["commit_cols"]], "col_hash_input_len": [c[1] for c in b["commit_cols"]],
                            "col_hash_first_elems": [c[2] for c in b["commit_cols"]], "col_hash_output": [c[3] for c in b["commit_cols"]],
                            "two_to_one": [list(x) for x in sorted(b["two_to_one"])], "sponge": []}})
    if mutate:
        mutate(cases)
    p = tmp_path / "dump.json"
    p.write_text(json.dumps(cases))
    return str(p)


def test_compare_accepts_a_faithful_dump_and_names_what_differs(tmp_path):
    from golden.compare_rust_dump import compare
    assert len(compare(_self_dump(tmp_path))) == 3

    def other_generator(cases):           # as if Fq's domain came from another primitive root
        cases[2]["prove"]["col_hash_first_elems"][1][0] = "22" * 48
    with pytest.raises(AssertionError, match="multiplicative generator"):
        compare(_self_dump(tmp_path, other_generator))

    def no_length_prefix(cases):          # as if serialize_compressed had no u64 length prefix
        cases[1]["prove"]["col_hash_input_sha256"][5] = "00" * 32
    with pytest.raises(AssertionError, match="length prefix"):
        compare(_self_dump(tmp_path, no_length_prefix))

    def other_seed(cases):                # as if squeeze_bytes packed differently
        cases[1]["prove"]["sponge"][1]["bytes"] = "11" * 32
    with pytest.raises(AssertionError, match="squeeze_bytes #1"):
        compare(_self_dump(tmp_path, other_seed))

    def raw_leaf_level(cases):            # as if the bottom level compressed raw digests (no LE64(32) prefixes)
        cases[0]["prove"]["two_to_one"] = [x for x in cases[0]["prove"]["two_to_one"] if x[0] != "evaluate"]
    with pytest.raises(AssertionError, match="ByteDigestConverter"):
        compare(_self_dump(tmp_path, raw_leaf_level))
