"""CPU tests: pin the oracle (C restatement + Python model) against every known answer we
have -- RFC/FIPS hash vectors, hashlib, the committed model-derived fixtures, and the
reference's own acceptance facts (cube compiles to 15 nodes; the Poseidon witness makes every
output evaluate to 1)."""
import hashlib
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, mont_matrix, random_mont


def test_field_constants(vectors, model):
    f = vectors["field"]
    assert int(f["modulus"]) == model.P
    assert int(f["R"]) == (1 << 256) % model.P
    assert int(f["two_adic_root"]) == pow(5, (model.P - 1) >> 28, model.P)
    for size, w in f["omega"].items():
        w = int(w)
        assert pow(w, int(size), model.P) == 1 and pow(w, int(size) // 2, model.P) == model.P - 1
        assert model.domain_generator(int(size)) == w
    # small_domain = large_domain^(n/k): omega_k = omega_n^8 (src/ligero/mod.rs:89)
    assert pow(int(f["omega"]["1024"]), 8, model.P) == int(f["omega"]["128"])


def test_modulus_matches_reference_fixture(model):
    prime, n_wires, cons = model.read_r1cs(os.path.join(GOLDEN, "poseidon.r1cs"))
    assert prime == model.P and n_wires == 265 and len(cons) == 261


def test_oracle_field_ops(oracle, model):
    vals = model.random_elements(7, 200) + [0, 1, model.P - 1, model.P - 2, 2]
    a = oracle.ints_to_limbs(vals)
    am = oracle.to_mont(a)
    assert oracle.limbs_to_ints(am) == [model.to_mont(v) for v in vals]
    assert oracle.limbs_to_ints(oracle.from_mont(am)) == vals
    L = oracle.lib()
    out = np.zeros(4, dtype=np.uint64)
    for x, y in zip(vals[:-1], vals[1:]):
        xm = oracle.ints_to_limbs([model.to_mont(x)])
        ym = oracle.ints_to_limbs([model.to_mont(y)])
        L.orc_fr_mul(xm.ctypes.data, ym.ctypes.data, out.ctypes.data)
        assert oracle.limbs_to_ints(out)[0] == model.to_mont(x * y % model.P)
        L.orc_fr_add(xm.ctypes.data, ym.ctypes.data, out.ctypes.data)
        assert oracle.limbs_to_ints(out)[0] == model.to_mont((x + y) % model.P)
        L.orc_fr_sub(xm.ctypes.data, ym.ctypes.data, out.ctypes.data)
        assert oracle.limbs_to_ints(out)[0] == model.to_mont((x - y) % model.P)


def test_hash_kats(oracle, vectors):
    assert oracle.blake2s256(b"abc").hex() == vectors["blake2s_abc"]
    assert oracle.sha256(b"abc").hex() == vectors["sha256_abc"]
    assert oracle.sha256(b"").hex() == "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"
    assert oracle.blake2s256(b"").hex() == "69217a3079908094e11121d042354a7c1f55b6482ca1a51e1b250dfd1ed0eef9"
    for n in [1, 8, 40, 55, 56, 63, 64, 65, 72, 119, 120, 127, 128, 129, 1000, 11016]:
        d = bytes((i * 131 + 7) & 255 for i in range(n))
        assert oracle.sha256(d) == hashlib.sha256(d).digest(), n
        assert oracle.blake2s256(d) == hashlib.blake2s(d).digest(), n


@pytest.mark.parametrize("k", [2, 4, 8, 16, 64, 128, 512])
def test_oracle_fft_vs_model(oracle, model, k):
    v = model.random_elements(100 + k, k)
    x = oracle.to_mont(oracle.ints_to_limbs(v))
    assert oracle.limbs_to_ints(oracle.from_mont(oracle.fft(x))) == model.ntt(v, model.domain_generator(k))
    assert oracle.limbs_to_ints(oracle.from_mont(oracle.ifft(x))) == model.intt(v, model.domain_generator(k))
    if k <= 16:
        assert model.ntt(v, model.domain_generator(k)) == model.naive_dft(v, model.domain_generator(k))


def test_rs_known_answer(oracle, model, vectors):
    g = vectors["rs_k4"]
    msg = oracle.to_mont(oracle.ints_to_limbs(g["msg"]))
    co = oracle.reed_solomon_interpolate(msg, 4)
    assert [str(x) for x in oracle.limbs_to_ints(oracle.from_mont(co))] == g["coeffs"]
    cw = oracle.reed_solomon_evaluate(co, 32)
    got = oracle.limbs_to_ints(oracle.from_mont(cw))
    assert [str(x) for x in got] == g["codeword"]
    assert [got[8 * q] for q in range(4)] == g["msg"]          # systematic
    # shorter message is zero-padded (mod.rs:1000)
    short = oracle.reed_solomon_interpolate(msg[:3], 4)
    assert oracle.limbs_to_ints(oracle.from_mont(short)) == model.reed_solomon_interpolate([1, 2, 3], 4)


def test_col_hash_and_merkle_known_answers(oracle, model, vectors):
    col = oracle.to_mont(oracle.ints_to_limbs([1, 2, 3]))
    assert oracle.col_hash(col).hex() == vectors["col_hash_1_2_3"]
    expected = hashlib.blake2s(struct.pack("<Q", 3) + b"".join(i.to_bytes(32, "little") for i in (1, 2, 3))).hexdigest()
    assert vectors["col_hash_1_2_3"] == expected
    leaves = np.frombuffer(bytes.fromhex("".join(vectors["merkle_4"]["leaves"])), dtype=np.uint8)
    nodes = oracle.merkle_tree(leaves)
    assert nodes[0].tobytes().hex() == vectors["merkle_4"]["root"]
    # ragged column lengths: odd / even / single element
    for ln in (1, 2, 5, 16, 345):
        vals = model.random_elements(ln, ln)
        assert oracle.col_hash(oracle.to_mont(oracle.ints_to_limbs(vals))) == model.col_hash(vals)


def test_dimensions_and_t(model, vectors):
    for k, t in vectors["calculate_t"].items():
        k = int(k)
        assert model.reed_solomon_parameters(k, k, 128) == (8 * k, t)
    assert model.compute_dimensions(7274) == (86, 128)
    assert model.compute_dimensions(15) == (4, 4)
    assert model.compute_dimensions(6291458) == (2509, 4096)
    assert model.compute_dimensions(25165826) == (5017, 8192)


def test_cube_fixture(oracle, model, vectors, cube_case):
    g = vectors["cube"]
    c = cube_case
    assert (c["m"], c["k"], c["n"], c["t"]) == (4, 4, 32, 32)
    assert len(c["circ"].nodes) == 15 == g["nodes"]            # src/arithmetic_circuit/tests.rs:239
    assert [[str(v) for v in r] for r in c["preenc"]] == g["preenc_u"]
    # worked example of SURVEY appendix A7 (W block)
    sgn = lambda v: v if v < model.P // 2 else v - model.P
    assert [[sgn(v) for v in r] for r in c["preenc"][12:16]] == [[1, 3, 9, -3], [-9, -9, 27, 9], [-27, 0, 1, 0], [1, 0, 0, 0]]
    r = oracle.encode_commit(mont_matrix(oracle, c["preenc"], 4), 4, 32)
    assert r["root"].hex() == g["root"]
    assert [x.tobytes().hex() for x in r["leaves"]] == g["leaves"]
    assert [x.tobytes().hex() for x in r["nodes"]] == g["nodes_heap"]
    co = oracle.limbs_to_ints(oracle.from_mont(r["coeffs"]))
    assert [str(v) for v in co] == [v for row in g["coeffs"] for v in row]
    u = oracle.limbs_to_ints(oracle.from_mont(r["u"]))
    assert [str(v) for v in u[:32]] == g["u_row0"] and [str(v) for v in u[15 * 32:]] == g["u_row15"]


def test_poseidon_fixture(oracle, model, vectors, poseidon_case):
    g = vectors["poseidon"]
    c = poseidon_case
    assert (c["m"], c["k"], c["n"], c["t"]) == (86, 128, 1024, 156) == (g["m"], g["k"], g["n"], g["t"])
    assert len(c["circ"].nodes) == g["nodes"] == 7787 and len(c["circ"].constants) == g["constants"] == 775
    pre = mont_matrix(oracle, c["preenc"], 128)
    r = oracle.encode_commit(pre, 128, 1024)
    assert r["root"].hex() == g["root"]
    assert hashlib.sha256(r["leaves"].tobytes()).hexdigest() == g["leaves_sha256"]
    assert hashlib.sha256(r["nodes"].tobytes()).hexdigest() == g["nodes_sha256"]
    assert hashlib.sha256(oracle.from_mont(r["coeffs"]).tobytes()).hexdigest() == g["coeffs_sha256"]
    assert hashlib.sha256(oracle.from_mont(r["u"]).tobytes()).hexdigest() == g["u_sha256"]
    assert r["leaves"][0].tobytes().hex() == g["leaf_0"] and r["leaves"][1023].tobytes().hex() == g["leaf_1023"]
    # all-cores variant computes the same thing
    assert oracle.encode_commit(pre, 128, 1024, threads=4)["root"].hex() == g["root"]
    # openings (mod.rs:944-952) + Path::verify
    cols, sib, paths = oracle.open_columns(r["u"], r["leaves"], r["nodes"], g["open_idx"])
    assert hashlib.sha256(oracle.from_mont(cols).tobytes()).hexdigest() == g["open_cols_sha256"]
    assert [s.tobytes().hex() for s in sib] == g["open_sib"]
    assert [[x.tobytes().hex() for x in p] for p in paths] == g["open_paths"]
    root = bytes.fromhex(g["root"])
    for i, j in enumerate(g["open_idx"]):
        leaf = oracle.col_hash(cols[i])
        assert leaf == r["leaves"][j].tobytes()
        assert model.merkle_verify(root, leaf, j, sib[i].tobytes(), [x.tobytes() for x in paths[i]])
        assert not model.merkle_verify(root, leaf, j ^ 2, sib[i].tobytes(), [x.tobytes() for x in paths[i]])


def test_oracle_properties_random(oracle, model):
    """size-independent properties the GPU tests rely on at full size, checked here on the oracle"""
    k, n, rows = 64, 512, 6
    pre = random_mont(3, rows * k).reshape(rows, k, 4)
    r = oracle.encode_commit(pre, k, n)
    u = r["u"]
    assert np.array_equal(u[:, ::8, :], pre)                                    # systematic: U[i][8q] = msg[i][q]
    L = oracle.lib()
    s = np.zeros((k, 4), dtype=np.uint64)
    for q in range(k):                                                          # linearity: enc(a+b) = enc(a)+enc(b)
        L.orc_fr_add(pre[0, q].ctypes.data, pre[1, q].ctypes.data, s[q].ctypes.data)
    es = oracle.reed_solomon_evaluate(oracle.reed_solomon_interpolate(s, k), n)
    t = np.zeros((n, 4), dtype=np.uint64)
    for j in range(n):
        L.orc_fr_add(u[0, j].ctypes.data, u[1, j].ctypes.data, t[j].ctypes.data)
    assert np.array_equal(es, t)


def test_subproof_polynomials_model_vs_c(oracle, model):
    """SURVEY 8f #1-2 arithmetic: C restatement (FFT products) vs big-int model (schoolbook)"""
    m, k = 3, 8
    rows = 4 * m
    pre = model.random_elements(61, rows * k)
    pre_rows = [pre[i * k:(i + 1) * k] for i in range(rows)]
    coeffs = [model.reed_solomon_interpolate(r, k) for r in pre_rows]
    r_int = model.random_elements(62, rows)
    r_a = model.random_elements(63, rows * k)
    r_a_rows = [r_a[i * k:(i + 1) * k] for i in range(rows)]
    r_q = model.random_elements(64, m)
    to_m = lambda vals: oracle.to_mont(oracle.ints_to_limbs(vals))
    from_m = lambda a: oracle.limbs_to_ints(oracle.from_mont(a))
    lc = oracle.dense_row_mul(to_m(pre).reshape(rows, k, 4), to_m(r_int))
    assert from_m(lc) == model.dense_row_mul(pre_rows, r_int)
    cm = to_m([v for r in coeffs for v in r]).reshape(rows, k, 4)
    lin = oracle.linear_constraint_poly(cm, to_m(r_a).reshape(rows, k, 4))
    assert from_m(lin) == model.linear_constraint_poly(coeffs, r_a_rows, k)
    quad = oracle.quadratic_constraint_poly(cm, to_m(r_q))
    assert from_m(quad) == model.quadratic_constraint_poly(coeffs, r_q, m, k)
    # degree bound the verifier checks (mod.rs:782, 886): degree < 2k - 1
    assert from_m(lin)[2 * k - 1] == 0 and from_m(quad)[2 * k - 1] == 0
    # known answer of the reference's own unit test (src/matrices/mod.rs:181-194)
    P = model.P
    mat = to_m([1, 2, 8, 3, 4, 5]).reshape(2, 3, 4)
    assert from_m(oracle.dense_row_mul(mat, to_m([P - 5, 17]))) == [46, 58, 45]


@pytest.mark.parametrize("rows,k,block", [(5, 8, 2), (13, 64, 4), (70, 16, 64)])
def test_streamed_commit_equals_in_memory_commit(oracle, rows, k, block):
    """the blocked-row variant that produces the full-size goldens (tests/golden/large_roots.json) hashes exactly the
    same byte string per column as the reference-shaped commit (mod.rs:521-551)"""
    pre = random_mont(rows * 7 + k, rows * k).reshape(rows, k, 4)
    a = oracle.encode_commit(pre, k, 8 * k)
    for threads in (1, 3):
        b = oracle.encode_commit_streamed(pre, k, 8 * k, threads=threads, block_rows=block)
        assert a["root"] == b["root"] and np.array_equal(a["leaves"], b["leaves"]) and np.array_equal(a["nodes"], b["nodes"])


def test_large_goldens_are_committed():
    import json
    from bench import LARGE_SEED, WORKLOADS
    g = json.load(open(os.path.join(GOLDEN, "large_roots.json")))
    for name in ("s20", "s22"):
        assert (g[name]["rows"], g[name]["k"]) == WORKLOADS[name][:2] and g[name]["seed"] == LARGE_SEED
        assert len(bytes.fromhex(g[name]["root"])) == 32


def test_generic_field_constants():
    """the constants the generic-field path embeds (ligero_amd/csrc/generic_path.hip make_field) recomputed from the moduli and
    multiplicative generators alone; BN254's must equal what model.py derives from the modulus read out of the .r1cs fixtures"""
    import re
    from oracle import model_field as mf
    src = open(os.path.join(os.path.dirname(GOLDEN), "..", "ligero_amd", "csrc", "generic_path.hip")).read()
    consts = re.findall(r'(?:p|root) = "([0-9a-f]+)";', src)
    assert len(consts) == 4
    fq, fr = mf.BLS12_377_FQ, mf.BN254_FR
    assert int(consts[0], 16) == fq.p and int(consts[1], 16) == fq.root
    assert int(consts[2], 16) == fr.p == model_P() and int(consts[3], 16) == fr.root
    assert fq.p.bit_length() == 377 and fq.two_adicity == 46 and fq.nbytes == 48
    # the curve equation the reference's BLS12-377 test proves (tests.rs:186-193 / arithmetic_circuit/tests.rs: y^2 = x^3 + 1 over Fq)
    # holds for the G1 generator's affine coordinates, which ties this modulus to that test's field
    gx = 0x008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9ef
    gy = 0x01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6
    assert (gy * gy - gx * gx * gx - 1) % fq.p == 0


def model_P():
    from oracle import model
    return model.P


def test_generic_field_model_agrees_with_bn254_model():
    from oracle import model, model_field as mf
    import random
    random.seed(5)
    f = mf.BN254_FR
    pre = [[random.randrange(f.p) for _ in range(8)] for _ in range(5)]
    assert f.encode_commit(pre, 8, 64) == model.encode_commit(pre, 8, 64)
    # 48-byte elements: framing against hashlib directly
    q = mf.BLS12_377_FQ
    col = [1, q.p - 1, 12345]
    assert q.col_hash(col) == hashlib.blake2s(struct.pack("<Q", 3) + b"".join(v.to_bytes(48, "little") for v in col)).digest()
    msg = [random.randrange(q.p) for _ in range(4)]
    cw = q.reed_solomon_evaluate(q.reed_solomon_interpolate(msg, 4), 32)
    assert cw[::8] == msg
    w = q.domain_generator(32)
    co = q.reed_solomon_interpolate(msg, 4)
    assert cw == [sum(c * pow(w, j * d, q.p) for d, c in enumerate(co)) % q.p for j in range(32)]
