"""CPU tests of the C++ host-side input pipeline (ligero_amd/host/circuit.hpp through
include/ligero_host.h): the reference's own assertions about this code, restated --

  * src/arithmetic_circuit/tests.rs:189-241  cube.r1cs compiles to exactly 15 nodes
  * src/ligero/tests.rs:35-142               exact constraint matrix A of the curve-equation circuit
  * src/ligero/tests.rs:245-346              exact A of the three-output circuit (exercises insert_one)
  * src/ligero/tests.rs:364-415              Poseidon: every output evaluates to 1; dimensions
and the pins we can add: preenc_u equals the oracle model's on cube-free fixtures, A.row_mul
equals a big-int model of SparseMatrix::row_mul.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, mont_matrix, random_mont

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


@pytest.fixture(scope="module")
def hp():
    from ligero_amd import host_pipeline
    host_pipeline.lib()
    return host_pipeline


def _mont(oracle, v):
    return oracle.to_mont(oracle.ints_to_limbs([v % P]))[0]


def _a_as_dict(oracle, inst):
    """A as {row: [(value, column), ...]} with the entry order of the rows preserved (the
    reference compares SparseMatrix row vectors, duplicates included)"""
    rows, cols, vals = inst.a_entries()
    ints = oracle.limbs_to_ints(oracle.from_mont(vals))
    out = {}
    for r, c, v in zip(rows, cols, ints):
        out.setdefault(int(r), []).append((v if v < P // 2 else v - P, int(c)))
    return out


def _expected_a(mk, p_x, p_y, p_z, p_add):
    """A = [[I | -P_x; -P_y; -P_z], [0 | P_add]] (src/ligero/mod.rs:423-432) from the tables the
    reference's tests spell out (row index -> [(value, column)])"""
    a = {i: [(1, i)] for i in range(3 * mk)}
    for blk, tab in enumerate((p_x, p_y, p_z)):
        for r, entries in tab.items():
            a[blk * mk + r] += [(-v, 3 * mk + c) for v, c in entries]
    for r, entries in p_add.items():
        a[3 * mk + r] = [(v, 3 * mk + c) for v, c in entries]
    return a


def test_cube_compiles_to_15_nodes_and_new_panics(hp):
    c = hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "cube.r1cs"))
    assert c.num_nodes() == 15 and c.outputs == [12, 14]                 # arithmetic_circuit/tests.rs:239
    # the reference cannot build LigeroCircuit for cube (Mul of two constants, mod.rs:345)
    with pytest.raises(hp.HostPanic):
        hp.LigeroInstance(c)


def test_construction_curve_equation_matrix(hp, oracle):
    """src/ligero/tests.rs:35-142 (there over BLS12-377 Fq; the matrix has only +-1 entries, so the
    same table must come out over BN254 Fr)"""
    c = hp.ArithmeticCircuit()
    one = c.constant(_mont(oracle, 1))
    x, y = c.new_variable(), c.new_variable()
    y2 = c.pow(y, 2)
    my2 = c.minus(y2)
    x3 = c.pow(x, 3)
    out = c.add(c.add(c.add(x3, one), my2), one)                         # add_nodes([x_cubed, one, minus_y_squared, one])
    inst = hp.LigeroInstance(c, [out])
    assert (inst.m, inst.k) == (4, 4)
    p_x = {3: [(1, 2)], 4: [(-1, 0)], 5: [(1, 1)], 6: [(1, 5)]}
    p_y = {3: [(1, 2)], 4: [(1, 3)], 5: [(1, 1)], 6: [(1, 1)]}
    p_z = {3: [(1, 3)], 4: [(1, 4)], 5: [(1, 5)], 6: [(1, 6)]}
    p_add = {7: [(1, 6), (1, 0), (-1, 7)], 8: [(1, 7), (1, 4), (-1, 8)], 9: [(1, 8), (1, 0), (-1, 9)], 10: [(1, 8), (1, 0), (-1, 0)]}
    assert _a_as_dict(oracle, inst) == _expected_a(16, p_x, p_y, p_z, p_add)


def test_multioutput_1_matrix(hp, oracle):
    """src/ligero/tests.rs:245-346: no constant 1 in the circuit -> insert_one shifts every index"""
    c = hp.ArithmeticCircuit()
    x, y = c.new_variable(), c.new_variable()
    c1, c2, c3 = c.constant(_mont(oracle, -8)), c.constant(_mont(oracle, -63)), c.constant(_mont(oracle, -6))
    x2 = c.mul(x, x)
    y3 = c.pow(y, 3)
    s = c.add(x, y)
    o1, o2, o3 = c.add(x2, c1), c.add(y3, c2), c.add(s, c3)
    inst = hp.LigeroInstance(c, [o1, o2, o3])
    assert (inst.m, inst.k, inst.m * inst.k) == (4, 4, 16)
    p_x = {3: [(1, 1)], 4: [(1, 2)], 5: [(1, 4)]}
    p_y = {3: [(1, 1)], 4: [(1, 2)], 5: [(1, 2)]}
    p_z = {3: [(1, 3)], 4: [(1, 4)], 5: [(1, 5)]}
    p_add = {6: [(1, 1), (1, 2), (-1, 6)], 7: [(1, 3), (-8, 0), (-1, 7)], 8: [(1, 5), (-63, 0), (-1, 8)], 9: [(1, 6), (-6, 0), (-1, 9)],
             10: [(1, 3), (-8, 0), (-1, 0)], 11: [(1, 5), (-63, 0), (-1, 0)], 12: [(1, 6), (-6, 0), (-1, 0)]}
    assert _a_as_dict(oracle, inst) == _expected_a(16, p_x, p_y, p_z, p_add)
    # a satisfying assignment (x = 3, y = 4) makes all three outputs 1; a wrong one does not
    pre, ok = inst.build_preenc_u([x, y], np.stack([_mont(oracle, 3), _mont(oracle, 4)]))
    assert ok and pre.shape == (16, 4, 4)
    _, bad = inst.build_preenc_u([x, y], np.stack([_mont(oracle, 4), _mont(oracle, 4)]))
    assert not bad


def test_poseidon_from_fixture_matches_model(hp, oracle, model, poseidon_case):
    c = hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "poseidon.r1cs"))
    inst = hp.LigeroInstance(c)
    assert (inst.m, inst.k, inst.n, inst.t) == (86, 128, 1024, 156)
    assert (inst.num_nodes, inst.num_constants, inst.num_outputs) == (7787, 775, 261)
    assert c.outputs == poseidon_case["outputs"]
    w = model.load_witness_json(os.path.join(GOLDEN, "poseidon_witness.json"))
    vals = oracle.to_mont(oracle.ints_to_limbs(w[1:]))
    pre, ok = inst.build_preenc_u(list(range(1, len(w))), vals)          # tests.rs:389: enumerate().skip(1)
    assert ok                                                            # tests.rs:391-394
    assert np.array_equal(pre, mont_matrix(oracle, poseidon_case["preenc"], 128))
    bad = vals.copy()
    bad[0] = _mont(oracle, w[1] + 1)
    assert not inst.build_preenc_u(list(range(1, len(w))), bad)[1]       # tests.rs:160-170 style rejection
    with pytest.raises(hp.HostPanic):                                    # missing variable: mod.rs:477 panic
        inst.build_preenc_u(list(range(2, len(w))), vals[1:])


def test_a_row_mul_matches_model(hp, oracle, model):
    c = hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "poseidon.r1cs"))
    inst = hp.LigeroInstance(c)
    r_int = model.random_elements(5, 4 * inst.m * inst.k)
    got = oracle.limbs_to_ints(oracle.from_mont(inst.a_row_mul(oracle.to_mont(oracle.ints_to_limbs(r_int)))))
    rows, cols, vals = inst.a_entries()
    ints = oracle.limbs_to_ints(oracle.from_mont(vals))
    exp = [0] * (4 * inst.m * inst.k)
    for r, cidx, v in zip(rows, cols, ints):                             # src/matrices/mod.rs:100-110
        exp[int(cidx)] = (exp[int(cidx)] + r_int[int(r)] * v) % P
    assert got == exp
    # reference's own known answer for SparseMatrix::row_mul is covered structurally: identity block
    mk = inst.m * inst.k
    assert got[:3 * mk] == r_int[:3 * mk]


def test_host_library_exports(hp):
    import re, subprocess
    from conftest import ROOT
    hdr = open(os.path.join(ROOT, "include", "ligero_host.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(lgh_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(hp.SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", hp.LIB_PATH], text=True)
    exported = set(re.findall(r" T (lgh_[a-z0-9_]+)", out))
    assert set(declared) <= exported
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "ligero_host.h")])


def test_prover_library_exports():
    """include/ligero_prover.h <-> libligero_prover.so <-> the ctypes mirror (no device call: loads without a GPU)"""
    import re, subprocess
    from conftest import ROOT
    from ligero_amd import prover
    hdr = open(os.path.join(ROOT, "include", "ligero_prover.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(lgp_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(prover.SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", prover.LIB_PATH], text=True)
    assert set(declared) <= set(re.findall(r" T (lgp_[a-z0-9_]+)", out))
    L = prover.lib()
    for name in declared:
        assert getattr(L, name) is not None
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "include", "ligero_prover.h")])


def test_witness_readers(hp, oracle, model):
    """witness.json and the .wtns of the same witness (both from the reference's circom/poseidon/) give the same wires,
    equal to the test loader's; malformed files are refused"""
    import tempfile
    from conftest import GOLDEN
    want = oracle.to_mont(oracle.ints_to_limbs(model.load_witness_json(os.path.join(GOLDEN, "poseidon_witness.json"))))
    got_json = hp.read_witness(os.path.join(GOLDEN, "poseidon_witness.json"))
    got_wtns = hp.read_witness(os.path.join(GOLDEN, "poseidon_witness.wtns"))
    assert got_json.shape == (265, 4) and np.array_equal(got_json, want) and np.array_equal(got_wtns, want)
    with tempfile.TemporaryDirectory() as d:
        blob = open(os.path.join(GOLDEN, "poseidon_witness.wtns"), "rb").read()
        for name, data in (("short.wtns", blob[:100]), ("prime.wtns", blob[:31] + b"\x00" + blob[32:]), ("junk.json", b'{"a": 1}'), ("open.json", b'["1", "2"')):
            path = os.path.join(d, name)
            open(path, "wb").write(data)
            with pytest.raises(hp.HostPanic):
                hp.read_witness(path)
        path = os.path.join(d, "plain.json")
        open(path, "wb").write(b'[ "0", 7 ,"21888242871839275222246405745257275088548364400416034343698204186575808495618" ]')
        got = hp.read_witness(path)                                     # p + 1 reduces to 1
        assert np.array_equal(got, oracle.to_mont(oracle.ints_to_limbs([0, 7, 1])))


def test_r1cs_reader_refuses_corrupt_files(hp):
    """truncations, section lengths that run past the file (or wrap around), header counts the file cannot hold and
    wire indices out of range are refused with an error, never read out of bounds or turned into huge allocations
    (the same mutations pass an AddressSanitizer + UBSan build of the library, tools/fuzz_readers.py)"""
    import tempfile
    from conftest import GOLDEN
    good = open(os.path.join(GOLDEN, "poseidon.r1cs"), "rb").read()
    cases = [good[:n] for n in (0, 3, 11, 12, 23, 24, 60, 100, 5000, len(good) - 1)]
    cases.append(good[:16] + (2 ** 64 - 8).to_bytes(8, "little") + good[24:])          # first section length wraps the offset
    cases.append(good[:16] + (1 << 40).to_bytes(8, "little") + good[24:])              # ... or just runs past the end
    hdr = good.index(bytes.fromhex("20000000010000f093f5e143"))                         # field size 32 + start of the prime
    cases.append(good[:hdr + 36] + (0x7FFFFFFF).to_bytes(4, "little") + good[hdr + 40:])   # n_wires far beyond the file
    cases.append(good[:hdr + 60] + (0x7FFFFFFF).to_bytes(4, "little") + good[hdr + 64:])   # n_constraints likewise
    with tempfile.TemporaryDirectory() as d:
        for i, data in enumerate(cases):
            path = os.path.join(d, f"c{i}.r1cs")
            open(path, "wb").write(data)
            with pytest.raises(hp.HostPanic):
                hp.ArithmeticCircuit.from_r1cs(path)


def _gen_rs():
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("gen_rs", os.path.join(ROOT, "tools", "gen_repeated_squaring_r1cs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("log_n", [3, 10])
def test_repeated_squaring_generator_and_pipeline(hp, model, tmp_path, log_n):
    """the synthetic R1CS of BASELINE configs[2] / [3] at small N: file layout (parsed by the model's reader AND the C++ one),
    node count 5 N + 3, dimensions by mod.rs:171-175 / 275-294, all outputs one, preenc_u equal to the model's"""
    gen = _gen_rs()
    n = 1 << log_n
    r1cs, wtns = str(tmp_path / "rs.r1cs"), str(tmp_path / "rs.wtns")
    gen.write_r1cs(r1cs, log_n)
    wit = gen.witness(log_n, 7)
    gen.write_wtns(wtns, wit)
    assert all(wit[i + 2] == wit[i + 1] ** 2 % model.P for i in range(n))
    circ = hp.ArithmeticCircuit.from_r1cs(r1cs)
    assert circ.num_nodes() == 5 * n + 3 and len(circ.outputs) == n
    inst = hp.LigeroInstance(circ)
    sol = 1 + (5 * n + 3) - 2 + n
    m, k = model.compute_dimensions(sol)
    assert (inst.m, inst.k, inst.n) == (m, k, 8 * k)
    w = hp.read_witness(wtns)
    assert w.shape[0] == n + 2
    pre, ok = inst.build_preenc_u(np.arange(1, n + 2, dtype=np.uint64), w[1:])
    assert ok
    mm, kk, nn, tt, mpre, _, outs = model.preenc_from_r1cs(r1cs, wit)        # (asserts itself that every output evaluates to 1)
    assert (mm, kk, nn, tt) == (inst.m, inst.k, inst.n, inst.t) and len(outs) == n
    from oracle import binding as orc
    want = orc.to_mont(orc.ints_to_limbs([v for row in mpre for v in row])).reshape(4 * mm, kk, 4)
    assert np.array_equal(pre, want)


def test_repeated_squaring_dimensions_at_baseline_sizes():
    """LigeroCircuit dimensions of configs[2] / [3] from the node counts alone (mod.rs:171-175, 275-294), as SURVEY 8(d) lists them"""
    from ligero_amd.ligero import compute_dimensions, reed_solomon_parameters
    for log_n, want in ((20, (2509, 4096, 32768, 156)), (22, (5017, 8192, 65536, 156))):
        n = 1 << log_n
        sol = 1 + (5 * n + 3) - 2 + n
        m, k = compute_dimensions(sol)
        nn, t = reed_solomon_parameters(m, k, 128)
        assert (m, k, nn, t) == want


def rebuild_preenc_from_w(w_block, left, right, consts, m, k):
    """what the device's gather does (include/ligero_hip.h lg_encode_commit_from_witness), in numpy: x[p], y[p] = the operands
    of the Mul gate at position p (a position of w or a constant), z[p] = w[p], zero elsewhere (src/ligero/mod.rs:495-503)"""
    mk = m * k
    w = np.ascontiguousarray(w_block).reshape(mk, 4)
    pre = np.zeros((4, mk, 4), dtype=np.uint64)
    pre[3] = w
    gates = np.nonzero(left != 0xffffffff)[0]
    for blk, src in ((0, left), (1, right)):
        s = src[gates]
        is_const = (s & 0x80000000) != 0
        vals = np.empty((gates.shape[0], 4), dtype=np.uint64)
        vals[is_const] = consts[(s[is_const] & 0x7fffffff).astype(np.int64)]
        vals[~is_const] = w[s[~is_const].astype(np.int64)]
        pre[blk, gates] = vals
    pre[2, gates] = w[gates]
    return pre.reshape(4 * m, k, 4)


def test_gate_map_and_w_rebuild_preenc_u(hp, oracle, model):
    """a1 on the device: preenc_u = w + the circuit's wiring.  The gate map (which positions hold Mul gates, where their
    operands live) and the W block alone must reproduce build_preenc_u -- Poseidon (775 constants without a position) and an
    expression-made circuit (to_arithmetic_circuit's numbering; the device path's forward-reference case is driven with a
    hand-made map in tests/test_gpu_witness.py)"""
    c = hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "poseidon.r1cs"))
    inst = hp.LigeroInstance(c)
    w = model.load_witness_json(os.path.join(GOLDEN, "poseidon_witness.json"))
    vals = oracle.to_mont(oracle.ints_to_limbs(w[1:]))
    idx = list(range(1, len(w)))
    pre, ok = inst.build_preenc_u(idx, vals)
    wblk, ok2 = inst.build_w(idx, vals)
    assert ok and ok2 and np.array_equal(wblk, pre[3 * inst.m:])
    left, right, consts = inst.gate_map()
    assert left.shape[0] == 1 + inst.num_nodes - inst.num_constants == 7013 and consts.shape[0] == inst.num_constants - 1
    assert int((left != 0xffffffff).sum()) == 3611                               # 3611 Mul nodes
    assert np.array_equal((left == 0xffffffff), (right == 0xffffffff))
    assert np.array_equal(rebuild_preenc_from_w(wblk, left, right, consts, inst.m, inst.k), pre)
    gates = np.nonzero(left != 0xffffffff)[0]
    for s in (left[gates], right[gates]):                                        # compiled circuits refer backwards only
        pos = s[(s & 0x80000000) == 0]
        assert (pos < gates[(s & 0x80000000) == 0]).all()
    # an expression: x * (x + 3) * y
    x, y = hp.Expression.variable("x"), hp.Expression.variable("y")
    circ = ((x * (x + 3)) * y).to_arithmetic_circuit()
    inst2 = hp.LigeroInstance(circ, [circ.num_nodes() - 1])
    labels, v = ["x", "y"], np.stack([_mont(oracle, 5), _mont(oracle, 7)])
    pre2, _ = inst2.build_preenc_u_with_labels(labels, v)
    l2, r2, c2 = inst2.gate_map()
    assert np.array_equal(rebuild_preenc_from_w(pre2[3 * inst2.m:], l2, r2, c2, inst2.m, inst2.k), pre2)


def run_trace_program(oracle, model, prog, in_pos, in_vals_mont, mk):
    """what the device does with lgh_trace_program's output (include/ligero_hip.h lg_encode_commit_from_inputs), in Python ints:
    scatter the assignment, then level by level every gate from operands that earlier levels (or the scatter) wrote"""
    p = model.P
    w = [None] * mk
    consts = oracle.limbs_to_ints(oracle.from_mont(prog["constants"])) if len(prog["constants"]) else []
    for pos, v in zip(in_pos, oracle.limbs_to_ints(oracle.from_mont(np.ascontiguousarray(in_vals_mont)))):
        w[int(pos)] = v
    if prog["op"][0] == 3:
        w[0] = 1
    lo = prog["level_off"]
    for lev in range(len(lo) - 1):
        vals = {}
        for g in prog["order"][int(lo[lev]):int(lo[lev + 1])]:
            g = int(g)
            a, b = int(prog["left"][g]), int(prog["right"][g])
            x = consts[a & 0x7fffffff] if a & 0x80000000 else w[a]
            y = consts[b & 0x7fffffff] if b & 0x80000000 else w[b]
            assert x is not None and y is not None, "operand not ready: not a level schedule"
            vals[g] = (x * y if prog["op"][g] == 2 else x + y) % p
        for g, v in vals.items():                                   # a level's gates do not see one another
            assert w[g] is None
            w[g] = v
    npos = len(prog["op"])
    assert all(v is not None for v in w[:npos])
    return oracle.to_mont(oracle.ints_to_limbs([v if v is not None else 0 for v in w]))


def test_trace_program_reproduces_build_w(hp, oracle, model):
    """f3 on the device: the level-scheduled program (lgh_trace_program) executed the way the device executes it gives the W block
    build_w gives -- Poseidon (64 levels) and an expression-made circuit; input_positions follows the assignment convention"""
    c = hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "poseidon.r1cs"))
    inst = hp.LigeroInstance(c)
    wit = model.load_witness_json(os.path.join(GOLDEN, "poseidon_witness.json"))
    vals = oracle.to_mont(oracle.ints_to_limbs(wit[1:]))
    idx = list(range(1, len(wit)))
    wblk, ok = inst.build_w(idx, vals)
    prog = inst.trace_program()
    left, right, consts = inst.gate_map()
    assert np.array_equal(prog["constants"], consts)                              # one list of constants for both maps
    muls = prog["op"] == 2
    assert np.array_equal(prog["left"][muls], left[muls]) and np.array_equal(muls, left != 0xffffffff)
    assert prog["num_inputs"] == 264 and len(prog["level_off"]) - 1 == 64 and len(prog["order"]) == int((prog["op"] == 1).sum() + muls.sum())
    pos = inst.input_positions(idx)
    assert (prog["op"][pos] == 0).all() and len(set(pos.tolist())) == 264
    got = run_trace_program(oracle, model, prog, pos, vals, inst.m * inst.k)
    assert np.array_equal(got.reshape(wblk.shape), wblk)
    one = oracle.to_mont(oracle.ints_to_limbs([1]))[0]
    assert ok and all(np.array_equal(got[int(o)], one) for o in prog["outputs"])
    with pytest.raises(hp.HostPanic, match="non-variable"):
        inst.input_positions([0])
    x, y = hp.Expression.variable("x"), hp.Expression.variable("y")
    circ = ((x * (x + 3)) * y - y.pow(3) + x).to_arithmetic_circuit()
    inst2 = hp.LigeroInstance(circ, [circ.num_nodes() - 1])
    nodes = [circ.get_variable("x"), circ.get_variable("y")]
    v = np.stack([_mont(oracle, 5), _mont(oracle, 7)])
    w2, _ = inst2.build_w(nodes, v)
    got2 = run_trace_program(oracle, model, inst2.trace_program(), inst2.input_positions(nodes), v, inst2.m * inst2.k)
    assert np.array_equal(got2.reshape(w2.shape), w2)


def test_fast_host_product_equals_the_portable_one(tmp_path):
    """ligero_amd/csrc/host_fr.h: the mulx / adcx / adox Montgomery product (and its chain form without the final subtraction, which the
    sponge's S-box uses) against the portable CIOS on two million random operands, and the whole sponge with the fast path switched
    off (LG_HOST_NO_ADX=1) against the default in another process"""
    import subprocess
    import sys
    exe = str(tmp_path / "host_mul_bench")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "ligero_amd", "csrc"), "-o", exe, os.path.join(ROOT, "tools", "host_mul_bench.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "mismatching limbs on 2 M random products: 0" in r.stdout, r.stdout + r.stderr
    exe2 = str(tmp_path / "host_sbox_check")                 # the single-block x^17 (sbox17_lazy_adx), canonical and lazy inputs
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "ligero_amd", "csrc"), "-o", exe2, os.path.join(ROOT, "tools", "host_sbox_check.cpp")], check=True)
    r = subprocess.run([exe2], capture_output=True, text=True)
    assert r.returncode == 0 and "mismatching limbs: 0" in r.stdout, r.stdout + r.stderr
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from ligero_amd import host_pipeline as hp\n"
            "el = np.random.default_rng(7).integers(0, 2**62, size=(301, 4), dtype=np.uint64)\n"
            "s = hp.PoseidonSponge(); s.absorb_bytes(b'x' * 32); s.absorb_elements(el); a = s.squeeze_bytes(48); s.absorb_elements(el[:5]); b = s.squeeze_elements(3)\n"
            "print(a.hex(), b.tobytes().hex())\n") % ROOT
    outs = []
    for no_adx in ("0", "1"):
        env = dict(os.environ, LG_HOST_NO_ADX=no_adx)
        outs.append(subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, check=True).stdout.strip())
    assert outs[0] == outs[1] and len(outs[0]) > 100
