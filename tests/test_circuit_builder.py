"""CPU tests of the ArithmeticCircuit builder and evaluator mirror (ligero_amd/host/circuit.hpp through
include/ligero_host.h), restating the reference's own assertions in src/arithmetic_circuit/tests.rs:

  test_add_constants / test_mul_constants / test_pow_constants          tests.rs:107-131
  test_add_variables / test_mul_variables / test_pow_variable           tests.rs:133-160
  test_indicator                                                        tests.rs:163-172
  test_fibonacci / test_fibonacci_with_const                            tests.rs:243-293
  test_lemniscate_circuit / test_generate_3_by_3_determinant_circuit    tests.rs:52-104, 308-348
plus the label API (new_variable_with_label / get_variable, mod.rs:92-117) and the front half of
prove_with_labels (src/ligero/mod.rs:580-611): same preenc_u as the index form, reference panic messages."""
import numpy as np
import pytest

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


@pytest.fixture(scope="module")
def hp():
    from ligero_amd import host_pipeline
    host_pipeline.lib()
    return host_pipeline


@pytest.fixture(scope="module")
def fr(oracle):
    class Fr:
        @staticmethod
        def mont(v):
            return oracle.to_mont(oracle.ints_to_limbs([v % P]))[0]

        @staticmethod
        def monts(vs):
            return oracle.to_mont(oracle.ints_to_limbs([v % P for v in vs]))

        @staticmethod
        def int(limbs):
            return oracle.limbs_to_ints(oracle.from_mont(np.ascontiguousarray(limbs).reshape(1, 4)))[0]
    return Fr


def _evaluate(c, fr, assignment, node=None):
    """circuit.evaluate(vars) / evaluate_node(vars, node) as a Python int"""
    idx = [i for i, _ in assignment]
    vals = fr.monts([v for _, v in assignment]) if assignment else np.zeros((0, 4), dtype=np.uint64)
    return fr.int(c.evaluate_node(idx, vals, c.num_nodes() - 1 if node is None else node))


def test_add_mul_pow_constants(hp, fr):
    c = hp.ArithmeticCircuit()
    c.add(c.constant(fr.mont(1)), c.constant(fr.mont(2)))
    assert _evaluate(c, fr, []) == 3
    c = hp.ArithmeticCircuit()
    c.mul(c.constant(fr.mont(6)), c.constant(fr.mont(2)))
    assert _evaluate(c, fr, []) == 12
    c = hp.ArithmeticCircuit()
    c.pow(c.constant(fr.mont(2)), 5)
    assert _evaluate(c, fr, []) == 32


def test_add_mul_pow_variables(hp, fr):
    c = hp.ArithmeticCircuit()
    a, b = c.new_variables(2)
    c.add(a, b)
    assert _evaluate(c, fr, [(a, 2), (b, 3)]) == 5
    c = hp.ArithmeticCircuit()
    a, b = c.new_variables(2)
    c.mul(a, b)
    assert _evaluate(c, fr, [(a, 2), (b, 3)]) == 6
    c = hp.ArithmeticCircuit()
    a = c.new_variable()
    c.pow(a, 4)
    assert _evaluate(c, fr, [(a, 2)]) == 16
    assert c.num_gates() == 2 and c.num_nodes() == 3


def test_pow_bigint_matches_modular_power(hp, fr):
    e = (1 << 70) + 12345
    c = hp.ArithmeticCircuit()
    a = c.new_variable()
    c.pow_bigint(a, e)
    assert _evaluate(c, fr, [(a, 3)]) == pow(3, e, P)
    # square-and-multiply, most significant bit first: one squaring per bit below the top one, one more product per set bit
    assert c.num_gates() == (e.bit_length() - 1) + (bin(e).count("1") - 1)
    # exponent 0 has an empty bit list in the reference (mod.rs:171-176) and pow_binary then returns the node itself
    assert c.pow(a, 0) == a
    with pytest.raises(hp.HostPanic, match="not in the circuit"):
        c.pow(10**6, 3)


def test_indicator(hp, fr):
    c = hp.ArithmeticCircuit()
    a = c.new_variable()
    ind = c.indicator(a)
    rng = np.random.default_rng(5)
    v = int.from_bytes(rng.bytes(32), "little") % P
    assert _evaluate(c, fr, [(a, v)], ind) == 1
    assert _evaluate(c, fr, [(a, 0)], ind) == 0
    # x^(p-1): one gate per bit below the top of p - 1, plus one per further set bit
    assert c.num_gates() == ((P - 1).bit_length() - 1) + (bin(P - 1).count("1") - 1)


def test_fibonacci(hp, fr):
    c = hp.ArithmeticCircuit()
    f0, f1 = c.new_variable(), c.new_variable()
    a, b = f0, f1
    for _ in range(3, 50):
        a, b = b, c.add(a, b)
    assert _evaluate(c, fr, [(f0, 1), (f1, 1)], 42 - 1) == 267914296
    assert _evaluate(c, fr, [(f0, 5), (f1, 8)], 42 - 5) == 267914296
    c = hp.ArithmeticCircuit()
    f0, f1 = c.constant(fr.mont(1)), c.new_variable()
    a, b = f0, f1
    for _ in range(3, 50):
        a, b = b, c.add(a, b)
    assert _evaluate(c, fr, [(f1, 1)], 42 - 1) == 267914296


def _lemniscate(hp, fr):
    c = hp.ArithmeticCircuit()
    one = c.constant(fr.mont(1))
    x, y = c.new_variable(), c.new_variable()
    a, b = c.constant(fr.mont(120)), c.constant(fr.mont(80))
    x2, y2 = c.mul(x, x), c.mul(y, y)
    ax2, by2 = c.mul(a, x2), c.mul(b, y2)
    m_ax2 = c.minus(ax2)
    s = c.add(x2, y2)
    d = c.add(by2, m_ax2)
    s2 = c.mul(s, s)
    return c, c.add_nodes([s2, d, one])


def test_lemniscate_circuit(hp, fr):
    c, out = _lemniscate(hp, fr)
    assert out == c.num_nodes() - 1
    assert _evaluate(c, fr, [(1, 8), (2, 4)]) == 1
    assert _evaluate(c, fr, [(1, 8), (2, 5)]) != 1


def _determinant(hp, fr):
    c = hp.ArithmeticCircuit()
    one = c.constant(fr.mont(1))
    v = c.new_variables(9)
    det = c.new_variable()
    aei, bfg, cdh = c.mul_nodes([v[0], v[4], v[8]]), c.mul_nodes([v[1], v[5], v[6]]), c.mul_nodes([v[2], v[3], v[7]])
    ceg, bdi, afh = c.mul_nodes([v[2], v[4], v[6]]), c.mul_nodes([v[1], v[3], v[8]]), c.mul_nodes([v[0], v[5], v[7]])
    s1, s2 = c.add_nodes([aei, bfg, cdh]), c.add_nodes([ceg, bdi, afh])
    c.add_nodes([s1, c.minus(s2), c.minus(det), one])
    return c


def test_3_by_3_determinant_circuit(hp, fr):
    c = _determinant(hp, fr)
    assert _evaluate(c, fr, [(i, i) for i in range(1, 10)] + [(10, 0)]) == 1
    m = [2, 0, -1, 3, 5, 2, -4, 1, 4]
    assert _evaluate(c, fr, [(i + 1, m[i]) for i in range(9)] + [(10, 13)]) == 1
    assert _evaluate(c, fr, [(i + 1, m[i]) for i in range(9)] + [(10, 12)]) != 1


def test_scalar_product(hp, fr):
    c = hp.ArithmeticCircuit()
    xs, ys = c.new_variables(3), c.new_variables(3)
    sp = c.scalar_product(xs, ys)
    assert c.num_gates() == 3 + 2                                   # no 1 * x / 0 * x shortcuts (mod.rs:225-227)
    assert _evaluate(c, fr, list(zip(xs, [1, 2, 3])) + list(zip(ys, [4, 5, 6])), sp) == 32


def test_evaluate_multioutput_is_demand_driven_and_in_node_order(hp, fr):
    c = hp.ArithmeticCircuit()
    x, y, z = c.new_variables(3)
    xy = c.mul(x, y)
    yz = c.mul(y, z)
    xx = c.mul(x, x)
    # outputs listed out of order and twice: values come back in node order, each once (filter_map over nodes, mod.rs:381-387)
    got = c.evaluate_multioutput([x, y], fr.monts([3, 5]), [xx, xy, xx])
    assert [fr.int(g) for g in got] == [15, 9]
    # z is not needed for these outputs; asking for yz needs it (inner_evaluate's panic, mod.rs:255)
    with pytest.raises(hp.HostPanic, match="Uninitialised variable"):
        c.evaluate_multioutput([x, y], fr.monts([3, 5]), [yz])
    with pytest.raises(hp.HostPanic, match="Value supplied for non-variable node"):
        c.evaluate_multioutput([xy], fr.monts([3]), [xx])
    # duplicates: the latest value in the list is the one used (mod.rs:343-345)
    got = c.evaluate_multioutput([x, x], fr.monts([3, 4]), [xx])
    assert fr.int(got[0]) == 16


def test_variable_labels(hp, fr):
    c = hp.ArithmeticCircuit()
    x = c.new_variable_with_label("x")
    v0 = c.new_variable()                                           # "var_1": named after the variable count (mod.rs:107-109)
    y = c.new_variable_with_label("y")
    assert (c.get_variable("x"), c.get_variable("var_1"), c.get_variable("y")) == (x, v0, y)
    with pytest.raises(hp.HostPanic, match="Variable label already in use"):
        c.new_variable_with_label("x")
    with pytest.raises(hp.HostPanic, match="Variable not in circuit"):
        c.get_variable("nope")
    c2 = hp.ArithmeticCircuit()
    c2.new_variable_with_label("var_1")
    with pytest.raises(hp.HostPanic, match="Variable label already in use"):   # new_variable's documented panic (mod.rs:105-106)
        c2.new_variable()


def _multioutput_1(hp, fr):
    """src/ligero/tests.rs:245-266"""
    c = hp.ArithmeticCircuit()
    x, y = c.new_variable_with_label("x"), c.new_variable_with_label("y")
    c1, c2, c3 = c.constant(fr.mont(-9 + 1)), c.constant(fr.mont(-64 + 1)), c.constant(fr.mont(-7 + 1))
    x2 = c.mul(x, x)
    y3 = c.pow(y, 3)
    s = c.add(x, y)
    return c, (x, y), [c.add(x2, c1), c.add(y3, c2), c.add(s, c3)]


def test_prove_with_labels_front_half(hp, fr):
    """labels resolve in the FORMATTED circuit (after insert_one) straight to prove_inner; the index form bumps first
    (mod.rs:449-452 vs 594-609): both must give the same preenc_u"""
    c, (x, y), outs = _multioutput_1(hp, fr)
    inst = hp.LigeroInstance(c, outs)
    by_index, ok1 = inst.build_preenc_u([x, y], fr.monts([3, 4]))
    by_label, ok2 = inst.build_preenc_u_with_labels(["x", "y"], fr.monts([3, 4]))
    assert ok1 and ok2 and np.array_equal(by_index, by_label)
    swapped, ok3 = inst.build_preenc_u_with_labels(["y", "x"], fr.monts([4, 3]))
    assert ok3 and np.array_equal(by_index, swapped)
    _, bad = inst.build_preenc_u_with_labels(["x", "y"], fr.monts([3, 5]))
    assert not bad
    with pytest.raises(hp.HostPanic, match="Variable not found: w"):
        inst.build_preenc_u_with_labels(["x", "w"], fr.monts([3, 4]))
    with pytest.raises(hp.HostPanic, match="Uninitialised variable"):
        inst.build_preenc_u_with_labels(["x"], fr.monts([3]))


def test_prove_inner_panics_on_a_gate_no_output_depends_on(hp, fr):
    """prove_inner expects a value on EVERY node (mod.rs:476-478) and only evaluates what the outputs depend on: a
    dangling gate is the reference's "Uninitialised variable. Make sure the circuit only contains nodes ..." panic"""
    c, (x, y), outs = _multioutput_1(hp, fr)
    c.mul(x, y)                                                     # no output uses it
    inst = hp.LigeroInstance(c, outs)
    with pytest.raises(hp.HostPanic, match="only contains nodes upon which the final output truly depends"):
        inst.build_preenc_u([x, y], fr.monts([3, 4]))


# ---------------------------------------------------------------- Expression front end (src/expression)
def test_cpp_expression_suite():
    """the reference's expression tests (exact node numbering included) and test_constant_filtering, restated in C++ over
    the operator API (ligero_amd/host/test_expression.cpp)"""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "ligero_amd", "host", "test_expression")
    assert os.path.exists(exe), "run `make -C ligero_amd/host`"
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    for name in ("test_get_variables", "test_same_reference", "test_to_arithmetic_circuit_1", "test_to_arithmetic_circuit_2",
                 "test_to_arithmetic_circuit_3", "test_to_arithmetic_circuit_4", "test_to_arithmetic_circuit_5", "test_constant_filtering",
                 "test_mat_mul_sparse"):
        assert name + ": ok" in out.stdout


def _lemniscate_expression(hp):
    """src/expression/tests.rs:21-26"""
    x, y = hp.Expression.variable("x"), hp.Expression.variable("y")
    return 1 + (x.pow(2) + y.pow(2)).pow(2) - 120 * x.pow(2) + 80 * y.pow(2)


def _determinant_expression(hp):
    """src/expression/tests.rs:28-60"""
    m = [[hp.Expression.variable(f"x_{i}_{j}") for j in range(3)] for i in range(3)]

    def diagonals(js):
        total = None
        for k in range(3):
            prod = None
            for i in range(3):
                f = m[i][(js[i] + k) % 3]
                prod = f if prod is None else prod * f
            total = prod if total is None else total + prod
        return total
    return 1 + (diagonals([0, 4, 8]) - diagonals([2, 4, 6]) - hp.Expression.variable("det"))


def test_python_expression_matches_reference_numbering(hp, fr):
    """src/expression/tests.rs:62-74 and 303-345 through the ctypes mirror"""
    c = _lemniscate_expression(hp).to_arithmetic_circuit()
    assert (c.get_variable("x"), c.get_variable("y")) == (10, 8)
    a, b, c0 = (hp.Expression.variable(s) for s in "abc")
    circ = ((a + b) * (c0 + a * b)).to_arithmetic_circuit()
    want = [("Mul", 5, 2), ("Add", 4, 3), ("Variable", "a"), ("Variable", "b"), ("Add", 1, 0), ("Variable", "c"), ("Mul", 4, 3)][::-1]
    assert [circ.node(i) for i in range(circ.num_nodes())] == want
    idx = [circ.get_variable(s) for s in "abc"]
    assert fr.int(circ.evaluate_node(idx, fr.monts([3, 2, 1]), circ.last())) == 35


def test_python_expression_circuits_evaluate_to_one(hp, fr):
    c = _lemniscate_expression(hp).to_arithmetic_circuit()
    assert fr.int(c.evaluate_node([c.get_variable("x"), c.get_variable("y")], fr.monts([8, 4]), c.last())) == 1
    d = _determinant_expression(hp).to_arithmetic_circuit()
    labels = [f"x_{i}_{j}" for i in range(3) for j in range(3)] + ["det"]
    vals = [(3 * i + j) ** 2 for i in range(3) for j in range(3)] + [-216]
    assert fr.int(d.evaluate_node([d.get_variable(s) for s in labels], fr.monts(vals), d.last())) == 1


def test_expression_circuit_through_ligero_front_half(hp, fr):
    """test_proof_and_verify_expression's input side (src/ligero/tests.rs:172-184): LigeroCircuit::new on a circuit whose gates
    refer forwards, assignment by get_variable index and by label give the same preenc_u, outputs all one"""
    c = _lemniscate_expression(hp).to_arithmetic_circuit()
    inst = hp.LigeroInstance(c, [c.last()])
    by_index, ok = inst.build_preenc_u([c.get_variable("x"), c.get_variable("y")], fr.monts([8, 4]))
    by_label, ok2 = inst.build_preenc_u_with_labels(["x", "y"], fr.monts([8, 4]))
    assert ok and ok2 and np.array_equal(by_index, by_label)
    _, bad = inst.build_preenc_u_with_labels(["x", "y"], fr.monts([9, 4]))
    assert not bad


def test_multiplication_r1cs(hp, fr):
    """src/arithmetic_circuit/tests.rs:174-187: multiplication.r1cs compiled by from_constraint_system; (a, b, c) = (6, 3, 2) on
    wires 1..3 makes the last node (the single output) evaluate to 1"""
    import os
    from conftest import GOLDEN
    c = hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "multiplication.r1cs"))
    assert _evaluate(c, fr, [(1, 6), (2, 3), (3, 2)]) == 1
    assert _evaluate(c, fr, [(1, 6), (2, 3), (3, 3)]) != 1


def test_cube_multioutput(hp, fr):
    """src/arithmetic_circuit/tests.rs:189-241: cube.r1cs -> 15 nodes, both outputs 1 at (x, x^3) = (3, 9) ... wires (1, 2) = (3, 9);
    three ways of building x^3 - 26 give the SAME circuit (3 gates) and evaluate to 1 at x = 3"""
    import os
    from conftest import GOLDEN
    c = hp.ArithmeticCircuit.from_r1cs(os.path.join(GOLDEN, "cube.r1cs"))
    assert c.num_nodes() == 15
    got = c.evaluate_multioutput([1, 2], fr.monts([3, 9]), c.outputs)
    assert [fr.int(g) for g in got] == [1, 1]

    def clever(build):
        k = hp.ArithmeticCircuit()
        x = k.new_variable()
        cubed = build(k, x)
        k.add(cubed, k.constant(fr.mont(-26)))
        return k
    circuits = [clever(lambda k, x: k.pow(x, 3)), clever(lambda k, x: k.mul(k.mul(x, x), x)), clever(lambda k, x: k.mul_nodes([x, x, x]))]
    for k in circuits:
        assert _evaluate(k, fr, [(0, 3)]) == 1
        assert k.num_gates() == 3

    def nodes(k):
        out = []
        for i in range(k.num_nodes()):
            nd = k.node(i)
            out.append((nd[0], fr.int(nd[1])) if nd[0] == "Constant" else nd)
        return out
    assert nodes(circuits[0]) == nodes(circuits[1]) == nodes(circuits[2])            # assert_eq!(clever_circuit, another_clever_circuit) ...
