// The reference's assertions about the Expression front end and filter_constants, restated over expression.hpp and
// circuit.hpp (no device code; run by tests/test_circuit_builder.py):
//   src/expression/tests.rs:62-74     test_get_variables          exact variable indices of two compiled expressions
//   src/expression/tests.rs:76-97     test_same_reference         a clone is the same node
//   src/expression/tests.rs:99-141    test_addition / test_multiplication / test_subtraction
//   src/expression/tests.rs:143-157   test_some_operations
//   src/expression/tests.rs:214-277   test_to_arithmetic_circuit_1   exact nodes, constants map and evaluation trace
//   src/expression/tests.rs:303-345   test_to_arithmetic_circuit_2   exact nodes and trace (a gate referring forwards)
//   src/expression/tests.rs:347-387   test_to_arithmetic_circuit_3 / _4 / _5
//   src/arithmetic_circuit/tests.rs:350-393   test_constant_filtering
//   src/matrices/mod.rs:195-209               test_mat_mul_sparse
// usage: test_expression      (one line per test; exit code = number of failures)
#include <algorithm>
#include <array>
#include <cstdio>

#include "expression.hpp"

using namespace ligero;
using FqE = Fq377;
using FF = Field<Fr>;
using FQ = Field<FqE>;

static Fr fr(long v) { return Expression::field_from_int(v); }
static FqE fq(long v) { return ExpressionT<FqE>::field_from_int(v); }

template <class E>
static bool same_nodes(const std::vector<NodeT<E>>& a, const std::vector<NodeT<E>>& b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++) {
        if (a[i].kind != b[i].kind) return false;
        switch (a[i].kind) {
            case NodeBase::Variable: if (a[i].label != b[i].label) return false; break;
            case NodeBase::Constant: if (!Field<E>::eq(a[i].value, b[i].value)) return false; break;
            default: if (a[i].l != b[i].l || a[i].r != b[i].r) return false;
        }
    }
    return true;
}
template <class E> static NodeT<E> var(const char* l) { NodeT<E> n; n.kind = NodeBase::Variable; n.label = l; return n; }
template <class E> static NodeT<E> cst(const E& v) { NodeT<E> n; n.kind = NodeBase::Constant; n.value = v; return n; }
template <class E> static NodeT<E> add(size_t l, size_t r) { NodeT<E> n; n.kind = NodeBase::Add; n.l = l; n.r = r; return n; }
template <class E> static NodeT<E> mul(size_t l, size_t r) { NodeT<E> n; n.kind = NodeBase::Mul; n.l = l; n.r = r; return n; }

static ExpressionT<FqE> generate_bls12_377_expression() {                 // expression/tests.rs:13-18
    auto x = ExpressionT<FqE>::variable("x"), y = ExpressionT<FqE>::variable("y");
    return 1 + (1 + x.pow(3) - y.pow(2));
}
static Expression generate_lemniscate_expression() {                      // expression/tests.rs:21-26
    auto x = Expression::variable("x"), y = Expression::variable("y");
    return 1 + (x.pow(2) + y.pow(2)).pow(2) - 120 * x.pow(2) + 80 * y.pow(2);
}
static Expression generate_3_by_3_determinant_expression() {              // expression/tests.rs:28-60
    std::vector<std::vector<Expression>> m;
    for (int i = 0; i < 3; i++) {
        m.emplace_back();
        for (int j = 0; j < 3; j++) m[i].push_back(Expression::variable("x_" + std::to_string(i) + "_" + std::to_string(j)));
    }
    auto diagonals = [&](std::array<int, 3> js) {
        std::vector<Expression> terms;
        for (int k = 0; k < 3; k++) {
            std::vector<Expression> f;
            for (int i = 0; i < 3; i++) f.push_back(m[i][(js[i] + k) % 3]);
            terms.push_back(Expression::product(f));
        }
        return Expression::sum(terms);
    };
    const Expression positive = diagonals({0, 4, 8}), negative = diagonals({2, 4, 6});
    return 1 + (positive - negative - Expression::variable("det"));
}

static int failures = 0;
static void report(const char* name, bool ok) {
    printf("%s: %s\n", name, ok ? "ok" : "FAILED");
    failures += ok ? 0 : 1;
}

int main() {
    try {
        {
            const auto c1 = generate_bls12_377_expression().to_arithmetic_circuit();
            const auto c2 = generate_lemniscate_expression().to_arithmetic_circuit();
            report("test_get_variables", c1.get_variable("x") == 4 && c1.get_variable("y") == 0 && c2.get_variable("x") == 10 && c2.get_variable("y") == 8);
        }
        {
            Expression f1 = Expression::variable("x"), f2 = Expression::variable("y");
            const Expression original_f1 = f1;
            for (int i = 0; i < 10; i++) {
                Expression next = f1 + f2;
                f1 = f2;
                f2 = next;
            }
            const auto c = (f2 * original_f1).to_arithmetic_circuit();
            // the product's right operand IS the first variable: one "x" node in the circuit, and it is the root's right operand
            size_t xs = 0;
            for (const auto& n : c.nodes) xs += (n.kind == NodeBase::Variable && n.label == "x");
            report("test_same_reference", xs == 1 && c.nodes.back().kind == NodeBase::Mul && c.nodes.back().r == c.get_variable("x"));
        }
        {
            const auto x = Expression::variable("x"), y = Expression::variable("y");
            const std::vector<std::pair<std::string, Fr>> v = {{"x", fr(3)}, {"y", fr(5)}};
            report("test_addition", FF::eq((x + y).to_arithmetic_circuit().evaluate_with_labels(v), fr(8)));
            report("test_multiplication", FF::eq((x * y).to_arithmetic_circuit().evaluate_with_labels(v), fr(15)));
            report("test_subtraction", FF::eq((x - y).to_arithmetic_circuit().evaluate_with_labels(v), fr(-2)));
        }
        {
            const Fr want = FF::add(FF::add(FF::pow_u64(fr(5), 3), FF::pow_u64(FF::sub(fr(3), FF::one()), 11)), fr(13));
            const auto xe = Expression::constant(fr(5)), ye = Expression::constant(fr(3));
            const auto out = 13 + xe.pow(3) + (ye - FF::one()).pow(11);
            report("test_some_operations", FF::eq(out.to_arithmetic_circuit().evaluate({}), want));
        }
        {
            const auto x = Expression::variable("x"), y = Expression::variable("y");
            const auto e = (3 + 2 * (x * y)) + ((3 + 2 * x) * (1 + 2 * y));
            const auto c = e.to_arithmetic_circuit();
            std::vector<Node> want = {add<Fr>(12, 7), add<Fr>(5, 11), mul<Fr>(0, 10), mul<Fr>(9, 8), var<Fr>("x"), var<Fr>("y"), mul<Fr>(6, 3), add<Fr>(5, 4),
                                      cst(fr(3)), mul<Fr>(0, 9), add<Fr>(2, 1), cst(fr(1)), mul<Fr>(0, 8), cst(fr(2))};
            std::reverse(want.begin(), want.end());
            bool ok = same_nodes(c.nodes, want);
            ok = ok && c.constants.size() == 3 && c.constants.at(fr(3)) == 5 && c.constants.at(fr(1)) == 2 && c.constants.at(fr(2)) == 0;
            const auto t = c.evaluation_trace(c.with_labels({{"x", fr(3)}, {"y", fr(2)}}), 13);
            const long vals[14] = {2, 4, 1, 5, 6, 3, 9, 45, 2, 3, 6, 12, 15, 60};
            for (size_t i = 0; i < 14; i++) ok = ok && t.set[i] && FF::eq(t.value[i], fr(vals[i]));
            report("test_to_arithmetic_circuit_1", ok);
        }
        {
            const auto a = Expression::variable("a"), b = Expression::variable("b"), c0 = Expression::variable("c");
            const auto c = ((a + b) * (c0 + a * b)).to_arithmetic_circuit();
            std::vector<Node> want = {mul<Fr>(5, 2), add<Fr>(4, 3), var<Fr>("a"), var<Fr>("b"), add<Fr>(1, 0), var<Fr>("c"), mul<Fr>(4, 3)};
            std::reverse(want.begin(), want.end());
            bool ok = same_nodes(c.nodes, want) && c.constants.empty();
            const auto t = c.evaluation_trace(c.with_labels({{"a", fr(3)}, {"b", fr(2)}, {"c", fr(1)}}), 6);
            const long vals[7] = {6, 1, 7, 2, 3, 5, 35};
            for (size_t i = 0; i < 7; i++) ok = ok && t.set[i] && FF::eq(t.value[i], fr(vals[i]));
            report("test_to_arithmetic_circuit_2", ok);
        }
        {
            const auto c = generate_3_by_3_determinant_expression().to_arithmetic_circuit();
            std::vector<std::pair<std::string, Fr>> v;
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) v.emplace_back("x_" + std::to_string(i) + "_" + std::to_string(j), fr((3 * i + j) * (3 * i + j)));
            v.emplace_back("det", fr(-216));
            report("test_to_arithmetic_circuit_3", FF::eq(c.evaluate_with_labels(v), FF::one()));
        }
        {
            // a point of y^2 = x^3 + 1 over Fq: the reference draws a random G1 point; (2, 3) is on the curve over any field
            const auto c = generate_bls12_377_expression().to_arithmetic_circuit();
            const bool on = FQ::eq(c.evaluate_with_labels({{"x", fq(2)}, {"y", fq(3)}}), FQ::one());
            const bool off = FQ::eq(c.evaluate_with_labels({{"x", fq(2)}, {"y", fq(4)}}), FQ::one());
            report("test_to_arithmetic_circuit_4", on && !off);
        }
        {   // src/arithmetic_circuit/tests.rs:17-33, 296-306 test_bls12_377_circuit: the builder-made y^2 = x^3 + 1 circuit over Fq
            ArithmeticCircuitT<FqE> c;
            const size_t one = c.constant(FQ::one());
            const size_t x = c.new_variable(), y = c.new_variable();
            const size_t y2 = c.pow(y, 2), my2 = c.minus(y2), x3 = c.pow(x, 3);
            c.add_nodes({x3, one, my2, one});
            report("test_bls12_377_circuit", x == 1 && y == 2 && FQ::eq(c.evaluate({{1, fq(2)}, {2, fq(3)}}), FQ::one()) &&
                                                 !FQ::eq(c.evaluate({{1, fq(2)}, {2, fq(5)}}), FQ::one()));
        }
        {
            const auto c = generate_lemniscate_expression().to_arithmetic_circuit();
            report("test_to_arithmetic_circuit_5", FF::eq(c.evaluate_with_labels({{"x", fr(8)}, {"y", fr(4)}}), FF::one()));
        }
        {
            using N = NodeT<FqE>;
            const std::vector<N> nodes = {var<FqE>("x"), cst(fq(3)), cst(fq(3)), var<FqE>("y"), mul<FqE>(18, 2), cst(fq(-1)), mul<FqE>(4, 1), mul<FqE>(2, 2),
                                          cst(fq(4)), mul<FqE>(7, 7), cst(fq(-1)), add<FqE>(8, 5), add<FqE>(8, 14), mul<FqE>(17, 10), cst(fq(3)), cst(fq(-2)),
                                          var<FqE>("z"), cst(fq(-1)), add<FqE>(12, 5)};
            const std::vector<N> want = {var<FqE>("x"), cst(fq(3)), var<FqE>("y"), mul<FqE>(14, 1), cst(fq(-1)), mul<FqE>(3, 1), mul<FqE>(1, 1), cst(fq(4)),
                                         mul<FqE>(6, 6), add<FqE>(7, 4), add<FqE>(7, 1), mul<FqE>(4, 4), cst(fq(-2)), var<FqE>("z"), add<FqE>(10, 4)};
            report("test_constant_filtering", same_nodes(ArithmeticCircuitT<FqE>::filter_constants(nodes).first, want));
        }
        {   // src/matrices/mod.rs:195-209 test_mat_mul_sparse: the reference's known answer for SparseMatrix::row_mul
            SparseMatrixT<Fr> m(3);
            m.push_row({{fr(1), 0}, {fr(8), 2}});
            m.push_row({{fr(4), 1}, {fr(5), 2}});
            const std::vector<Fr> got = m.row_mul({fr(-5), fr(17)});
            report("test_mat_mul_sparse", got.size() == 3 && FF::eq(got[0], fr(-5)) && FF::eq(got[1], fr(68)) && FF::eq(got[2], fr(45)));
        }
        {
            // a long chain is built, compiled, evaluated and released without recursion
            Expression acc = Expression::variable("x");
            for (int i = 0; i < 300000; i++) acc += FF::one();
            const auto c = acc.to_arithmetic_circuit();
            report("long_chain", c.num_nodes() == 300002 && FF::eq(c.evaluate_with_labels({{"x", fr(5)}}), fr(300005)));
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 100;
    }
    return failures;
}
