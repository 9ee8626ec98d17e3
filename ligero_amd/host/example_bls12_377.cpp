// The reference's test_prove_and_verify_bls12_377 (src/ligero/tests.rs:186-193 with the helper at 144-170), mirrored over the
// templated host classes and the device library's generic-field path: the ten-node circuit of generate_bls12_377_circuit
// (src/arithmetic_circuit/tests.rs:17-33: y^2 = x^3 + 1 over ark_bls12_377::Fq), LigeroCircuit::new -> (m, k) = (4, 4) as the
// reference's test_construction_bls12_377 asserts (tests.rs:35-142), prove + verify accept a point of the curve and reject
// the assignment with x + 1.  The reference draws a random G1 point from test_rng(); here: the G1 generator and its double.
//
//   usage: example_bls12_377      (prints one line per case, exit code 0 = every expectation held)
#include <cstdio>

#include "expression.hpp"
#include "prover.hpp"

using namespace ligero;
using E = Fq377;
using F = Field<E>;

static bool proof_and_verify(const ArithmeticCircuitT<E>& circuit, const std::vector<std::pair<size_t, E>>& vars) {   // tests.rs:144-158
    const size_t output_node = circuit.last();
    LigeroInstanceT<E> ligero_circuit(circuit, {output_node}, 128);
    if (ligero_circuit.m != 4 || ligero_circuit.k != 4 || ligero_circuit.n != 32 || ligero_circuit.t != 32) {
        fprintf(stderr, "unexpected dimensions m=%zu k=%zu n=%zu t=%zu\n", ligero_circuit.m, ligero_circuit.k, ligero_circuit.n, ligero_circuit.t);
        return false;
    }
    HipLigeroT<E> prover(ligero_circuit);
    PoseidonSpongeT<E> sponge = PoseidonSpongeT<E>::test_sponge();
    PoseidonSpongeT<E> ps = sponge, vs = sponge;
    const LigeroProofT<E> proof = prover.prove(vars, ps);
    return prover.verify(proof, vs);
}

static int run();
int main() {
    try {
        return run();
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
}
static int run() {
    // generate_bls12_377_circuit
    ArithmeticCircuitT<E> circuit;
    const size_t one = circuit.constant(F::one());
    const size_t x = circuit.new_variable_with_label("x");
    const size_t y = circuit.new_variable_with_label("y");
    const size_t y_squared = circuit.pow(y, 2);
    const size_t minus_y_squared = circuit.minus(y_squared);
    const size_t x_cubed = circuit.pow(x, 3);
    circuit.add_nodes({x_cubed, one, minus_y_squared, one});
    if (x != 1 || y != 2) return 2;                       // the reference assigns (1, x), (2, y)

    // affine coordinates of the BLS12-377 G1 generator (canonical), and of 2 G computed here by the tangent rule
    const E gx = {{0xeab9b16eb21be9efULL, 0xd5481512ffcd394eULL, 0x188282c8bd37cb5cULL, 0x85951e2caa9d41bbULL, 0xc8fc6225bf87ff54ULL, 0x008848defe740a67ULL}};
    const E gy = {{0xfd82de55559c8ea6ULL, 0xc2fe3d3634a9591aULL, 0x6d182ad44fb82305ULL, 0xbd7fb348ca3e52d9ULL, 0x1f674f5d30afeec4ULL, 0x01914a69c5102effULL}};
    const E X = F::to_mont(gx), Y = F::to_mont(gy);
    auto inv = [](const E& a) {   // a^(p-2)
        E e = F::modulus();
        e.l[0] -= 2;
        E acc = F::one(), b = a;
        for (int i = 0; i < F::kLimbs; i++)
            for (int bit = 0; bit < 64; bit++) {
                if ((e.l[i] >> bit) & 1) acc = F::mul(acc, b);
                b = F::mul(b, b);
            }
        return acc;
    };
    const E three = F::from_u64(3), two = F::from_u64(2);
    const E lam = F::mul(F::mul(three, F::mul(X, X)), inv(F::mul(two, Y)));          // 3 x^2 / (2 y)   (a = 0)
    const E X2 = F::sub(F::mul(lam, lam), F::mul(two, X));
    const E Y2 = F::sub(F::mul(lam, F::sub(X, X2)), Y);

    int failures = 0;
    {   // test_construction_bls12_377 (tests.rs:35-142): the induced matrices P_x, P_y, P_z, P_add inside A = [[I, -P_xyz], [0, P_add]]
        LigeroInstanceT<E> lc(circuit, {circuit.last()}, 128);
        const size_t mk = lc.m * lc.k;   // 16
        const E one_e = F::one(), minus = F::neg(F::one());
        struct Ent { int sign; size_t col; };
        auto row_is = [&](size_t row, bool identity, std::initializer_list<Ent> want, int flip) {
            const auto r = lc.a.row(row);
            size_t pos = 0;
            if (identity) { if (r.size() < 1 || !F::eq(r[0].first, one_e) || r[0].second != row) return false; pos = 1; }
            if (r.size() != pos + want.size()) return false;
            for (const Ent& w : want) {
                const E v = (w.sign * flip > 0) ? one_e : minus;
                if (!F::eq(r[pos].first, v) || r[pos].second != 3 * mk + w.col) return false;
                pos++;
            }
            return true;
        };
        bool ok = lc.a.num_rows() == 4 * mk && lc.a.num_cols == 4 * mk;
        // rows 3..6 of P_x, P_y, P_z (negated in A), rows 7..10 of P_add
        const Ent px[4][1] = {{{1, 2}}, {{-1, 0}}, {{1, 1}}, {{1, 5}}}, py[4][1] = {{{1, 2}}, {{1, 3}}, {{1, 1}}, {{1, 1}}}, pz[4][1] = {{{1, 3}}, {{1, 4}}, {{1, 5}}, {{1, 6}}};
        for (int i = 0; i < 4; i++) {
            ok = ok && row_is(3 + i, true, {px[i][0]}, -1) && row_is(mk + 3 + i, true, {py[i][0]}, -1) && row_is(2 * mk + 3 + i, true, {pz[i][0]}, -1);
            ok = ok && lc.a.row(3 * mk + 3 + i).empty();
        }
        ok = ok && row_is(3 * mk + 7, false, {{1, 6}, {1, 0}, {-1, 7}}, 1) && row_is(3 * mk + 8, false, {{1, 7}, {1, 4}, {-1, 8}}, 1) &&
             row_is(3 * mk + 9, false, {{1, 8}, {1, 0}, {-1, 9}}, 1) && row_is(3 * mk + 10, false, {{1, 8}, {1, 0}, {-1, 0}}, 1);
        for (size_t r0 : {size_t{0}, size_t{1}, size_t{2}, size_t{7}, size_t{11}, size_t{15}}) ok = ok && row_is(r0, true, {}, -1);
        printf("test_construction_bls12_377 (A matrix tables): %s\n", ok ? "ok" : "MISMATCH");
        fflush(stdout);
        failures += ok ? 0 : 1;
    }
    const std::pair<const char*, std::pair<E, E>> points[] = {{"G", {X, Y}}, {"2G", {X2, Y2}}};
    for (const auto& pt : points) {
        std::vector<std::pair<size_t, E>> vars = {{1, pt.second.first}, {2, pt.second.second}};
        const bool ok = proof_and_verify(circuit, vars);                     // assert!(proof_and_verify(circuit.clone(), vars))
        std::vector<std::pair<size_t, E>> invalid = vars;
        invalid[0].second = F::add(invalid[0].second, F::one());             // invalid_assignment[0].1 += F::ONE
        const bool bad = proof_and_verify(circuit, invalid);                 // assert!(!proof_and_verify(circuit, invalid_assignment))
        printf("bls12_377 %s: valid assignment %s, x + 1 %s\n", pt.first, ok ? "accepted" : "REJECTED", bad ? "ACCEPTED" : "rejected");
        failures += (ok ? 0 : 1) + (bad ? 1 : 0);
    }
    {   // test_proof_and_verify_expression(generate_bls12_377_expression(), [("x", x), ("y", y)]) (tests.rs:172-184, 192;
        // expression/tests.rs:13-18): the circuit numbered from the expression DAG, variables found by label; and the
        // same statement through prove_with_labels
        const auto xe = ExpressionT<E>::variable("x"), ye = ExpressionT<E>::variable("y");
        const ArithmeticCircuitT<E> ec = (1 + (1 + xe.pow(3) - ye.pow(2))).to_arithmetic_circuit();
        std::vector<std::pair<size_t, E>> vars = {{ec.get_variable("x"), X}, {ec.get_variable("y"), Y}};
        const bool ok = ec.get_variable("x") == 4 && ec.get_variable("y") == 0 && proof_and_verify(ec, vars);
        vars[0].second = F::add(vars[0].second, F::one());
        const bool bad = proof_and_verify(ec, vars);
        LigeroInstanceT<E> lc(ec, {ec.last()}, 128);
        HipLigeroT<E> prover(lc);
        PoseidonSpongeT<E> ps = PoseidonSpongeT<E>::test_sponge(), vs = ps;
        const bool labelled = prover.verify(prover.prove_with_labels({{"x", X2}, {"y", Y2}}, ps), vs);
        printf("bls12_377 expression: valid assignment %s, x + 1 %s, prove_with_labels(2G) %s\n", ok ? "accepted" : "REJECTED",
               bad ? "ACCEPTED" : "rejected", labelled ? "accepted" : "REJECTED");
        failures += (ok ? 0 : 1) + (bad ? 1 : 0) + (labelled ? 0 : 1);
    }
    printf(failures ? "FAILED\n" : "test_prove_and_verify_bls12_377: ok\n");
    return failures ? 1 : 0;
}
