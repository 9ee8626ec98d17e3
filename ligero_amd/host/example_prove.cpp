// From the reference's input files to a verified proof, C++ only:
//     example_prove <circuit.r1cs> <witness.json | witness.wtns> [security bits]
// .r1cs -> ArithmeticCircuit::from_constraint_system -> LigeroCircuit::new -> prove (device) -> verify,
// i.e. src/ligero/tests.rs:365-415 (test_poseidon) as a program.  Prints the commitment root.
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "prover.hpp"

using namespace ligero;

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <circuit.r1cs> <witness.json|witness.wtns> [lambda]\n", argv[0]);
        return 2;
    }
    try {
        const size_t lambda = argc > 3 ? (size_t)std::atoi(argv[3]) : 128;
        const R1cs r1cs = read_r1cs(argv[1]);
        auto compiled = ArithmeticCircuit::from_constraint_system(r1cs);
        ArithmeticCircuit circuit = std::move(compiled.first);
        const std::vector<size_t> outputs = compiled.second;
        const std::vector<Fr> witness = read_witness(argv[2]);
        if (witness.size() != r1cs.n_wires) throw std::runtime_error("witness length does not match the circuit's wire count");
        LigeroInstance inst(std::move(circuit), outputs, lambda);
        std::printf("m = %zu  k = %zu  n = %zu  t = %zu  nnz(A) = %zu\n", inst.m, inst.k, inst.n, inst.t, inst.a.nnz());
        // wire i is variable node i (arithmetic_circuit/mod.rs:459-460); wire 0 is the constant one
        std::vector<std::pair<size_t, Fr>> assignment;
        for (size_t i = 1; i < witness.size(); i++) assignment.emplace_back(i, witness[i]);
        HipLigero ligero(inst);
        PoseidonSponge prover_sponge = PoseidonSponge::test_sponge(), verifier_sponge = PoseidonSponge::test_sponge();
        const auto t0 = std::chrono::steady_clock::now();
        const LigeroProof proof = ligero.prove(assignment, prover_sponge);
        const auto t1 = std::chrono::steady_clock::now();
        const bool ok = ligero.verify(proof, verifier_sponge);
        const auto t2 = std::chrono::steady_clock::now();
        std::printf("u_root = ");
        for (uint8_t b : proof.u_root) std::printf("%02x", b);
        std::printf("\nprove %.2f ms  verify %.2f ms  accepted = %s\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(t2 - t1).count(), ok ? "true" : "false");
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
}
