// ThreadSanitizer harness (tests/test_sanitizers.py): the host phases of the batch prover -- WorkerPool, per-proof assembly of
// w, sponge absorbs, challenge derivation, unpacking of opened columns -- over the race-detector stub of the device ABI
// (stub_ligero_hip.cpp).  Exit status 0 and no ThreadSanitizer report = pass; the proofs themselves are meaningless here.
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../ligero_amd/host/prover.hpp"

using namespace ligero;

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: tsan_host_phases <poseidon.r1cs> <witness_batch64.bin>\n"); return 2; }
    // 1. the pool alone: many short generations, an exception in one of them, reuse afterwards
    {
        WorkerPool pool(6);
        std::atomic<uint64_t> sum{0};
        for (int round = 0; round < 200; round++) pool.run(37, [&](size_t i) { sum += i; });
        if (sum != 200ull * (36 * 37 / 2)) { fprintf(stderr, "pool sum wrong\n"); return 1; }
        bool threw = false;
        try {
            pool.run(64, [&](size_t i) { if (i == 13) throw std::runtime_error("boom"); sum += 1; });
        } catch (const std::runtime_error&) { threw = true; }
        if (!threw) { fprintf(stderr, "exception lost\n"); return 1; }
        pool.run(8, [&](size_t) { sum += 1; });
    }
    // 2. the batch prover's host phases on the Poseidon instance (8 proofs, 4 host threads), twice over the same storage
    const R1cs r1cs = read_r1cs(argv[1]);
    auto compiled = ArithmeticCircuit::from_constraint_system(r1cs);
    LigeroInstance inst(std::move(compiled.first), compiled.second, 128);
    FILE* f = fopen(argv[2], "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
    const uint32_t B = 8;
    std::vector<std::vector<std::pair<size_t, Fr>>> assignments(B);
    for (uint32_t b = 0; b < B; b++) {
        for (size_t j = 0; j < 265; j++) {
            uint8_t raw[32];
            if (fread(raw, 1, 32, f) != 32) { fprintf(stderr, "short witness file\n"); return 2; }
            if (j == 0) continue;
            Fr v;
            for (int l = 0; l < 4; l++) { v.l[l] = 0; for (int i = 0; i < 8; i++) v.l[l] |= (uint64_t)raw[8 * l + i] << (8 * i); }
            assignments[b].emplace_back(j, lg_host::to_mont(v));
        }
    }
    fclose(f);
    HipLigeroBatch prover(inst, B, 0, 4);
    const std::vector<LigeroProof>& first = prover.prove(assignments);
    std::vector<Digest> roots;
    std::vector<size_t> lens;
    for (const auto& p : first) { roots.push_back(p.u_root); lens.push_back(p.linear_constraints_proof.polynomial.size() + p.quadratic_constraints_proof.open.columns.size()); }
    const std::vector<LigeroProof>& second = prover.prove(assignments);
    for (uint32_t b = 0; b < B; b++)
        if (second[b].u_root != roots[b] || second[b].linear_constraints_proof.polynomial.size() + second[b].quadratic_constraints_proof.open.columns.size() != lens[b]) {
            fprintf(stderr, "proof %u differs between two runs over the stub\n", b);
            return 1;
        }
    printf("tsan harness done: %u proofs x 2, %u host threads\n", B, prover.threads());
    return 0;
}
