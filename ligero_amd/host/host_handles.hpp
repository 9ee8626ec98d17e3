// Opaque handles of include/ligero_host.h, shared by the two host libraries (libligero_host.so, libligero_prover.so)
#pragma once
#include "circuit.hpp"

struct lgh_circuit {
    ligero::ArithmeticCircuit c;
    std::vector<size_t> outputs;   // set by lgh_circuit_from_r1cs
    uint32_t n_wires = 0;
};
struct lgh_instance {
    ligero::LigeroInstance inst;
    uint32_t n_wires;
    lgh_instance(ligero::ArithmeticCircuit c, std::vector<size_t> outs, size_t lambda, uint32_t wires) : inst(std::move(c), std::move(outs), lambda), n_wires(wires) {}
};
