// One absorb of 8192 elements by the host sponge (4096 permutations): what one 2k-coefficient polynomial costs the transcript of a large proof.
//   g++ -O2 -std=c++17 -I ligero_amd/host -I ligero_amd/csrc -I include -o /tmp/host_sponge_bench tools/host_sponge_bench.cpp && /tmp/host_sponge_bench
#include <chrono>
#include <cstdio>
#include <vector>
#include "transcript.hpp"
using namespace ligero;
int main() {
    std::vector<Fr> el(8192);
    for (size_t i = 0; i < el.size(); i++) el[i] = lg_host::to_mont(Fr{{i * 77 + 1, i, 3, 5}});
    double best = 1e9;
    std::array<uint8_t, 32> out{};
    for (int r = 0; r < 15; r++) {
        PoseidonSponge s = PoseidonSponge::test_sponge();
        auto t0 = std::chrono::steady_clock::now();
        s.absorb_elements(el);
        out = s.squeeze_seed();
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("absorb 8192: %.2f ms, %.3f us per permutation (%02x%02x)\n", best / 1e3, best / 4096, out[0], out[1]);
}
