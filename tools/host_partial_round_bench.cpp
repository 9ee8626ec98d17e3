// What a partial round of the host permutation costs beyond its S-box chain: the chain alone, with the round-constant additions of the two idle
// state elements, with the whole linear layer.   g++ -O2 -std=c++17 -I ligero_amd/csrc -o /tmp/hprb tools/host_partial_round_bench.cpp && /tmp/hprb
#include <array>
#include <chrono>
#include <cstdio>
#include "host_fr.h"
using namespace lg_host;
template <int V> double run(Fr s0, Fr s1, Fr s2, const std::array<Fr,3>* ark, int N, unsigned long long* sink) {
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < N; r++)
        for (int i = 0; i < 31; i++) {
            s0 = add_mod(s0, ark[i][0]);
            if (V >= 1) { s1 = add_mod(s1, ark[i][1]); s2 = add_mod(s2, ark[i][2]); }
            s0 = reduce_lazy(sbox17_lazy_adx(s0));
            const Fr n0 = add_mod(s0, s2);
            if (V >= 2) { const Fr n1 = add_mod(s0, s1), n2 = add_mod(s1, s2); s1 = n1; s2 = n2; }
            s0 = n0;
        }
    *sink = s0.l[0] ^ s1.l[0] ^ s2.l[0];
    return std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / N / 31;
}
int main() {
    Fr s0 = to_mont(Fr{{1, 2, 3, 4}}), s1 = to_mont(Fr{{5, 6, 7, 8}}), s2 = to_mont(Fr{{9, 10, 11, 12}});
    std::array<Fr, 3> ark[31];
    for (int i = 0; i < 31; i++) for (int j = 0; j < 3; j++) ark[i][j] = to_mont(Fr{{(uint64_t)(i * 3 + j + 1), 7, 7, 7}});
    unsigned long long sink;
    for (int rep = 0; rep < 2; rep++) {
        printf("chain only %.1f | + ark adds of s1 s2 %.1f | + n1 n2 (full round) %.1f ns\n", run<0>(s0, s1, s2, ark, 100000, &sink), run<1>(s0, s1, s2, ark, 100000, &sink), run<2>(s0, s1, s2, ark, 100000, &sink));
    }
}
