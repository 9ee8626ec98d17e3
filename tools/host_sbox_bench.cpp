// Latency pieces of the host Poseidon permutation: one S-box chain, S-box behind an addition, a chain of additions, three independent S-boxes (a full round).
//   g++ -O2 -std=c++17 -I ligero_amd/csrc -o /tmp/host_sbox_bench tools/host_sbox_bench.cpp && /tmp/host_sbox_bench
#include <chrono>
#include <cstdio>
#include "host_fr.h"
using namespace lg_host;
static inline Fr sbox(const Fr& x) {
    Fr y = mul_lazy_adx(x, x); y = mul_lazy_adx(y, y); y = mul_lazy_adx(y, y); y = mul_lazy_adx(y, y);
    return reduce_lazy(mul_lazy_adx(y, x));
}
int main() {
    Fr x = to_mont(Fr{{123456789, 987654321, 5, 7}}), c = to_mont(Fr{{99, 1, 2, 3}});
    const int N = 4000000;
    auto t0 = std::chrono::steady_clock::now();
    Fr y = x;
    for (int i = 0; i < N; i++) y = sbox(y);
    printf("sbox chain: %.1f ns per sbox (%llx)\n", std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / N, (unsigned long long)y.l[0]);
    t0 = std::chrono::steady_clock::now();
    y = x;
    for (int i = 0; i < N; i++) y = sbox(add_mod(y, c));
    printf("sbox(add) chain: %.1f ns (%llx)\n", std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / N, (unsigned long long)y.l[0]);
    t0 = std::chrono::steady_clock::now();
    y = x;
    for (int i = 0; i < N * 4; i++) y = add_mod(y, c);
    printf("add_mod chain: %.1f ns (%llx)\n", std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / (N * 4), (unsigned long long)y.l[0]);
    // three independent sboxes per iteration (a full round)
    Fr a = x, b = c, d = add_mod(x, c);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) { a = sbox(a); b = sbox(b); d = sbox(d); }
    printf("3 parallel sboxes: %.1f ns per round (%llx)\n", std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / N, (unsigned long long)(a.l[0] ^ b.l[0] ^ d.l[0]));
    a = x; b = c; d = add_mod(x, c);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
        Fr p = mul_lazy_adx(a, a), q = mul_lazy_adx(b, b), r = mul_lazy_adx(d, d);
        p = mul_lazy_adx(p, p); q = mul_lazy_adx(q, q); r = mul_lazy_adx(r, r);
        p = mul_lazy_adx(p, p); q = mul_lazy_adx(q, q); r = mul_lazy_adx(r, r);
        p = mul_lazy_adx(p, p); q = mul_lazy_adx(q, q); r = mul_lazy_adx(r, r);
        a = reduce_lazy(mul_lazy_adx(p, a)); b = reduce_lazy(mul_lazy_adx(q, b)); d = reduce_lazy(mul_lazy_adx(r, d));
    }
    printf("3 sboxes, products interleaved: %.1f ns per round (%llx)\n", std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / N, (unsigned long long)(a.l[0] ^ b.l[0] ^ d.l[0]));
}
