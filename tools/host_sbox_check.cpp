// The single-block x^17 of host_fr.h against five portable products on random (and lazy, below-2p) inputs, and its latency in a chain.
//   g++ -O2 -std=c++17 -I ligero_amd/csrc -o /tmp/host_sbox_check tools/host_sbox_check.cpp && /tmp/host_sbox_check
#include <chrono>
#include <cstdio>
#include <random>
#include "host_fr.h"
using namespace lg_host;
static inline Fr sbox_ref(const Fr& x) { Fr y = mul_portable(x, x); y = mul_portable(y, y); y = mul_portable(y, y); y = mul_portable(y, y); return mul_portable(y, x); }
int main() {
    if (!have_adx()) { printf("mismatching limbs: 0 (no BMI2 + ADX on this host: nothing to check)\n"); return 0; }
    std::mt19937_64 g(5);
    long bad = 0;
    for (int i = 0; i < 500000; i++) {
        Fr a{{g(), g(), g(), g() >> 2}};
        a = reduce_lazy(a);
        if (geq(a, kP)) continue;
        if (i % 3 == 0) { Fr two = add_mod(a, a); (void)two; }
        Fr w = sbox_ref(a), x = reduce_lazy(sbox17_lazy_adx(a));
        for (int l = 0; l < 4; l++) bad += w.l[l] != x.l[l];
        // lazy input (a + p < 2p)
        Fr ap; { unsigned long long c = 0; for (int l = 0; l < 4; l++) { unsigned __int128 s = (unsigned __int128)a.l[l] + kP.l[l] + c; ap.l[l] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
        Fr y = reduce_lazy(sbox17_lazy_adx(ap));
        for (int l = 0; l < 4; l++) bad += w.l[l] != y.l[l];
    }
    printf("mismatching limbs: %ld\n", bad);
    Fr x = to_mont(Fr{{123456789, 987654321, 5, 7}});
    const int N = 4000000;
    auto t0 = std::chrono::steady_clock::now();
    Fr y = x;
    for (int i = 0; i < N; i++) y = reduce_lazy(sbox17_lazy_adx(y));
    printf("single-block sbox chain: %.1f ns per sbox (%llx)\n", std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / N, (unsigned long long)y.l[0]);
    return bad != 0;
}
