// Latency of ONE dependent Montgomery product on the host (the Fiat-Shamir chain of a large proof is a chain of these): the portable
// CIOS, the mulx / adcx / adox form with its final subtraction, and without it; checked against each other on random operands first.
//   g++ -O2 -std=c++17 -I ligero_amd/csrc -o /tmp/host_mul_bench tools/host_mul_bench.cpp && /tmp/host_mul_bench
#include <chrono>
#include <cstdio>
#include <random>

#include "host_fr.h"
using namespace lg_host;

int main() {
    std::mt19937_64 g(1);
    long bad = 0;
    for (int i = 0; i < 2000000; i++) {
        Fr a{{g(), g(), g(), g() >> 2}}, b{{g(), g(), g(), g() >> 2}};
        a = reduce_lazy(a); b = reduce_lazy(b);
        if (geq(a, kP) || geq(b, kP)) continue;
        const Fr w = mul_portable(a, b), x = mul(a, b), y = reduce_lazy(mul_lazy(a, b));
        for (int l = 0; l < 4; l++) bad += (w.l[l] != x.l[l]) + (w.l[l] != y.l[l]);
    }
    printf("fast path %s; mismatching limbs on 2 M random products: %ld\n", have_adx() ? "on (bmi2 + adx)" : "off", bad);
    const Fr x = to_mont(Fr{{123456789, 987654321, 5, 7}});
    const int N = 20000000;
    for (int variant = 0; variant < 4; variant++) {
        Fr y = x;
        const auto t0 = std::chrono::steady_clock::now();
        if (variant == 0) for (int i = 0; i < N; i++) y = mul_portable(y, y);
        if (variant == 1) for (int i = 0; i < N; i++) y = mul(y, y);
        if (variant == 2) { for (int i = 0; i < N; i++) y = mul_lazy(y, y); y = reduce_lazy(y); }
#ifdef LG_HOST_HAVE_ADX_PATH
        if (variant == 3) { if (!have_adx()) break; for (int i = 0; i < N; i++) y = mul_lazy_adx(y, y); y = reduce_lazy(y); }
#else
        if (variant == 3) break;
#endif
        const double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("%-62s %6.2f ns per dependent squaring   (%016llx)\n",
               variant == 0 ? "portable CIOS" : variant == 1 ? "mul (fast path + final subtraction)" : variant == 2 ? "mul_lazy chain (dispatch per product)" : "mul_lazy_adx chain (dispatch hoisted: the sponge's S-box)", ns,
               (unsigned long long)(y.l[0] ^ y.l[3]));
    }
    return bad != 0;
}
