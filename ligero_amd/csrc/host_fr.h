// Host-side BN254 Fr (4 x u64 Montgomery) used only to build the domain tables the kernels
// read (twiddles for small_domain / large_domain of src/ligero/mod.rs:204-211).  Product
// code: deliberately independent of oracle/.
#pragma once
#include <stdint.h>

namespace lg_host {

typedef unsigned __int128 u128;
struct Fr {
    uint64_t l[4];
};

static const Fr kP = {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
static const Fr kOneMont = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};  // R mod p
static const Fr kR2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};      // R^2 mod p
static const uint64_t kInv64 = 0xc2e1f593efffffffULL;
// 5^((p-1)/2^28), canonical: ark_bn254::Fr TWO_ADIC_ROOT_OF_UNITY (TWO_ADICITY = 28)
static const Fr kTwoAdicRootCanon = {{0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL, 0x2a3c09f0a58a7e85ULL}};
static const int kTwoAdicity = 28;

inline bool geq(const Fr& a, const Fr& b) {
    for (int i = 3; i >= 0; i--) {
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    }
    return true;
}
inline Fr sub_raw(const Fr& a, const Fr& b) {
    Fr r;
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return r;
}
// Montgomery product a*b/R mod p (CIOS, unrolled: half the latency of the looped form, and the prover's
// transcript spends most of its host time here)
inline Fr mul(const Fr& a, const Fr& b) {
    const uint64_t p0 = kP.l[0], p1 = kP.l[1], p2 = kP.l[2], p3 = kP.l[3];
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#define LG_HOST_MUL_ROUND(bi)                                                 \
    {                                                                         \
        u128 c = (u128)a.l[0] * (bi) + t0; t0 = (uint64_t)c; c >>= 64;        \
        c += (u128)a.l[1] * (bi) + t1; t1 = (uint64_t)c; c >>= 64;            \
        c += (u128)a.l[2] * (bi) + t2; t2 = (uint64_t)c; c >>= 64;            \
        c += (u128)a.l[3] * (bi) + t3; t3 = (uint64_t)c; c >>= 64;            \
        t4 += (uint64_t)c;             /* p has two spare bits: no overflow */ \
        const uint64_t m = t0 * kInv64;                                       \
        c = ((u128)m * p0 + t0) >> 64;                                        \
        c += (u128)m * p1 + t1; t0 = (uint64_t)c; c >>= 64;                   \
        c += (u128)m * p2 + t2; t1 = (uint64_t)c; c >>= 64;                   \
        c += (u128)m * p3 + t3; t2 = (uint64_t)c; c >>= 64;                   \
        c += t4; t3 = (uint64_t)c; t4 = (uint64_t)(c >> 64);                  \
    }
    LG_HOST_MUL_ROUND(b.l[0]) LG_HOST_MUL_ROUND(b.l[1]) LG_HOST_MUL_ROUND(b.l[2]) LG_HOST_MUL_ROUND(b.l[3])
#undef LG_HOST_MUL_ROUND
    Fr r = {{t0, t1, t2, t3}};
    if (t4 || geq(r, kP)) r = sub_raw(r, kP);
    return r;
}
inline Fr to_mont(const Fr& a) { return mul(a, kR2); }
inline Fr from_mont(const Fr& a) {
    Fr one = {{1, 0, 0, 0}};
    return mul(a, one);
}
inline Fr pow_u64(Fr base, uint64_t e) {
    Fr acc = kOneMont;
    while (e) {
        if (e & 1) acc = mul(acc, base);
        base = mul(base, base);
        e >>= 1;
    }
    return acc;
}
inline Fr inverse(const Fr& a) {  // a^(p-2)
    Fr e = kP;
    e.l[0] -= 2;
    Fr acc = kOneMont, b = a;
    for (int i = 0; i < 4; i++) {
        uint64_t w = e.l[i];
        for (int j = 0; j < 64; j++) {
            if (w & 1) acc = mul(acc, b);
            b = mul(b, b);
            w >>= 1;
        }
    }
    return acc;
}
// generator of the order-`size` subgroup (size = 2^log_size), Montgomery form:
// GeneralEvaluationDomain::new(size).group_gen
inline Fr domain_generator(int log_size) { return pow_u64(to_mont(kTwoAdicRootCanon), 1ULL << (kTwoAdicity - log_size)); }

}  // namespace lg_host
