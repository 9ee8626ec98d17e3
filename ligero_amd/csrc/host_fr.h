// Host-side BN254 Fr (4 x u64 Montgomery) used only to build the domain tables the kernels
// read (twiddles for small_domain / large_domain of src/ligero/mod.rs:204-211).  Product
// code: deliberately independent of oracle/.
#pragma once
#include <stdint.h>

#include <cstdlib>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

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
// a - p if a >= p, without a branch (on field data the outcome is a coin toss: a mispredicted branch costs more than the subtraction),
// and a + b mod p for a, b < p (the sum is below 2p < 2^255: no carry leaves the top limb).  With the carry intrinsics where the host
// has them: the 128-bit-integer loops below compile to 19 ns per addition in a chain, the intrinsics to 4.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
inline Fr cond_sub_p(const Fr& a) {
    unsigned long long d0, d1, d2, d3;
    unsigned char br = _subborrow_u64(0, a.l[0], kP.l[0], &d0);
    br = _subborrow_u64(br, a.l[1], kP.l[1], &d1);
    br = _subborrow_u64(br, a.l[2], kP.l[2], &d2);
    br = _subborrow_u64(br, a.l[3], kP.l[3], &d3);
    Fr r;
    r.l[0] = br ? a.l[0] : d0; r.l[1] = br ? a.l[1] : d1; r.l[2] = br ? a.l[2] : d2; r.l[3] = br ? a.l[3] : d3;
    return r;
}
inline Fr add_mod(const Fr& a, const Fr& b) {
    unsigned long long s0, s1, s2, s3;
    unsigned char c = _addcarry_u64(0, a.l[0], b.l[0], &s0);
    c = _addcarry_u64(c, a.l[1], b.l[1], &s1);
    c = _addcarry_u64(c, a.l[2], b.l[2], &s2);
    (void)_addcarry_u64(c, a.l[3], b.l[3], &s3);
    return cond_sub_p(Fr{{s0, s1, s2, s3}});
}
#else
inline Fr cond_sub_p(const Fr& a) {
    Fr d;
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        const u128 x = (u128)a.l[i] - kP.l[i] - borrow;
        d.l[i] = (uint64_t)x;
        borrow = (uint64_t)(x >> 64) & 1;
    }
    const uint64_t keep = 0 - borrow;              // all ones: a < p
    Fr r;
    for (int i = 0; i < 4; i++) r.l[i] = (a.l[i] & keep) | (d.l[i] & ~keep);
    return r;
}
inline Fr add_mod(const Fr& a, const Fr& b) {
    Fr s;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.l[i] + b.l[i];
        s.l[i] = (uint64_t)c;
        c >>= 64;
    }
    return cond_sub_p(s);
}
#endif
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define LG_HOST_HAVE_ADX_PATH 1
// The same product with mulx / adcx / adox: two carry chains that do not wait for each other (the compiler's add-with-carry code has
// one flags register to thread everything through).  The Fiat-Shamir transcript of a large proof is ONE chain of dependent products
// -- 10 240 Poseidon permutations of 275 products each at 2^20 constraints, 60 % of the proof's time once everything else is on
// the device -- so the LATENCY of a product is what counts: 44 -> 33 ns per dependent product with the final subtraction, 23 ns
// without it (Xeon 2.1 GHz; tools/host_mul_bench.cpp).  No final subtraction: for inputs below 2p the result is below 2p (4p < 2^256).
__attribute__((always_inline)) inline void mont_mul_adx(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t t0, t1, t2, t3, A, lo, m, m2;
    const uint64_t a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    const uint64_t p0 = kP.l[0], p1 = kP.l[1], p2 = kP.l[2], p3 = kP.l[3];
    uint64_t bi = b[0];
    __asm__(
        "xorq %[lo], %[lo]\n\t"
        "mulxq %[a0], %[t0], %[t1]\n\t"
        "mulxq %[a1], %[lo], %[t2]\n\t" "adoxq %[lo], %[t1]\n\t"
        "mulxq %[a2], %[lo], %[t3]\n\t" "adoxq %[lo], %[t2]\n\t"
        "mulxq %[a3], %[lo], %[A]\n\t"  "adoxq %[lo], %[t3]\n\t"
        "movl $0, %k[lo]\n\t" "adoxq %[lo], %[A]\n\t"
        : [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3), [A] "=&r"(A), [lo] "=&r"(lo)
        : "d"(bi), [a0] "r"(a0), [a1] "r"(a1), [a2] "r"(a2), [a3] "r"(a3)
        : "cc");
#define LG_ADX_REDUCE                                                                                         \
    m = t0 * kInv64;                                                                                          \
    __asm__(                                                                                                  \
        "xorq %[lo], %[lo]\n\t"                                                                               \
        "mulxq %[p0], %[lo], %[m2]\n\t" "adcxq %[t0], %[lo]\n\t" "movq %[m2], %[t0]\n\t"                     \
        "adcxq %[t1], %[t0]\n\t" "mulxq %[p1], %[lo], %[t1]\n\t" "adoxq %[lo], %[t0]\n\t"                    \
        "adcxq %[t2], %[t1]\n\t" "mulxq %[p2], %[lo], %[t2]\n\t" "adoxq %[lo], %[t1]\n\t"                    \
        "adcxq %[t3], %[t2]\n\t" "mulxq %[p3], %[lo], %[t3]\n\t" "adoxq %[lo], %[t2]\n\t"                    \
        "movl $0, %k[lo]\n\t" "adcxq %[lo], %[t3]\n\t" "adoxq %[A], %[t3]\n\t"                               \
        : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [lo] "=&r"(lo), [m2] "=&r"(m2)     \
        : "d"(m), [A] "r"(A), [p0] "r"(p0), [p1] "r"(p1), [p2] "r"(p2), [p3] "r"(p3)                          \
        : "cc");
    LG_ADX_REDUCE
    for (int i = 1; i < 4; i++) {
        bi = b[i];
        __asm__(
            "xorq %[lo], %[lo]\n\t"
            "mulxq %[a0], %[lo], %[A]\n\t" "adoxq %[lo], %[t0]\n\t"
            "adcxq %[A], %[t1]\n\t" "mulxq %[a1], %[lo], %[A]\n\t" "adoxq %[lo], %[t1]\n\t"
            "adcxq %[A], %[t2]\n\t" "mulxq %[a2], %[lo], %[A]\n\t" "adoxq %[lo], %[t2]\n\t"
            "adcxq %[A], %[t3]\n\t" "mulxq %[a3], %[lo], %[A]\n\t" "adoxq %[lo], %[t3]\n\t"
            "movl $0, %k[lo]\n\t" "adcxq %[lo], %[A]\n\t" "adoxq %[lo], %[A]\n\t"
            : [t0] "+&r"(t0), [t1] "+&r"(t1), [t2] "+&r"(t2), [t3] "+&r"(t3), [A] "=&r"(A), [lo] "=&r"(lo)
            : "d"(bi), [a0] "r"(a0), [a1] "r"(a1), [a2] "r"(a2), [a3] "r"(a3)
            : "cc");
        LG_ADX_REDUCE
    }
#undef LG_ADX_REDUCE
    r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
// BMI2 + ADX on this CPU, and not switched off (LG_HOST_NO_ADX=1: the portable product everywhere, for A/B and tests)
inline bool detect_adx() {
    const char* e = std::getenv("LG_HOST_NO_ADX");
    if (e && std::atoi(e) != 0) return false;
    __builtin_cpu_init();       // (this runs among a shared library's dynamic initialisers)
    return __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
}
inline const bool g_have_adx = detect_adx();       // (a plain load per use: a function-local static's guard cost 10 ns per product in a chain)
inline bool have_adx() { return g_have_adx; }
#else
inline bool have_adx() { return false; }
#endif

// Montgomery product a*b/R mod p (CIOS, unrolled: half the latency of the looped form, and the prover's
// transcript spends most of its host time here) -- the portable form
inline Fr mul_portable(const Fr& a, const Fr& b) {
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
inline Fr mul(const Fr& a, const Fr& b) {
#ifdef LG_HOST_HAVE_ADX_PATH
    if (have_adx()) {
        Fr r;
        mont_mul_adx(r.l, a.l, b.l);
        return cond_sub_p(r);
    }
#endif
    return mul_portable(a, b);
}
// the product WITHOUT the final subtraction where the fast path exists: inputs below 2p give a result below 2p -- for chains of
// products (an S-box) that reduce once at the end (reduce_lazy); elsewhere it is the ordinary product
inline Fr mul_lazy(const Fr& a, const Fr& b) {
#ifdef LG_HOST_HAVE_ADX_PATH
    if (have_adx()) {
        Fr r;
        mont_mul_adx(r.l, a.l, b.l);
        return r;
    }
#endif
    return mul_portable(a, b);
}
#ifdef LG_HOST_HAVE_ADX_PATH
__attribute__((always_inline)) inline Fr mul_lazy_adx(const Fr& a, const Fr& b) {     // have_adx() checked by the caller, once
    Fr r;
    mont_mul_adx(r.l, a.l, b.l);
    return r;
}
#endif
inline Fr reduce_lazy(const Fr& a) { return cond_sub_p(a); }     // [0, 2p) -> [0, p)
#ifdef LG_HOST_HAVE_ADX_PATH
// x^17 (four squarings and a product, every one without its final subtraction) as ONE block of instructions: the running value never
// leaves its four registers between the products -- as five separate blocks the compiler moves it through the stack in between and
// reloads the modulus (15 registers for 16 live values), on the critical path of a chain that is nothing but these.  Input below 2p,
// output below 2p (reduce_lazy afterwards).
#define LG_SBOX_REDUCE                                                                                   \
    "movq %[t0], %%rdx\n\t" "imulq %[inv], %%rdx\n\t"                                                    \
    "xorq %[lo], %[lo]\n\t"                                                                              \
    "mulxq %[p0], %[lo], %[m2]\n\t" "adcxq %[t0], %[lo]\n\t" "movq %[m2], %[t0]\n\t"                    \
    "adcxq %[t1], %[t0]\n\t" "mulxq %[p1], %[lo], %[t1]\n\t" "adoxq %[lo], %[t0]\n\t"                   \
    "adcxq %[t2], %[t1]\n\t" "mulxq %[p2], %[lo], %[t2]\n\t" "adoxq %[lo], %[t1]\n\t"                   \
    "adcxq %[t3], %[t2]\n\t" "mulxq %[p3], %[lo], %[t3]\n\t" "adoxq %[lo], %[t2]\n\t"                   \
    "movl $0, %k[lo]\n\t" "adcxq %[lo], %[t3]\n\t" "adoxq %[A], %[t3]\n\t"
#define LG_SBOX_ROUND0(B)                                                                                \
    "movq " B ", %%rdx\n\t" "xorq %[lo], %[lo]\n\t"                                                      \
    "mulxq %[y0], %[t0], %[t1]\n\t"                                                                      \
    "mulxq %[y1], %[lo], %[t2]\n\t" "adoxq %[lo], %[t1]\n\t"                                             \
    "mulxq %[y2], %[lo], %[t3]\n\t" "adoxq %[lo], %[t2]\n\t"                                             \
    "mulxq %[y3], %[lo], %[A]\n\t"  "adoxq %[lo], %[t3]\n\t"                                             \
    "movl $0, %k[lo]\n\t" "adoxq %[lo], %[A]\n\t" LG_SBOX_REDUCE
#define LG_SBOX_ROUND(B)                                                                                 \
    "movq " B ", %%rdx\n\t" "xorq %[lo], %[lo]\n\t"                                                      \
    "mulxq %[y0], %[lo], %[A]\n\t" "adoxq %[lo], %[t0]\n\t"                                              \
    "adcxq %[A], %[t1]\n\t" "mulxq %[y1], %[lo], %[A]\n\t" "adoxq %[lo], %[t1]\n\t"                     \
    "adcxq %[A], %[t2]\n\t" "mulxq %[y2], %[lo], %[A]\n\t" "adoxq %[lo], %[t2]\n\t"                     \
    "adcxq %[A], %[t3]\n\t" "mulxq %[y3], %[lo], %[A]\n\t" "adoxq %[lo], %[t3]\n\t"                     \
    "movl $0, %k[lo]\n\t" "adcxq %[lo], %[A]\n\t" "adoxq %[lo], %[A]\n\t" LG_SBOX_REDUCE
#define LG_SBOX_SQUARE                                                                                   \
    LG_SBOX_ROUND0("%[y0]") LG_SBOX_ROUND("%[y1]") LG_SBOX_ROUND("%[y2]") LG_SBOX_ROUND("%[y3]")         \
    "movq %[t0], %[y0]\n\t" "movq %[t1], %[y1]\n\t" "movq %[t2], %[y2]\n\t" "movq %[t3], %[y3]\n\t"
__attribute__((always_inline)) inline Fr sbox17_lazy_adx(const Fr& x) {
    uint64_t y0 = x.l[0], y1 = x.l[1], y2 = x.l[2], y3 = x.l[3];
    uint64_t t0, t1, t2, t3, A, lo, m2;
    __asm__(
        LG_SBOX_SQUARE LG_SBOX_SQUARE LG_SBOX_SQUARE LG_SBOX_SQUARE
        LG_SBOX_ROUND0("%[x0]") LG_SBOX_ROUND("%[x1]") LG_SBOX_ROUND("%[x2]") LG_SBOX_ROUND("%[x3]")
        : [y0] "+&r"(y0), [y1] "+&r"(y1), [y2] "+&r"(y2), [y3] "+&r"(y3), [t0] "=&r"(t0), [t1] "=&r"(t1), [t2] "=&r"(t2), [t3] "=&r"(t3),
          [A] "=&r"(A), [lo] "=&r"(lo), [m2] "=&r"(m2)
        : [x0] "m"(x.l[0]), [x1] "m"(x.l[1]), [x2] "m"(x.l[2]), [x3] "m"(x.l[3]), [p0] "m"(kP.l[0]), [p1] "m"(kP.l[1]), [p2] "m"(kP.l[2]),
          [p3] "m"(kP.l[3]), [inv] "m"(kInv64)
        : "cc", "rdx");
    return Fr{{t0, t1, t2, t3}};
}
#undef LG_SBOX_SQUARE
#undef LG_SBOX_ROUND
#undef LG_SBOX_ROUND0
#undef LG_SBOX_REDUCE
#endif
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
