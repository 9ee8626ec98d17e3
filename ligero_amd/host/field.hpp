// Element types of the host-side mirror.  The reference is generic over `F: PrimeField + Absorb` (src/ligero/mod.rs:146) and
// its tests instantiate ark_bn254::Fr (every circom fixture) and ark_bls12_377::Fq (src/ligero/tests.rs:23, 186-193); the
// host classes (circuit.hpp, transcript.hpp, prover.hpp) are templates over the element type E and find everything
// field-specific in Field<E>:
//   arithmetic on Montgomery-form limbs (the in-memory ark_ff::Fp, what crosses the C ABI), MODULUS_BIT_SIZE (the sponge's
//   byte packing and F::rand's bit shaving depend on it), TWO_ADICITY / two-adic root (GeneralEvaluationDomain), and the
//   lg_field id of the device context.
// BN254 Fr keeps its hand-unrolled product (csrc/host_fr.h); BLS12-377 Fq uses a plain six-limb CIOS -- its only user is a
// ten-node circuit.
#pragma once
#include <cstdint>
#include <cstring>

#include "../../include/ligero_hip.h"
#include "../csrc/host_fr.h"

namespace ligero {

using lg_host::Fr;

// ark_bls12_377::Fq: 377-bit modulus, 6 x u64 limbs, R = 2^384
struct Fq377 {
    uint64_t l[6];
};

namespace fq377_detail {
typedef unsigned __int128 u128;
static const Fq377 kP = {{0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL, 0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x01ae3a4617c510eaULL}};
static const uint64_t kInv64 = 0x8508bfffffffffffULL;   // -p^-1 mod 2^64
// R mod p, R^2 mod p and the canonical 2^46-th root 15^((p-1)/2^46): recomputed by tests/test_oracle.py::test_generic_field_constants
static const Fq377 kR1 = {{0x02cdffffffffff68ULL, 0x51409f837fffffb1ULL, 0x9f7db3a98a7d3ff2ULL, 0x7b4e97b76e7c6305ULL, 0x4cf495bf803c84e8ULL, 0x008d6661e2fdf49aULL}};
static const Fq377 kR2 = {{0xb786686c9400cd22ULL, 0x0329fcaab00431b1ULL, 0x22a5f11162d6b46dULL, 0xbfdf7d03827dc3acULL, 0x837e92f041790bf9ULL, 0x006dfccb1e914b88ULL}};
static const Fq377 kRootCanon = {{0x7eca603cc563b9a1ULL, 0x06df0a4306fe0bc3ULL, 0xb44d994a0ddff8c6ULL, 0x40fbe05b4512a3d4ULL, 0x30f152488aeffc9bULL, 0x0036a92e05198a80ULL}};
inline bool geq(const Fq377& a, const Fq377& b) {
    for (int i = 5; i >= 0; i--)
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    return true;
}
inline Fq377 sub_raw(const Fq377& a, const Fq377& b) {
    Fq377 r;
    uint64_t borrow = 0;
    for (int i = 0; i < 6; i++) {
        const u128 d = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    return r;
}
inline Fq377 mul(const Fq377& a, const Fq377& b) {   // a b R^-1 mod p (CIOS; p has seven spare bits)
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        u128 c = 0;
        for (int j = 0; j < 6; j++) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[6] = (uint64_t)c;
        t[7] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * kInv64;
        c = ((u128)m * kP.l[0] + t[0]) >> 64;
        for (int j = 1; j < 6; j++) {
            c += (u128)m * kP.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[5] = (uint64_t)c;
        t[6] = t[7] + (uint64_t)(c >> 64);
        t[7] = 0;
    }
    Fq377 r;
    memcpy(r.l, t, sizeof(r.l));
    if (t[6] || geq(r, kP)) r = sub_raw(r, kP);
    return r;
}
}  // namespace fq377_detail

template <class E>
struct Field;

template <>
struct Field<Fr> {
    using Elem = Fr;
    static constexpr int kLimbs = 4, kModulusBits = 254, kTwoAdicity = lg_host::kTwoAdicity, kLgField = LG_FIELD_BN254_FR;
    static const char* name() { return "ark_bn254::Fr"; }
    static Fr modulus() { return lg_host::kP; }
    static Fr zero() { return Fr{{0, 0, 0, 0}}; }
    static Fr one() { return lg_host::kOneMont; }
    static bool is_zero(const Fr& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
    static bool eq(const Fr& a, const Fr& b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
    static bool geq_modulus(const Fr& a) { return lg_host::geq(a, lg_host::kP); }
    static Fr neg(const Fr& a) { return is_zero(a) ? a : lg_host::sub_raw(lg_host::kP, a); }
    static Fr add(const Fr& a, const Fr& b) { return lg_host::add_mod(a, b); }     // (branch free: host_fr.h)
    static Fr sub(const Fr& a, const Fr& b) { return add(a, neg(b)); }
    static Fr mul(const Fr& a, const Fr& b) { return lg_host::mul(a, b); }
    static Fr to_mont(const Fr& canonical) { return lg_host::to_mont(canonical); }
    static Fr from_mont(const Fr& a) { return lg_host::from_mont(a); }
    static Fr from_u64(uint64_t v) { return lg_host::to_mont(Fr{{v, 0, 0, 0}}); }
    static Fr pow_u64(const Fr& b, uint64_t e) { return lg_host::pow_u64(b, e); }
    static Fr domain_generator(int log_size) { return lg_host::domain_generator(log_size); }
};

template <>
struct Field<Fq377> {
    using Elem = Fq377;
    static constexpr int kLimbs = 6, kModulusBits = 377, kTwoAdicity = 46, kLgField = LG_FIELD_BLS12_377_FQ;
    static const char* name() { return "ark_bls12_377::Fq"; }
    static Fq377 modulus() { return fq377_detail::kP; }
    static Fq377 zero() { return Fq377{{0, 0, 0, 0, 0, 0}}; }
    static Fq377 one() { return fq377_detail::kR1; }
    static bool is_zero(const Fq377& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3] | a.l[4] | a.l[5]) == 0; }
    static bool eq(const Fq377& a, const Fq377& b) { return memcmp(a.l, b.l, sizeof(a.l)) == 0; }
    static bool geq_modulus(const Fq377& a) { return fq377_detail::geq(a, fq377_detail::kP); }
    static Fq377 neg(const Fq377& a) { return is_zero(a) ? a : fq377_detail::sub_raw(fq377_detail::kP, a); }
    static Fq377 add(const Fq377& a, const Fq377& b) {
        Fq377 r;
        unsigned __int128 c = 0;
        for (int i = 0; i < 6; i++) {
            c += (unsigned __int128)a.l[i] + b.l[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
        if (c || fq377_detail::geq(r, fq377_detail::kP)) r = fq377_detail::sub_raw(r, fq377_detail::kP);
        return r;
    }
    static Fq377 sub(const Fq377& a, const Fq377& b) { return add(a, neg(b)); }
    static Fq377 mul(const Fq377& a, const Fq377& b) { return fq377_detail::mul(a, b); }
    static Fq377 to_mont(const Fq377& canonical) { return fq377_detail::mul(canonical, fq377_detail::kR2); }
    static Fq377 from_mont(const Fq377& a) { return fq377_detail::mul(a, Fq377{{1, 0, 0, 0, 0, 0}}); }
    static Fq377 from_u64(uint64_t v) { return to_mont(Fq377{{v, 0, 0, 0, 0, 0}}); }
    static Fq377 pow_u64(Fq377 b, uint64_t e) {
        Fq377 acc = one();
        while (e) {
            if (e & 1) acc = mul(acc, b);
            b = mul(b, b);
            e >>= 1;
        }
        return acc;
    }
    // GeneralEvaluationDomain::new(2^log_size).group_gen = TWO_ADIC_ROOT_OF_UNITY^(2^(46 - log_size))
    static Fq377 domain_generator(int log_size) { return pow_u64(to_mont(fq377_detail::kRootCanon), 1ULL << (kTwoAdicity - log_size)); }
};

// strict weak order on elements (keys of the constants map): by limbs, most significant first
template <class E>
struct ElemLess {
    bool operator()(const E& a, const E& b) const {
        for (int i = Field<E>::kLimbs - 1; i >= 0; i--)
            if (a.l[i] != b.l[i]) return a.l[i] < b.l[i];
        return false;
    }
};

// free-function spellings used throughout the host code, for either element type
template <class E> inline bool fr_is_zero(const E& a) { return Field<E>::is_zero(a); }
template <class E> inline bool fr_eq(const E& a, const E& b) { return Field<E>::eq(a, b); }
template <class E> inline E fr_neg(const E& a) { return Field<E>::neg(a); }
template <class E> inline E fr_add(const E& a, const E& b) { return Field<E>::add(a, b); }
template <class E> inline E fr_sub(const E& a, const E& b) { return Field<E>::sub(a, b); }
template <class E> inline E fr_mul(const E& a, const E& b) { return Field<E>::mul(a, b); }
// (BN254 spellings without an argument to deduce from)
inline Fr fr_zero() { return Field<Fr>::zero(); }
inline Fr fr_one() { return Field<Fr>::one(); }
inline Fr fr_from_u64(uint64_t v) { return Field<Fr>::from_u64(v); }
using FrLess = ElemLess<Fr>;

}  // namespace ligero
