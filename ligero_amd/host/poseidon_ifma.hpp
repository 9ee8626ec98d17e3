// Eight Poseidon sponges in lock-step on AVX-512 IFMA (x86-64 hosts that have it; everything else uses the scalar code of
// transcript.hpp).  What bounds Poseidon-shape proofs per second is the host's transcript: a proof absorbs k + 2k + 2k field
// elements at rate 2, a permutation (alpha = 17, 8 + 31 rounds) is 275 field products, and a scalar 4 x 64-bit Montgomery product
// costs about 80 cycles -- 2.2 ms of sponge per proof and core.  The proofs of a batch run the same
// absorb schedule on independent states, so eight of them fit the eight 64-bit lanes of a vector: elements in radix 2^52 (five
// limbs, Montgomery constant 2^260), products with vpmadd52luq / vpmadd52huq, every value kept fully reduced so that the lanes
// hold exactly the field elements the scalar sponge holds.  Results are bit-identical to the scalar sponge
// (tests/test_transcript.py, and tests/test_gpu_prover.py::test_batch_prover_matches_single_prover end to end).
//
// Only for ark_bn254::Fr with test_sponge()'s shape (width 3, alpha 17, the additions-only MDS [[1,0,1],[1,1,0],[0,1,1]]).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../csrc/host_fr.h"

#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define LG_HAVE_IFMA_BUILD 1
#else
#define LG_HAVE_IFMA_BUILD 0
#endif

namespace ligero {
namespace ifma {

using lg_host::Fr;

inline bool available() {
#if LG_HAVE_IFMA_BUILD
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && getenv("LG_NO_IFMA") == nullptr;
    return ok;
#else
    return false;
#endif
}

#if LG_HAVE_IFMA_BUILD
#define LG_IFMA __attribute__((target("avx512f,avx512ifma"), always_inline)) inline
#define LG_IFMA_FN __attribute__((target("avx512f,avx512ifma")))
#define LG_UNROLL _Pragma("GCC unroll 8")   // (the library is built with -O2: the limb loops must not stay loops)

struct V5 {
    __m512i l[5];   // limb i of eight field elements, radix 2^52, each limb < 2^52, value < p
};

constexpr uint64_t kMask52 = (1ull << 52) - 1;

// 4 x 64-bit little-endian limbs -> 5 x 52-bit
inline void to52(const Fr& x, uint64_t out[5]) {
    out[0] = x.l[0] & kMask52;
    out[1] = ((x.l[0] >> 52) | (x.l[1] << 12)) & kMask52;
    out[2] = ((x.l[1] >> 40) | (x.l[2] << 24)) & kMask52;
    out[3] = ((x.l[2] >> 28) | (x.l[3] << 36)) & kMask52;
    out[4] = x.l[3] >> 16;
}
inline Fr from52(const uint64_t in[5]) {
    Fr x;
    x.l[0] = in[0] | (in[1] << 52);
    x.l[1] = (in[1] >> 12) | (in[2] << 40);
    x.l[2] = (in[2] >> 24) | (in[3] << 28);
    x.l[3] = (in[3] >> 36) | (in[4] << 16);
    return x;
}

struct Consts {
    uint64_t p[5];        // the modulus
    uint64_t pinv;        // -p^-1 mod 2^52
    uint64_t enter[5];    // 2^264 mod p: mont(y, enter) = 16 y      (residue of the ABI's 2^256 form -> this file's 2^260 form)
    uint64_t leave[5];    // 2^256 mod p: mont(x, leave) = x / 16
};
inline const Consts& consts() {
    static const Consts c = [] {
        Consts k;
        to52(lg_host::kP, k.p);
        k.pinv = lg_host::kInv64 & kMask52;
        // as plain integers: lg_host::kOneMont = 2^256 mod p; 2^264 mod p = (2^256 mod p) * 2^8 mod p = mul(kOneMont, to_mont(256)) as integers
        const Fr two8 = lg_host::to_mont(Fr{{256, 0, 0, 0}});
        to52(lg_host::mul(lg_host::kOneMont, two8), k.enter);
        to52(lg_host::kOneMont, k.leave);
        return k;
    }();
    return c;
}

LG_IFMA __m512i bcast(uint64_t v) { return _mm512_set1_epi64((long long)v); }

// r = a or a - p, whichever lies in [0, p), for a < 2p with normalised limbs
LG_IFMA V5 cond_sub_p(const V5& a, const __m512i p[5]) {
    const __m512i mask = bcast(kMask52);
    V5 d;
    __m512i borrow = _mm512_setzero_si512();
    LG_UNROLL
    for (int i = 0; i < 5; i++) {
        const __m512i t = _mm512_sub_epi64(_mm512_sub_epi64(a.l[i], p[i]), borrow);
        borrow = _mm512_srli_epi64(t, 63);
        d.l[i] = _mm512_and_si512(t, mask);
    }
    const __mmask8 keep = _mm512_test_epi64_mask(borrow, borrow);   // borrow out = 1: a < p, keep a
    V5 r;
    LG_UNROLL
    for (int i = 0; i < 5; i++) r.l[i] = _mm512_mask_blend_epi64(keep, d.l[i], a.l[i]);
    return r;
}
LG_IFMA void normalise(__m512i acc[5]) {
    const __m512i mask = bcast(kMask52);
    LG_UNROLL
    for (int i = 0; i < 4; i++) {
        acc[i + 1] = _mm512_add_epi64(acc[i + 1], _mm512_srli_epi64(acc[i], 52));
        acc[i] = _mm512_and_si512(acc[i], mask);
    }
}
// a b 2^-260 mod p, fully reduced
LG_IFMA V5 mont_mul(const V5& a, const V5& b, const __m512i p[5], __m512i pinv) {
    const __m512i zero = _mm512_setzero_si512();
    __m512i acc[6] = {zero, zero, zero, zero, zero, zero};
    LG_UNROLL
    for (int i = 0; i < 5; i++) {
        const __m512i bi = b.l[i];
        LG_UNROLL
        for (int j = 0; j < 5; j++) acc[j] = _mm512_madd52lo_epu64(acc[j], a.l[j], bi);
        LG_UNROLL
        for (int j = 0; j < 5; j++) acc[j + 1] = _mm512_madd52hi_epu64(acc[j + 1], a.l[j], bi);
        const __m512i m = _mm512_madd52lo_epu64(zero, acc[0], pinv);
        LG_UNROLL
        for (int j = 0; j < 5; j++) acc[j] = _mm512_madd52lo_epu64(acc[j], m, p[j]);
        LG_UNROLL
        for (int j = 0; j < 5; j++) acc[j + 1] = _mm512_madd52hi_epu64(acc[j + 1], m, p[j]);
        const __m512i carry = _mm512_srli_epi64(acc[0], 52);          // the low 52 bits are zero now
        acc[0] = _mm512_add_epi64(acc[1], carry);
        acc[1] = acc[2]; acc[2] = acc[3]; acc[3] = acc[4]; acc[4] = acc[5]; acc[5] = zero;
    }
    normalise(acc);
    V5 r;
    LG_UNROLL
    for (int i = 0; i < 5; i++) r.l[i] = acc[i];
    return cond_sub_p(r, p);
}
LG_IFMA V5 mod_add(const V5& a, const V5& b, const __m512i p[5]) {
    __m512i s[5];
    LG_UNROLL
    for (int i = 0; i < 5; i++) s[i] = _mm512_add_epi64(a.l[i], b.l[i]);
    normalise(s);
    V5 r;
    LG_UNROLL
    for (int i = 0; i < 5; i++) r.l[i] = s[i];
    return cond_sub_p(r, p);
}

// Round constants of one sponge shape in this file's representation, and the eight-lane permutation / absorption
class Engine {
public:
    Engine(const std::vector<std::array<Fr, 3>>& ark_mont256, size_t full_rounds, size_t partial_rounds) : full_(full_rounds), partial_(partial_rounds) {
        const Fr sixteen = lg_host::to_mont(Fr{{16, 0, 0, 0}});
        ark_.resize(ark_mont256.size());
        for (size_t r = 0; r < ark_mont256.size(); r++)
            for (int w = 0; w < 3; w++) to52(lg_host::mul(ark_mont256[r][w], sixteen), ark_[r][w].data());   // residue 16 y mod p = x 2^260
    }

    // absorb_internal of transcript.hpp on eight sponges at once: `len` elements each (elems[j] = proof j's), starting at
    // position `start` of the rate (0 or 1); states[j] = that sponge's three state elements (ABI form, updated in place).
    // permute_first: the permutation absorb_elements runs before absorbing when the sponge was squeezing or its rate is full.
    // Returns the position after the last absorbed element (1 or 2), as absorb_internal leaves next_index_.
    LG_IFMA_FN size_t absorb8(Fr* const states[8], const Fr* const elems[8], size_t len, size_t start, bool permute_first) const {
        const Consts& k = consts();
        __m512i p[5];
        LG_UNROLL
        for (int i = 0; i < 5; i++) p[i] = bcast(k.p[i]);
        const __m512i pinv = bcast(k.pinv);
        V5 enter, leave;
        LG_UNROLL
        for (int i = 0; i < 5; i++) { enter.l[i] = bcast(k.enter[i]); leave.l[i] = bcast(k.leave[i]); }
        V5 st[3];
        LG_UNROLL
        for (int w = 0; w < 3; w++) st[w] = mont_mul(load8(states, w), enter, p, pinv);
        if (permute_first) permute(st, p, pinv);
        size_t pos = 0;
        for (;;) {
            const size_t left = len - pos;
            const size_t take = (start + left <= 2) ? left : 2 - start;
            for (size_t i = 0; i < take; i++) {
                const V5 e = mont_mul(load8(elems, pos + i), enter, p, pinv);
                st[1 + start + i] = mod_add(st[1 + start + i], e, p);          // state_[kCapacity + start + i]
            }
            if (start + left <= 2) { start += left; break; }
            permute(st, p, pinv);
            pos += take;
            start = 0;
        }
        LG_UNROLL
        for (int w = 0; w < 3; w++) store8(states, w, mont_mul(st[w], leave, p, pinv));
        return start;
    }
    LG_IFMA_FN void permute8(Fr* const states[8]) const {
        const Consts& k = consts();
        __m512i p[5];
        LG_UNROLL
        for (int i = 0; i < 5; i++) p[i] = bcast(k.p[i]);
        const __m512i pinv = bcast(k.pinv);
        V5 enter, leave;
        LG_UNROLL
        for (int i = 0; i < 5; i++) { enter.l[i] = bcast(k.enter[i]); leave.l[i] = bcast(k.leave[i]); }
        V5 st[3];
        LG_UNROLL
        for (int w = 0; w < 3; w++) st[w] = mont_mul(load8(states, w), enter, p, pinv);
        permute(st, p, pinv);
        LG_UNROLL
        for (int w = 0; w < 3; w++) store8(states, w, mont_mul(st[w], leave, p, pinv));
    }

private:
    size_t full_, partial_;
    std::vector<std::array<std::array<uint64_t, 5>, 3>> ark_;

    LG_IFMA static V5 load8(const Fr* const ptr[8], size_t index) {
        alignas(64) uint64_t limbs[5][8];
        for (int j = 0; j < 8; j++) {
            uint64_t t[5];
            to52(ptr[j][index], t);
            LG_UNROLL
            for (int i = 0; i < 5; i++) limbs[i][j] = t[i];
        }
        V5 v;
        LG_UNROLL
        for (int i = 0; i < 5; i++) v.l[i] = _mm512_load_si512(limbs[i]);
        return v;
    }
    LG_IFMA static V5 load8(Fr* const ptr[8], size_t index) { return load8(const_cast<const Fr* const*>(ptr), index); }
    LG_IFMA static void store8(Fr* const ptr[8], size_t index, const V5& v) {
        alignas(64) uint64_t limbs[5][8];
        LG_UNROLL
        for (int i = 0; i < 5; i++) _mm512_store_si512(limbs[i], v.l[i]);
        for (int j = 0; j < 8; j++) {
            const uint64_t t[5] = {limbs[0][j], limbs[1][j], limbs[2][j], limbs[3][j], limbs[4][j]};
            ptr[j][index] = from52(t);
        }
    }
    LG_IFMA V5 sbox17(const V5& x, const __m512i p[5], __m512i pinv) const {
        V5 y = mont_mul(x, x, p, pinv);
        y = mont_mul(y, y, p, pinv);
        y = mont_mul(y, y, p, pinv);
        y = mont_mul(y, y, p, pinv);
        return mont_mul(y, x, p, pinv);
    }
    LG_IFMA void permute(V5 st[3], const __m512i p[5], __m512i pinv) const {
        const size_t half = full_ / 2;
        for (size_t r = 0; r < full_ + partial_; r++) {
            LG_UNROLL
            for (int w = 0; w < 3; w++) {
                V5 c;
                LG_UNROLL
                for (int i = 0; i < 5; i++) c.l[i] = bcast(ark_[r][w][i]);
                st[w] = mod_add(st[w], c, p);
            }
            const bool full = r < half || r >= half + partial_;
            st[0] = sbox17(st[0], p, pinv);
            if (full) { st[1] = sbox17(st[1], p, pinv); st[2] = sbox17(st[2], p, pinv); }
            const V5 n0 = mod_add(st[0], st[2], p), n1 = mod_add(st[0], st[1], p), n2 = mod_add(st[1], st[2], p);   // [[1,0,1],[1,1,0],[0,1,1]]
            st[0] = n0; st[1] = n1; st[2] = n2;
        }
    }
};
#endif  // LG_HAVE_IFMA_BUILD

}  // namespace ifma
}  // namespace ligero
