// Fiat-Shamir side of the prover and verifier: everything the reference
// derives from its sponge.
//
//   ChaChaRng<ROUNDS>                 rand_chacha 0.3 ChaCha20Rng (src/utils.rs:27, 36) and, with 12 rounds,
//                                     rand 0.8 StdRng behind ark_std::test_rng() (round constants of test_sponge)
//   fr_rand                           ark-ff 0.4 `impl Distribution<Fp<P, N>> for Standard` (F::rand, src/utils.rs:28)
//   get_field_elements_from_prng      src/utils.rs:23-29
//   get_distinct_indices_from_prng    src/utils.rs:31-55 (rand 0.8 gen_range on usize)
//   PoseidonSponge                    ark-crypto-primitives 0.4 sponge::poseidon::PoseidonSponge with the parameters of
//                                     ark-poly-commit's test_sponge() (src/ligero/tests.rs:151, 399; README.md:98):
//                                     absorb(&Vec<u8>) mod.rs:560, 634; absorb(&Vec<F>) mod.rs:660, 738, 850;
//                                     squeeze_bytes(32) mod.rs:653, 719, 839, 941
//
// PARITY UNPINNED.  None of these crates is vendored under /root/reference and the reference's
// tests hold no transcript bytes, so this file restates the published algorithms from the
// crates' documented behaviour and cannot be checked against a run of the reference here.  What
// IS pinned: the ChaCha20 block function against the RFC 8439 section 2.3.2 vector
// (tests/test_transcript.py).  The prover and the verifier of this repository agree with each
// other by construction; byte equality with a Rust prover's transcript is a claim this file
// does not make.  Every device call takes its challenges as plain inputs, so a host that links
// the real arkworks sponge can ignore this file entirely.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <set>
#include <type_traits>
#include <vector>

#include "circuit.hpp"
#include "poseidon_ifma.hpp"

namespace ligero {

constexpr size_t kChachaSeedBytes = 32;  // CHACHA_SEED_BYTES, src/lib.rs

// ---- ChaCha block function (RFC 8439 layout with a 64-bit counter + 64-bit stream id, as rand_chacha)
template <int ROUNDS>
class ChaChaRng {
public:
    explicit ChaChaRng(const std::array<uint8_t, 32>& seed) {
        for (int i = 0; i < 8; i++)
            key_[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    }
    // BlockRng::next_u32 / next_u64: words are consumed in order, a u64 is (low word, high word)
    uint32_t next_u32() {
        if (index_ >= 128) refill();
        return buf_[index_++];
    }
    uint64_t next_u64() {
        const uint64_t lo = next_u32();
        const uint64_t hi = next_u32();
        return lo | (hi << 32);
    }
    // one 64-byte block for a given counter / nonce words (tests: RFC 8439 2.3.2 uses a 32-bit counter and a 96-bit
    // nonce, i.e. words 12..15 = counter, n0, n1, n2)
    static void block(const uint32_t key[8], const uint32_t w12_15[4], uint32_t out[16]) {
        uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; i++) s[4 + i] = key[i];
        for (int i = 0; i < 4; i++) s[12 + i] = w12_15[i];
        uint32_t x[16];
        memcpy(x, s, sizeof(x));
        auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
            x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        };
        for (int r = 0; r < ROUNDS / 2; r++) {
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
    }

    // eight consecutive blocks at once on 8-lane integer vectors (lane = block): what the compiler turns into
    // AVX2 code where the CPU has it (target_clones picks at load time); same bytes as eight calls of block()
    typedef uint32_t v8u __attribute__((vector_size(32)));
    // (no function multi-versioning under ThreadSanitizer: the IFUNC resolver would run instrumented code before the sanitizer's
    // runtime exists)
#if defined(__SANITIZE_THREAD__)
#define LG_CLONES_AVX2
#else
#define LG_CLONES_AVX2 __attribute__((target_clones("avx2", "default")))
#endif
    LG_CLONES_AVX2 static void blocks8(const uint32_t key[8], uint64_t counter, uint32_t out[128]) {
        v8u s[16], x[16];
        const uint32_t c4[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 4; i++) s[i] = v8u{c4[i], c4[i], c4[i], c4[i], c4[i], c4[i], c4[i], c4[i]};
        for (int i = 0; i < 8; i++) s[4 + i] = v8u{key[i], key[i], key[i], key[i], key[i], key[i], key[i], key[i]};
        for (int b = 0; b < 8; b++) {
            s[12][b] = (uint32_t)(counter + b);
            s[13][b] = (uint32_t)((counter + b) >> 32);
        }
        s[14] = v8u{0, 0, 0, 0, 0, 0, 0, 0};
        s[15] = s[14];
        for (int i = 0; i < 16; i++) x[i] = s[i];
#define LG_ROTL(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define LG_QR(a, b, c, d)                                   \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = LG_ROTL(x[d], 16);   \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = LG_ROTL(x[b], 12);   \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = LG_ROTL(x[d], 8);    \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = LG_ROTL(x[b], 7);
        for (int r = 0; r < ROUNDS / 2; r++) {
            LG_QR(0, 4, 8, 12) LG_QR(1, 5, 9, 13) LG_QR(2, 6, 10, 14) LG_QR(3, 7, 11, 15)
            LG_QR(0, 5, 10, 15) LG_QR(1, 6, 11, 12) LG_QR(2, 7, 8, 13) LG_QR(3, 4, 9, 14)
        }
#undef LG_QR
#undef LG_ROTL
        for (int i = 0; i < 16; i++) {
            const v8u v = x[i] + s[i];
            for (int b = 0; b < 8; b++) out[16 * b + i] = v[b];
        }
    }

private:
    void refill() {
        blocks8(key_, counter_, buf_);
        counter_ += 8;
        index_ = 0;
    }
    uint32_t key_[8];
    uint64_t counter_ = 0;
    uint32_t buf_[128];
    int index_ = 128;
};
using ChaCha20Rng = ChaChaRng<20>;
using StdRng = ChaChaRng<12>;  // rand 0.8

// F::rand: N u64 limbs, the bits above MODULUS_BIT_SIZE masked off (two for BN254 Fr, seven for BLS12-377 Fq), rejected while
// >= p; the limbs are the element's internal (Montgomery) representation as they are
template <class E, class Rng>
inline E field_rand(Rng& rng) {
    using F = Field<E>;
    for (;;) {
        E t;
        for (int i = 0; i < F::kLimbs; i++) t.l[i] = rng.next_u64();
        t.l[F::kLimbs - 1] &= ~0ull >> (64 * F::kLimbs - F::kModulusBits);
        if (!F::geq_modulus(t)) return t;
    }
}
template <class Rng>
inline Fr fr_rand(Rng& rng) { return field_rand<Fr>(rng); }

template <class E>
inline void fill_field_elements_from_prng(E* out, size_t n, const std::array<uint8_t, 32>& seed) {
    ChaCha20Rng rng(seed);
    for (size_t i = 0; i < n; i++) out[i] = field_rand<E>(rng);
}
template <class E = Fr>
inline std::vector<E> get_field_elements_from_prng(size_t n, const std::array<uint8_t, 32>& seed) {
    std::vector<E> out(n);
    fill_field_elements_from_prng(out.data(), n, seed);
    return out;
}

// rand 0.8 UniformInt<usize>::sample_single (gen_range(0..n)): widening multiply with a rejection zone
template <class Rng>
inline uint64_t gen_range(Rng& rng, uint64_t n) {
    const uint64_t zone = (n << __builtin_clzll(n)) - 1;
    for (;;) {
        const unsigned __int128 m = (unsigned __int128)rng.next_u64() * n;
        if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
    }
}

inline std::vector<uint64_t> get_distinct_indices_from_prng(uint64_t n, uint64_t t, const std::array<uint8_t, 32>& seed) {
    ChaCha20Rng rng(seed);
    std::set<uint64_t> selected;  // BTreeSet: iteration in ascending order
    const uint64_t to_select = std::min(t, n - t);
    while (selected.size() < to_select) selected.insert(gen_range(rng, n));
    std::vector<uint64_t> out;
    if (to_select == t) {
        out.assign(selected.begin(), selected.end());
    } else {
        for (uint64_t i = 0; i < n; i++)
            if (!selected.count(i)) out.push_back(i);
    }
    return out;
}

// ---- Poseidon duplex sponge, width 3 (rate 2 + capacity 1)
inline Fr fr_pow_u64(Fr base, uint64_t e) { return lg_host::pow_u64(base, e); }

template <class E>
class PoseidonSpongeT {
    using F = Field<E>;
    using Fr = E;   // (the body below was written for one field; within this class `Fr` is the element type)
    static Fr fr_zero() { return F::zero(); }
    static Fr fr_one() { return F::one(); }
    static Fr fr_pow_u64(const Fr& b, uint64_t e) { return F::pow_u64(b, e); }
    // bytes packed per field element by absorb(&Vec<u8>) and taken per element by squeeze_bytes: (MODULUS_BIT_SIZE - 1) / 8
    static constexpr size_t kUsableBytes = (F::kModulusBits - 1) / 8;

public:
    // test_sponge(): 8 full + 31 partial rounds, alpha = 17, mds [[1,0,1],[1,1,0],[0,1,1]], round constants
    // 39 x 3 draws of F::rand from ark_std::test_rng() (StdRng seeded with the bytes below)
    static PoseidonSpongeT test_sponge() {
        PoseidonSpongeT s;
        s.full_rounds_ = 8;
        s.partial_rounds_ = 31;
        s.alpha_ = 17;
        const Fr one = fr_one(), zero = fr_zero();
        s.mds_ = {{{{one, zero, one}}, {{one, one, zero}}, {{zero, one, one}}}};
        s.mds_is_test_ = true;
        static const std::vector<std::array<Fr, 3>> ark = [] {   // drawn once per process
            std::array<uint8_t, 32> seed = {1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            StdRng rng(seed);
            std::vector<std::array<Fr, 3>> a(8 + 31);
            for (auto& row : a)
                for (auto& x : row) x = field_rand<E>(rng);
            return a;
        }();
        s.ark_ = ark;
        return s;
    }

    // absorb(&Vec<u8>): LE64(len) || bytes, packed (MODULUS_BIT_SIZE - 1) / 8 bytes per field element (31 for BN254 Fr, 47 for
    // BLS12-377 Fq), little endian
    void absorb_bytes(const uint8_t* data, size_t len) {
        std::vector<uint8_t> bytes(8 + len);
        for (int i = 0; i < 8; i++) bytes[i] = (uint8_t)((uint64_t)len >> (8 * i));
        memcpy(bytes.data() + 8, data, len);
        std::vector<Fr> elems;
        for (size_t off = 0; off < bytes.size(); off += kUsableBytes) {
            const size_t take = std::min<size_t>(kUsableBytes, bytes.size() - off);
            Fr canon = F::zero();
            for (size_t i = 0; i < take; i++) canon.l[i / 8] |= (uint64_t)bytes[off + i] << (8 * (i % 8));
            elems.push_back(F::to_mont(canon));
        }
        absorb_elements(elems);
    }
    // absorb(&Vec<F>): the elements as they are
    void absorb_elements(const std::vector<Fr>& elems) {
        if (elems.empty()) return;
        if (squeezing_) {
            permute();
            absorb_internal(0, elems);
        } else {
            size_t idx = next_index_;
            if (idx == kRate) {
                permute();
                idx = 0;
            }
            absorb_internal(idx, elems);
        }
    }
    // absorb_elements on EIGHT sponges at once (the proofs of a batch run the same schedule on independent states): with AVX-512
    // IFMA the eight states ride the eight lanes of a vector (poseidon_ifma.hpp, about five times the scalar rate); without it,
    // or when the sponges are not in step (mode, position, lengths -- e.g. polynomials trimmed to different degrees), or for
    // another field or sponge shape, eight ordinary calls.  Same states either way, bit for bit.
    static bool same_ark(const std::vector<std::array<Fr, 3>>& a, const std::vector<std::array<Fr, 3>>& b) {
        return a.size() == b.size() && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(a[0])) == 0);
    }
    static void absorb_elements_x8(PoseidonSpongeT* const sp[8], const std::vector<Fr>* const elems[8]) {
        if constexpr (std::is_same<E, lg_host::Fr>::value) {
            bool vec = ifma::available() && !elems[0]->empty() && sp[0]->alpha_ == 17 && sp[0]->mds_is_test_;
            for (size_t j = 1; j < 8 && vec; j++)
                vec = sp[j]->squeezing_ == sp[0]->squeezing_ && sp[j]->next_index_ == sp[0]->next_index_ && elems[j]->size() == elems[0]->size() &&
                      sp[j]->alpha_ == 17 && sp[j]->mds_is_test_ && sp[j]->full_rounds_ == sp[0]->full_rounds_ &&
                      sp[j]->partial_rounds_ == sp[0]->partial_rounds_ && same_ark(sp[j]->ark_, sp[0]->ark_);
#if LG_HAVE_IFMA_BUILD
            if (vec) {
                // one engine per process: every sponge here is a test_sponge() (the comparison above checked the round constants)
                static const std::vector<std::array<Fr, 3>> engine_ark = sp[0]->ark_;
                static const ifma::Engine engine(engine_ark, sp[0]->full_rounds_, sp[0]->partial_rounds_);
                if (same_ark(sp[0]->ark_, engine_ark) && sp[0]->full_rounds_ == 8 && sp[0]->partial_rounds_ == 31) {
                    Fr* states[8];
                    const Fr* data[8];
                    for (size_t j = 0; j < 8; j++) { states[j] = sp[j]->state_.data(); data[j] = elems[j]->data(); }
                    const bool first = sp[0]->squeezing_ || sp[0]->next_index_ == kRate;
                    const size_t start = first ? 0 : sp[0]->next_index_;
                    const size_t next = engine.absorb8(states, data, elems[0]->size(), start, first);
                    for (size_t j = 0; j < 8; j++) { sp[j]->squeezing_ = false; sp[j]->next_index_ = next; }
                    return;
                }
            }
#endif
        }
        for (size_t j = 0; j < 8; j++) sp[j]->absorb_elements(*elems[j]);
    }
    std::vector<Fr> squeeze_native_field_elements(size_t n) {
        std::vector<Fr> out(n);
        if (!squeezing_) {
            permute();
            squeeze_internal(0, out);
        } else {
            size_t idx = next_index_;
            if (idx == kRate) {
                permute();
                idx = 0;
            }
            squeeze_internal(idx, out);
        }
        return out;
    }
    // squeeze_bytes: ceil(n / usable) elements, the low `usable` little-endian bytes of each, truncated to n
    std::vector<uint8_t> squeeze_bytes(size_t n) {
        const size_t usable = kUsableBytes, nelem = (n + usable - 1) / usable;
        const std::vector<Fr> src = squeeze_native_field_elements(nelem);
        std::vector<uint8_t> bytes;
        for (const Fr& e : src) {
            const Fr c = F::from_mont(e);
            for (size_t i = 0; i < usable; i++) bytes.push_back((uint8_t)(c.l[i / 8] >> (8 * (i % 8))));
        }
        bytes.resize(n);
        return bytes;
    }
    // the parameters, for a device that runs the same sponge (include/ligero_hip.h lg_prover_setup)
    size_t full_rounds() const { return full_rounds_; }
    size_t partial_rounds() const { return partial_rounds_; }
    uint64_t alpha() const { return alpha_; }
    const std::vector<std::array<Fr, 3>>& ark() const { return ark_; }
    const std::array<std::array<Fr, 3>, 3>& mds() const { return mds_; }
    std::array<uint8_t, 32> squeeze_seed() {
        const std::vector<uint8_t> b = squeeze_bytes(kChachaSeedBytes);
        std::array<uint8_t, 32> s;
        memcpy(s.data(), b.data(), 32);
        return s;
    }

private:
    static constexpr size_t kRate = 2, kCapacity = 1, kWidth = 3;
    // x^17: four squarings and one product (alpha is fixed by test_sponge)
    // (the five products are one dependent chain and the transcript of a large proof is 10^4 permutations in a row: where the host has the
    // fast product they run without their final subtractions -- values below 2p in, below 2p out -- and the result is reduced once; which
    // product is decided once per permutation, not per product)
    template <bool kAdx>
    __attribute__((always_inline)) static inline Fr sbox17(const Fr& x) {
#ifdef LG_HOST_HAVE_ADX_PATH
        if constexpr (kAdx && std::is_same<Fr, lg_host::Fr>::value) {
            Fr y = lg_host::mul_lazy_adx(x, x);
            y = lg_host::mul_lazy_adx(y, y);
            y = lg_host::mul_lazy_adx(y, y);
            y = lg_host::mul_lazy_adx(y, y);
            return lg_host::reduce_lazy(lg_host::mul_lazy_adx(y, x));
        }
#endif
        Fr y = fr_mul(x, x);
        y = fr_mul(y, y);
        y = fr_mul(y, y);
        y = fr_mul(y, y);
        return fr_mul(y, x);
    }
    void permute() {
#ifdef LG_HOST_HAVE_ADX_PATH
        if constexpr (std::is_same<Fr, lg_host::Fr>::value) {
            if (lg_host::have_adx() && alpha_ == 17 && mds_is_test_) { permute_chain(); return; }
        }
#endif
        if (lg_host::have_adx()) permute_impl<true>();
        else permute_impl<false>();
    }
#ifdef LG_HOST_HAVE_ADX_PATH
    // The configuration every prover of this repository runs (test_sponge: alpha = 17, the additions-only MDS) with the state in three
    // locals and nothing between the products but what the permutation needs: the transcript of a large proof is one chain of these.
    // The partial rounds' linear layer, folded: only the element that takes the S-box needs its round constant where it stands -- the
    // constants of the two idle elements travel through the (additions-only) mixing as constants and are settled once, after the last
    // partial round.  With s1 = u1 + k1, s2 = u2 + k2 (u the data, k known): u1' = a + u1, u2' = u1 + u2, and the next S-box input is
    // a + (u2 + d) with d = k2 + c2 + the next c0 -- four additions per round instead of six, ONE of them behind the S-box instead of
    // two.  folded_d_[0] = c0 of the first partial round, [j] = K2_{j-1} + c0_j, [P] = K2_{P-1}; K1_j = K1_{j-1} + c1_j,
    // K2_j = K1_{j-1} + K2_{j-1} + c2_j.  Same states, bit for bit (tests/test_transcript.py against the model's plain rounds).
    void ensure_folded() {
        if (!folded_d_.empty()) return;
        const size_t half = full_rounds_ / 2, P = partial_rounds_;
        folded_d_.resize(P + 1);
        Fr k1 = fr_zero(), k2 = fr_zero();
        for (size_t j = 0; j < P; j++) {
            folded_d_[j] = fr_add(k2, ark_[half + j][0]);
            const Fr n1 = fr_add(k1, ark_[half + j][1]), n2 = fr_add(fr_add(k1, k2), ark_[half + j][2]);
            k1 = n1; k2 = n2;
        }
        folded_d_[P] = k2;
        folded_k1_ = k1;
        folded_k2_ = fr_add(k1, k2);
    }
    // (-fno-gcse for this one function: global common-subexpression elimination lengthens the live ranges between the blocks of
    // instructions below and the register allocator pays for it -- 3.28 -> 3.06 us per permutation, same box, same arithmetic)
#if defined(__GNUC__) && !defined(__clang__)
    __attribute__((optimize("no-gcse")))
#endif
    void permute_chain() {
        using lg_host::add_mod;
        ensure_folded();
        Fr s0 = state_[0], s1 = state_[1], s2 = state_[2];
        const size_t half = full_rounds_ / 2, rounds = full_rounds_ + partial_rounds_, P = partial_rounds_;
        const std::array<Fr, 3>* ark = ark_.data();
        const Fr* d = folded_d_.data();
        Fr u1 = s1, u2 = s2, t0 = s0;
        // ONE loop over all rounds with the kind of round tested inside: the structure the compiler allocates registers best for (the
        // same arithmetic as three loops measured 3.7 us per permutation on the GPU box's EPYC, this 3.3; the S-box as a call +0.3)
        for (size_t i = 0; i < rounds; i++) {
            if (i < half || i >= half + P) {
                // a full round's three S-boxes, product by product side by side: written one S-box after the other, the five dependent
                // products of the first fill the out-of-order window before the core ever sees the second (156 ns per round; 3 x 58)
                using lg_host::mul_lazy_adx;
                s0 = add_mod(s0, ark[i][0]); s1 = add_mod(s1, ark[i][1]); s2 = add_mod(s2, ark[i][2]);
                Fr a = mul_lazy_adx(s0, s0), b = mul_lazy_adx(s1, s1), c = mul_lazy_adx(s2, s2);
                a = mul_lazy_adx(a, a); b = mul_lazy_adx(b, b); c = mul_lazy_adx(c, c);
                a = mul_lazy_adx(a, a); b = mul_lazy_adx(b, b); c = mul_lazy_adx(c, c);
                a = mul_lazy_adx(a, a); b = mul_lazy_adx(b, b); c = mul_lazy_adx(c, c);
                s0 = lg_host::reduce_lazy(mul_lazy_adx(a, s0)); s1 = lg_host::reduce_lazy(mul_lazy_adx(b, s1)); s2 = lg_host::reduce_lazy(mul_lazy_adx(c, s2));
                const Fr n0 = add_mod(s0, s2), n1 = add_mod(s0, s1), n2 = add_mod(s1, s2);
                s0 = n0; s1 = n1; s2 = n2;
                if (i + 1 == half) { u1 = s1; u2 = s2; t0 = add_mod(s0, d[0]); }
            } else {
                const size_t j = i - half;
                const Fr a = lg_host::reduce_lazy(lg_host::sbox17_lazy_adx(t0));     // (one block of instructions: the value stays in its registers)
                const Fr y = add_mod(u2, d[j + 1]);
                const Fr nu1 = add_mod(a, u1), nu2 = add_mod(u1, u2);
                t0 = add_mod(a, y);
                u1 = nu1; u2 = nu2;
                if (j + 1 == P) { s0 = t0; s1 = add_mod(u1, folded_k1_); s2 = add_mod(u2, folded_k2_); }
            }
        }
        state_[0] = s0; state_[1] = s1; state_[2] = s2;
    }
#endif
    template <bool kAdx>
    void permute_impl() {
        const size_t half = full_rounds_ / 2;
        for (size_t i = 0; i < full_rounds_ + partial_rounds_; i++) {
            for (size_t j = 0; j < kWidth; j++) state_[j] = fr_add(state_[j], ark_[i][j]);
            const bool full = i < half || i >= half + partial_rounds_;
            if (alpha_ == 17) {
                state_[0] = sbox17<kAdx>(state_[0]);
                if (full) { state_[1] = sbox17<kAdx>(state_[1]); state_[2] = sbox17<kAdx>(state_[2]); }
            } else {
                for (size_t j = 0; j < (full ? kWidth : 1); j++) state_[j] = fr_pow_u64(state_[j], alpha_);
            }
            std::array<Fr, 3> next;
            if (mds_is_test_) {  // [[1,0,1],[1,1,0],[0,1,1]]: additions only
                next[0] = fr_add(state_[0], state_[2]);
                next[1] = fr_add(state_[0], state_[1]);
                next[2] = fr_add(state_[1], state_[2]);
            } else {
                for (size_t r = 0; r < kWidth; r++) {
                    Fr acc = fr_zero();
                    for (size_t j = 0; j < kWidth; j++) acc = fr_add(acc, fr_mul(state_[j], mds_[r][j]));
                    next[r] = acc;
                }
            }
            state_ = next;
        }
    }
    void absorb_internal(size_t start, const std::vector<Fr>& elems) {
        size_t pos = 0;
        for (;;) {
            const size_t left = elems.size() - pos;
            if (start + left <= kRate) {
                for (size_t i = 0; i < left; i++) state_[kCapacity + start + i] = fr_add(state_[kCapacity + start + i], elems[pos + i]);
                squeezing_ = false;
                next_index_ = start + left;
                return;
            }
            const size_t take = kRate - start;
            for (size_t i = 0; i < take; i++) state_[kCapacity + start + i] = fr_add(state_[kCapacity + start + i], elems[pos + i]);
            permute();
            pos += take;
            start = 0;
        }
    }
    void squeeze_internal(size_t start, std::vector<Fr>& out) {
        size_t pos = 0;
        for (;;) {
            const size_t left = out.size() - pos;
            if (start + left <= kRate) {
                for (size_t i = 0; i < left; i++) out[pos + i] = state_[kCapacity + start + i];
                squeezing_ = true;
                next_index_ = start + left;
                return;
            }
            const size_t take = kRate - start;
            for (size_t i = 0; i < take; i++) out[pos + i] = state_[kCapacity + start + i];
            if (left != kRate) permute();  // "unless we are done with squeezing in this call, permute"
            pos += take;
            start = 0;
        }
    }
    size_t full_rounds_ = 0, partial_rounds_ = 0;
    uint64_t alpha_ = 0;
    std::array<std::array<Fr, 3>, 3> mds_;
    bool mds_is_test_ = false;
    std::vector<std::array<Fr, 3>> ark_;
    std::vector<Fr> folded_d_;                 // permute_chain's folded partial-round constants (ensure_folded), made on first use
    Fr folded_k1_ = fr_zero(), folded_k2_ = fr_zero();
    std::array<Fr, 3> state_ = {fr_zero(), fr_zero(), fr_zero()};
    bool squeezing_ = false;  // DuplexSpongeMode
    size_t next_index_ = 0;
};
using PoseidonSponge = PoseidonSpongeT<Fr>;

}  // namespace ligero
