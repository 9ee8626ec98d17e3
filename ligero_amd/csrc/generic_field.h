// Generic prime-field arithmetic for the second element type of the reference: any modulus of
// NW 32-bit words with at least one spare bit, elements as NW saturated words in Montgomery form (R = 2^(32 NW)) -- the
// in-memory layout of ark_ff::Fp<MontBackend<_, NW / 2>, NW / 2>, so ark_bls12_377::Fq (377 bits, 6 x u64, the field of the
// reference's test_prove_and_verify_bls12_377, src/ligero/tests.rs:23, 186-193) crosses the C ABI as it is.
//
// This is the PORTABLE path: CIOS Montgomery products over saturated 32-bit words with explicit carry chains, every value
// fully reduced.  The BN254 Fr fast path (fr29_gfx950.h: 29-bit unsaturated limbs, Barrett products by table constants)
// is about three times faster per product and stays the one BASELINE's configs run on; no BASELINE config uses Fq.
// Instantiated for NW = 12 (BLS12-377 Fq) and NW = 8 (BN254 Fr again, as a cross-check of these kernels against the
// fast path and the C oracle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lg {

template <int NW>
struct gfe {
    uint32_t v[NW];
};

template <int NW>
struct GfConsts {
    uint32_t p[NW];     // modulus
    uint32_t r2[NW];    // R^2 mod p (to Montgomery form)
    uint32_t inv32;     // -p^-1 mod 2^32
};

// r = t - p if t >= p (t given with an extra top word) else t
template <int NW>
__device__ __forceinline__ void gf_cond_sub(gfe<NW>& r, const uint32_t (&t)[NW], uint32_t top, const GfConsts<NW>& F) {
    uint32_t d[NW];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint32_t bo;
        d[i] = __builtin_subc(t[i], F.p[i], br, &bo);
        br = bo;
    }
    // t >= p  <=>  no borrow out of the NW words, or the extra word is set
    const uint32_t keep = (top == 0 && br) ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < NW; i++) r.v[i] = __builtin_amdgcn_bitop3_b32(keep, t[i], d[i], 0xca);
}

template <int NW>
__device__ __forceinline__ void gf_add(gfe<NW>& r, const gfe<NW>& a, const gfe<NW>& b, const GfConsts<NW>& F) {
    uint32_t t[NW];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint32_t co;
        t[i] = __builtin_addc(a.v[i], b.v[i], c, &co);
        c = co;
    }
    gf_cond_sub<NW>(r, t, c, F);
}

template <int NW>
__device__ __forceinline__ void gf_sub(gfe<NW>& r, const gfe<NW>& a, const gfe<NW>& b, const GfConsts<NW>& F) {
    uint32_t d[NW];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint32_t bo;
        d[i] = __builtin_subc(a.v[i], b.v[i], br, &bo);
        br = bo;
    }
    const uint32_t mask = 0u - br;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint32_t co;
        r.v[i] = __builtin_addc(d[i], F.p[i] & mask, c, &co);
        c = co;
    }
}

// Montgomery product a b R^-1 mod p, fully reduced (CIOS; the spare bit(s) of p keep the running sum within NW + 1 words)
template <int NW>
__device__ __forceinline__ void gf_mul(gfe<NW>& r, const gfe<NW>& a, const gfe<NW>& b, const GfConsts<NW>& F) {
    uint32_t t[NW];
    uint32_t top = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < NW; j++) {
            c = (uint64_t)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        const uint64_t hi = (uint64_t)top + c;      // <= 2^33
        const uint32_t m = t[0] * F.inv32;
        c = ((uint64_t)m * F.p[0] + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < NW; j++) {
            c = (uint64_t)m * F.p[j] + t[j] + c;
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += hi;
        t[NW - 1] = (uint32_t)c;
        top = (uint32_t)(c >> 32);
    }
    gf_cond_sub<NW>(r, t, top, F);
}

template <int NW>
__device__ __forceinline__ gfe<NW> gf_load(const gfe<NW>* p) {
    gfe<NW> r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < NW / 4; i++) {
        const uint4 x = q[i];
        r.v[4 * i] = x.x; r.v[4 * i + 1] = x.y; r.v[4 * i + 2] = x.z; r.v[4 * i + 3] = x.w;
    }
    return r;
}
template <int NW>
__device__ __forceinline__ void gf_store(gfe<NW>* p, const gfe<NW>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < NW / 4; i++) q[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}

}  // namespace lg
