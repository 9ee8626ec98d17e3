// BN254 Fr arithmetic for gfx950 (CDNA4): 8 x 32-bit limbs, Montgomery form, R = 2^256.
//
// Replaces ark_ff::Fp<MontBackend<FrConfig,4>> arithmetic on the hot path of
// NP-Eng/ligero (every `F` op under src/ligero/mod.rs:521-533).
// The 32x32->64 multiply-add (v_mad_u64_u32) is the only wide multiplier the
// vector ALU has, so everything is built from it.  All loops are over
// compile-time-constant indices so limbs stay in VGPRs.
//
// Value ranges: "lazy" elements live in [0, 2p); fr_reduce() brings them to
// [0, p).  p < 2^254, so 4p < 2^256 and sums of two lazy values never wrap.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lg {

struct alignas(16) fr {
    uint32_t v[8];
};

// modulus p, 2p, and -p^-1 mod 2^32
#define LG_P0 0xf0000001u
#define LG_P1 0x43e1f593u
#define LG_P2 0x79b97091u
#define LG_P3 0x2833e848u
#define LG_P4 0x8181585du
#define LG_P5 0xb85045b6u
#define LG_P6 0xe131a029u
#define LG_P7 0x30644e72u
#define LG_INV32 0xefffffffu

__device__ __forceinline__ constexpr uint32_t fr_p(int i) {
    constexpr uint32_t P[8] = {LG_P0, LG_P1, LG_P2, LG_P3, LG_P4, LG_P5, LG_P6, LG_P7};
    return P[i];
}
__device__ __forceinline__ constexpr uint32_t fr_2p(int i) {
    constexpr uint32_t P2[8] = {0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u,
                                0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u};
    return P2[i];
}

__device__ __forceinline__ uint64_t mad_wide(uint32_t a, uint32_t b, uint64_t c) {
    return (uint64_t)a * b + c;  // v_mad_u64_u32
}

// Carry chains are written with the add/sub-with-carry builtins (v_add_co / v_addc_co / v_sub_co / v_subb_co, 1.75 ns per
// wave-instruction) and selections with v_bitop3_b32 (a full-rate three-input boolean, 1.0 ns): expressed in 64-bit C
// arithmetic the compiler produced v_lshl_add_u64 pairs plus v_cndmask_b32 ..., vcc, and the VCC form of v_cndmask
// issues at ~10 ns per wave-instruction on gfx950 (tools/microbench5.hip, profiles/r02_microbench5_instruction_issue.log)
// -- the eight selects of one conditional subtraction cost more than sixty multiplies.
// select(mask, a, d) = (mask & a) | (~mask & d)
__device__ __forceinline__ uint32_t fr_select(uint32_t mask, uint32_t a, uint32_t d) { return __builtin_amdgcn_bitop3_b32(mask, a, d, 0xca); }

// r = a + b (no reduction; caller guarantees no wrap)
__device__ __forceinline__ void fr_add_raw(fr& r, const fr& a, const fr& b) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t co;
        r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &co);
        c = co;
    }
}

// r = a - m if a >= m else a, where m is the constant multiple of p selected by TWO_P
template <bool TWO_P>
__device__ __forceinline__ void fr_cond_sub(fr& r, const fr& a) {
#ifdef LG_CONDSUB_C64   // the round-1 form (64-bit C arithmetic + ?:), kept for A/B builds
    uint32_t dd[8];
    int64_t cc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        cc += (int64_t)a.v[i] - (int64_t)(TWO_P ? fr_2p(i) : fr_p(i));
        dd[i] = (uint32_t)cc;
        cc >>= 32;
    }
    const bool neg = cc < 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = neg ? a.v[i] : dd[i];
    return;
#endif
    uint32_t d[8];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t bo;
        d[i] = __builtin_subc(a.v[i], TWO_P ? fr_2p(i) : fr_p(i), br, &bo);
        br = bo;
    }
    const uint32_t keep = 0u - br;   // all ones when a < m
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = fr_select(keep, a.v[i], d[i]);
}

// lazy add: inputs < 2p, output < 2p
__device__ __forceinline__ void fr_add_lazy(fr& r, const fr& a, const fr& b) {
    fr t;
    fr_add_raw(t, a, b);
    fr_cond_sub<true>(r, t);
}

// lazy sub: inputs < 2p, output < 2p  (a - b, +2p if negative)
__device__ __forceinline__ void fr_sub_lazy(fr& r, const fr& a, const fr& b) {
    uint32_t d[8];
    uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t bo;
        d[i] = __builtin_subc(a.v[i], b.v[i], br, &bo);
        br = bo;
    }
    const uint32_t mask = 0u - br;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t co;
        r.v[i] = __builtin_addc(d[i], fr_2p(i) & mask, c, &co);
        c = co;
    }
}

// full reduction [0,2p) -> [0,p)
__device__ __forceinline__ void fr_reduce(fr& r, const fr& a) { fr_cond_sub<false>(r, a); }

// Montgomery product: r = a*b*2^-256 mod p, lazy.  Requires a*b < 2^256 * p, which holds
// for a < 4p (5p) and b < p, or a, b < 2p.  Output < 2p.  CIOS over 32-bit limbs; the top
// word never overflows because p has two spare bits ("no-carry" Montgomery).
__device__ __forceinline__ void fr_mul_lazy(fr& r, const fr& a, const fr& b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c = mad_wide(a.v[j], b.v[i], (uint64_t)t[j] + c);
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        uint32_t t8 = t[8] + (uint32_t)c;
        uint32_t m = t[0] * LG_INV32;
        c = mad_wide(m, fr_p(0), (uint64_t)t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c = mad_wide(m, fr_p(j), (uint64_t)t[j] + c);
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t8;
        t[7] = (uint32_t)c;
        t[8] = (uint32_t)(c >> 32);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
}

// r = a * 2^-256 mod p, fully reduced: Montgomery form -> canonical integer (a < 2^256)
__device__ __forceinline__ void fr_from_mont(fr& r, const fr& a) {
    uint32_t t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = a.v[i];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t m = t[0] * LG_INV32;
        uint64_t c = mad_wide(m, fr_p(0), (uint64_t)t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c = mad_wide(m, fr_p(j), (uint64_t)t[j] + c);
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        t[7] = (uint32_t)c;
    }
    fr u;
#pragma unroll
    for (int i = 0; i < 8; i++) u.v[i] = t[i];
    fr_cond_sub<false>(r, u);
}

__device__ __forceinline__ fr fr_load(const fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    fr r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void fr_store(fr* p, const fr& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
// streaming variants for data that is touched once per kernel (matrix rows in, codeword out): the
// nontemporal hint keeps them from evicting the twiddle tables every workgroup re-reads from L2
typedef uint32_t lg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fr fr_load_stream(const fr* p) {
    const lg_u32x4* q = reinterpret_cast<const lg_u32x4*>(p);
    const lg_u32x4 lo = __builtin_nontemporal_load(q), hi = __builtin_nontemporal_load(q + 1);
    fr r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
__device__ __forceinline__ void fr_store_stream(fr* p, const fr& a) {
    lg_u32x4* q = reinterpret_cast<lg_u32x4*>(p);
    lg_u32x4 lo, hi;
    lo.x = a.v[0]; lo.y = a.v[1]; lo.z = a.v[2]; lo.w = a.v[3];
    hi.x = a.v[4]; hi.y = a.v[5]; hi.z = a.v[6]; hi.w = a.v[7];
#ifdef LG_PLAIN_STORES
    *q = lo;
    q[1] = hi;
#else
    __builtin_nontemporal_store(lo, q);
    __builtin_nontemporal_store(hi, q + 1);
#endif
}

}  // namespace lg
