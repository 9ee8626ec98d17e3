// Encode-and-commit kernels over the generic field arithmetic (generic_field.h): the same path as ntt_kernels.h /
// hash_kernels.h -- reed_solomon_interpolate / reed_solomon_evaluate (src/ligero/mod.rs:998-1008 under the row loops at
// 521-533), the column hash Blake2s-256(LE64(rows) || canonical elements) (mod.rs:536-542, types.rs:18) with elements of
// 4 NW bytes (48 for ark_bls12_377::Fq), openings (mod.rs:944-952) -- written for correctness and reasonable speed, not
// tuned: radix-2 transforms of one row per workgroup in LDS, one lane per column in the hash.  The codeword uses the same
// coset-plane layout as the fast path: plane s holds columns j = 8 q + s as [row][q], canonical integers.
#pragma once
#include <type_traits>

#include "generic_field.h"
#include "hash_kernels.h"

namespace lg {

template <int NW>
struct GfNttArgs {
    const gfe<NW>* in;        // [rows][k] Montgomery (message rows or coefficient rows)
    gfe<NW>* out;             // interpolate: [rows][k] Montgomery coefficients; evaluate: planes [8][plane rows][k] canonical
    const gfe<NW>* tw;        // [k/2] powers of the size-k root of this direction, Montgomery
    const gfe<NW>* wn;        // evaluate: [n] powers of omega_n, Montgomery
    gfe<NW> scale;            // interpolate: 1/k in Montgomery form; evaluate: the integer 1 (leaves Montgomery form)
    GfConsts<NW> F;
    uint32_t rows, k, logk, n;
    uint32_t ncos;            // evaluate: planes per row in `cosets`; interpolate: 0
    uint8_t cosets[8];
    uint64_t plane_stride;    // elements between planes
    uint32_t evaluate;
};

// one workgroup = one size-k transform (a row, or one coset of a row); LDS holds it word-interleaved: word w of element i at
// lds[w * k + i], so lanes on adjacent elements hit adjacent banks
template <int NW>
__global__ void __launch_bounds__(256) gf_ntt_rows_kernel(const GfNttArgs<NW> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gf_smem[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(gf_smem);
    const uint32_t K = a.k, tid = threadIdx.x;
    const uint32_t per = a.evaluate ? a.ncos : 1u;
    const uint32_t row = blockIdx.x / per;
    const uint32_t s = a.evaluate ? a.cosets[blockIdx.x % per] : 0u;
    auto get = [&](uint32_t i) { gfe<NW> e;
#pragma unroll
        for (int w = 0; w < NW; w++) e.v[w] = lds[w * K + i];
        return e; };
    auto put = [&](uint32_t i, const gfe<NW>& e) {
#pragma unroll
        for (int w = 0; w < NW; w++) lds[w * K + i] = e.v[w]; };
    const gfe<NW>* src = a.in + (size_t)row * K;
    for (uint32_t i = tid; i < K; i += 256) {
        gfe<NW> x = gf_load<NW>(src + i);
        if (a.evaluate && s != 0) {                        // coset s of the order-k subgroup: c[d] * omega_n^(s d)
            gfe<NW> y;
            gf_mul<NW>(y, x, gf_load<NW>(a.wn + (((uint64_t)s * i) & (a.n - 1))), a.F);
            x = y;
        }
        put(__brev(i) >> (32 - a.logk), x);                // decimation in time: bit-reversed input, natural output
    }
    __syncthreads();
    for (uint32_t len = 2, shift = a.logk - 1; len <= K; len <<= 1, shift--) {
        const uint32_t half = len >> 1;
        for (uint32_t bf = tid; bf < (K >> 1); bf += 256) {
            const uint32_t j = bf & (half - 1), i0 = ((bf - j) << 1) + j, i1 = i0 + half;
            gfe<NW> u = get(i0), v, sum, dif;
            gf_mul<NW>(v, get(i1), gf_load<NW>(a.tw + ((size_t)j << shift)), a.F);   // omega_k^(j k / len)
            gf_add<NW>(sum, u, v, a.F);
            gf_sub<NW>(dif, u, v, a.F);
            put(i0, sum);
            put(i1, dif);
        }
        __syncthreads();
    }
    gfe<NW>* dst = a.evaluate ? a.out + (size_t)s * a.plane_stride + (size_t)row * K : a.out + (size_t)row * K;
    for (uint32_t i = tid; i < K; i += 256) {
        gfe<NW> y;
        gf_mul<NW>(y, get(i), a.scale, a.F);
        gf_store<NW>(dst + i, y);
    }
}

template <int NW>
struct GfHashArgs {
    const gfe<NW>* u;        // planes, canonical
    uint8_t* leaves;         // [batch][n][32]
    uint32_t rows, k, proofs;
    uint64_t plane_stride;   // elements
};

template <int B, int E, class Fn>
__device__ __forceinline__ void gf_static_for(Fn&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        gf_static_for<B + 1, E>(f);
    }
}

// One lane per column; the byte stream LE64(rows) || elements is cut into 64-byte blocks at compile-time positions: with
// NW words per element the block boundaries repeat every PER = lcm(16, NW) / NW rows (2 rows for 32-byte elements, 4 rows =
// three blocks for 48-byte elements).  8 + 4 NW rows is never a multiple of 64 for NW = 8, 12, so the final block is partial.
template <int NW>
__global__ void __launch_bounds__(256) gf_blake2s_columns_kernel(const GfHashArgs<NW> a) {
    constexpr int PER = (NW == 8) ? 2 : 4;
    static_assert((PER * NW) % 16 == 0, "period must end on a block boundary");
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = (uint64_t)a.proofs * 8 * a.k;
    if (gid >= total) return;
    const uint32_t q = (uint32_t)(gid % a.k);
    const uint32_t s = (uint32_t)((gid / a.k) & 7);
    const uint32_t b = (uint32_t)((gid / a.k) >> 3);
    const gfe<NW>* p = a.u + (uint64_t)s * a.plane_stride + ((uint64_t)b * a.rows) * a.k + q;
    uint32_t h[8], m[16];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = b2s_iv(i);
    h[0] ^= 0x01010020u;
    m[0] = a.rows;   // LE64(rows): serialize_compressed length prefix of Vec<F>
    m[1] = 0;
    uint64_t t = 0;
    // absorb R rows starting at stream word position 2 (mod 16); returns nothing: m / h / t are updated in place
    auto absorb = [&](const gfe<NW>* rowp, auto rc) {
        constexpr int R = decltype(rc)::value;
        gf_static_for<0, R>([&](auto ri) {
            constexpr int r = decltype(ri)::value;
            const gfe<NW> e = gf_load<NW>(rowp + (uint64_t)r * a.k);
            gf_static_for<0, NW>([&](auto wi) {
                constexpr int w = decltype(wi)::value;
                constexpr int pos = (2 + r * NW + w) % 16;
                m[pos] = e.v[w];
                if constexpr (pos == 15) {
                    t += 64;
                    b2s_compress(h, m, (uint32_t)t, (uint32_t)(t >> 32), false);
                }
            });
        });
    };
    uint32_t r0 = 0;
    for (; r0 + PER <= a.rows; r0 += PER) absorb(p + (uint64_t)r0 * a.k, std::integral_constant<int, PER>{});
    const uint32_t rem = a.rows - r0;
    uint32_t fill = 2;                                     // words of the last, partial block that are occupied
    gf_static_for<1, PER>([&](auto rc) {
        constexpr int R = decltype(rc)::value;
        if (rem == (uint32_t)R) {
            absorb(p + (uint64_t)r0 * a.k, rc);
            fill = (2 + R * NW) % 16;
        }
    });
    gf_static_for<0, 16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if ((uint32_t)i >= fill) m[i] = 0;
    });
    t = 8 + 4ull * NW * a.rows;
    b2s_compress(h, m, (uint32_t)t, (uint32_t)(t >> 32), true);
    uint4* out = reinterpret_cast<uint4*>(a.leaves + 32 * ((((uint64_t)b * a.k + q) << 3) + s));
    out[0] = make_uint4(h[0], h[1], h[2], h[3]);
    out[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// opened columns (src/matrices/mod.rs:169-171): canonical planes -> Montgomery, [proof][t][rows]
template <int NW>
__global__ void __launch_bounds__(256) gf_gather_columns_kernel(const gfe<NW>* u, uint64_t plane_stride, const uint32_t* idx, uint32_t t, uint32_t rows,
                                                                uint32_t k, uint32_t proof0, gfe<NW> r2, GfConsts<NW> F, gfe<NW>* cols) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)t * rows) return;
    const uint32_t pr = blockIdx.y;
    const uint32_t c = (uint32_t)(gid / rows), i = (uint32_t)(gid % rows);
    const uint32_t j = idx[(uint64_t)pr * t + c];
    const uint32_t s = j & 7, q = j >> 3;
    gfe<NW> y;
    gf_mul<NW>(y, gf_load<NW>(u + (uint64_t)s * plane_stride + ((uint64_t)(proof0 + pr) * rows + i) * k + q), r2, F);
    gf_store<NW>(cols + (uint64_t)pr * t * rows + gid, y);
}

// planes (canonical) -> rows in natural column order (Montgomery): out[i][8 q + s]
template <int NW>
__global__ void __launch_bounds__(256) gf_planes_to_rows_kernel(const gfe<NW>* u, uint64_t plane_stride, uint64_t row_base, uint32_t nrows, uint32_t k,
                                                                gfe<NW> r2, GfConsts<NW> F, gfe<NW>* out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (uint64_t)nrows * k * 8) return;
    const uint32_t q = (uint32_t)(gid % k);
    const uint32_t s = (uint32_t)((gid / k) & 7);
    const uint64_t i = (gid / k) >> 3;
    gfe<NW> y;
    gf_mul<NW>(y, gf_load<NW>(u + (uint64_t)s * plane_stride + (row_base + i) * k + q), r2, F);
    gf_store<NW>(out + (i * k + q) * 8 + s, y);
}

}  // namespace lg

namespace lg {

// ---- sub-proof reductions over a generic field (mod.rs:658, 723-736, 842-848), one thread per output point, a loop over rows.
// Written for the reference's second test field, whose only circuit has ten nodes: correctness first.

// prove_interleaved: out[c] = sum_i r[i] * preenc_u[i][c]   (Montgomery x Montgomery -> Montgomery)
template <int NW>
__global__ void __launch_bounds__(256) gf_row_mul_kernel(const gfe<NW>* pre, const gfe<NW>* r, uint32_t rows, uint32_t k, GfConsts<NW> F, gfe<NW>* out) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k) return;
    gfe<NW> acc;
#pragma unroll
    for (int w = 0; w < NW; w++) acc.v[w] = 0;
    for (uint32_t i = 0; i < rows; i++) {
        gfe<NW> t, s;
        gf_mul<NW>(t, gf_load<NW>(pre + (size_t)i * k + c), gf_load<NW>(r + i), F);
        gf_add<NW>(s, acc, t, F);
        acc = s;
    }
    gf_store<NW>(out + c, acc);
}

// point j of the size-2k domain is codeword index 4 j: plane (j & 1) * 4, slot j >> 1
// linear test: q[j] = sum_i u_i(eta_j) * r_i(eta_j), both canonical in their planes; out Montgomery (x R^3 after the sum of a b / R)
template <int NW>
__global__ void __launch_bounds__(256) gf_linear_points_kernel(const gfe<NW>* u, const gfe<NW>* rv, uint64_t plane_stride, uint32_t rows, uint32_t k,
                                                               gfe<NW> r3, GfConsts<NW> F, gfe<NW>* out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 2 * k) return;
    const uint64_t off = (uint64_t)((j & 1) * 4) * plane_stride + (j >> 1);
    gfe<NW> acc;
#pragma unroll
    for (int w = 0; w < NW; w++) acc.v[w] = 0;
    for (uint32_t i = 0; i < rows; i++) {
        gfe<NW> t, s;
        gf_mul<NW>(t, gf_load<NW>(u + off + (uint64_t)i * k), gf_load<NW>(rv + off + (uint64_t)i * k), F);
        gf_add<NW>(s, acc, t, F);
        acc = s;
    }
    gfe<NW> y;
    gf_mul<NW>(y, acc, r3, F);
    gf_store<NW>(out + j, y);
}

// quadratic test: q[j] = sum_{i < m} r_i * (x_i(eta_j) * y_i(eta_j) - z_i(eta_j)); rows of U: x = [0, m), y = [m, 2m), z = [2m, 3m)
template <int NW>
__global__ void __launch_bounds__(256) gf_quadratic_points_kernel(const gfe<NW>* u, const gfe<NW>* r, uint64_t plane_stride, uint32_t m, uint32_t k,
                                                                  gfe<NW> one_plain, gfe<NW> r3, GfConsts<NW> F, gfe<NW>* out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= 2 * k) return;
    const uint64_t off = (uint64_t)((j & 1) * 4) * plane_stride + (j >> 1);
    gfe<NW> acc;
#pragma unroll
    for (int w = 0; w < NW; w++) acc.v[w] = 0;
    for (uint32_t i = 0; i < m; i++) {
        gfe<NW> xy, z, d, t, s;
        gf_mul<NW>(xy, gf_load<NW>(u + off + (uint64_t)i * k), gf_load<NW>(u + off + (uint64_t)(m + i) * k), F);   // x y / R
        gf_mul<NW>(z, gf_load<NW>(u + off + (uint64_t)(2 * m + i) * k), one_plain, F);                                // z / R
        gf_sub<NW>(d, xy, z, F);
        gf_mul<NW>(t, d, gf_load<NW>(r + i), F);                                                                      // (x y - z) r / R
        gf_add<NW>(s, acc, t, F);
        acc = s;
    }
    gfe<NW> y;
    gf_mul<NW>(y, acc, r3, F);
    gf_store<NW>(out + j, y);
}

}  // namespace lg
