// Reed-Solomon row encoding kernels for gfx950: size-k inverse NTT (interpolate) and
// coset-pruned size-n evaluation (7 size-k NTTs on pre-scaled coefficients; coset 0 of the
// order-n domain is the message itself because the code is systematic).
//
// Replaces LigeroCircuit::reed_solomon_interpolate / reed_solomon_evaluate
// (src/ligero/mod.rs:998-1008) as driven by the row loops at mod.rs:521-533.
//
// Each NTT lives entirely in LDS (two 16-byte planes per element so that ds_read_b128 /
// ds_write_b128 of lane-contiguous elements are conflict free), butterflies are radix-8
// (radix-2/4 for the remainder pass) in registers, decimation in frequency, in place; the
// digit-reversed result is read back permuted so that global stores are coalesced.
#pragma once
#include <type_traits>

#include "fr_gfx950.h"

namespace lg {

constexpr int kMaxLdsLogK = 12;  // 4096 elements * 32 B = 128 KiB (+ padding) fits the 160 KiB LDS

struct NttArgs {
    const fr* in;        // interpolate: message rows [rows][k]; evaluate: coefficient rows [rows][k]
    fr* out;             // interpolate: coefficient rows; evaluate: base of the coset planes [8][rows][k]
    fr* canon_out;       // interpolate only: canonical (non-Montgomery) copy of the message = coset plane 0 (may be null)
    const fr* tw;        // w^e, e < k, Montgomery form; w = omega_k^-1 (interpolate) or omega_k (evaluate)
    const fr* coset_tw;  // evaluate only: omega_n^e, e < n, canonical integers (pre-scale leaves Montgomery form)
    fr w8[3];            // w_8^1, w_8^2 (= w_4), w_8^3 for this direction, Montgomery form
    fr scale;            // interpolate only: 1/k, Montgomery form
    uint32_t rows;       // rows handled by this launch
    uint32_t row0;       // first row (offset into in/out)
    uint32_t ncos;       // evaluate only: number of cosets in `cosets`
    uint32_t cosets[8];  // evaluate only: coset ids (1..7)
    uint64_t plane_stride;  // elements between coset planes (= total_rows * k)
};

// compile-time loop: f(integral_constant<int, I>) for I in [B, E) -- expanded in the front end, so
// register arrays are indexed by constants whatever the optimiser's unroll budget says
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ int lds_slot(int pos) { return pos + (pos >> 4); }
__host__ __device__ constexpr int lds_slots(int k) { return k + (k >> 4); }

struct LdsRow {
    uint4* lo;
    uint4* hi;
    __device__ __forceinline__ fr get(int pos) const {
        int s = lds_slot(pos);
        uint4 a = lo[s], b = hi[s];
        fr r;
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
        r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
        return r;
    }
    __device__ __forceinline__ void put(int pos, const fr& x) const {
        int s = lds_slot(pos);
        lo[s] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
        hi[s] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
    }
};

// (a, b) <- (a + b, a - b), lazy
__device__ __forceinline__ void bfly(fr& a, fr& b) {
    fr s, d;
    fr_add_lazy(s, a, b);
    fr_sub_lazy(d, a, b);
    a = s;
    b = d;
}

// In-register size-2^LOGR DFT, decimation in frequency.  On return y[m] = sum_q e[q] w_R^(q m)
// (natural order m).  w8[] holds w_8^1..3 of the transform direction.
template <int LOGR>
__device__ __forceinline__ void dft_regs(fr (&e)[1 << LOGR], const fr (&w8)[3]);

template <>
__device__ __forceinline__ void dft_regs<1>(fr (&e)[2], const fr (&)[3]) {
    bfly(e[0], e[1]);
}
template <>
__device__ __forceinline__ void dft_regs<2>(fr (&e)[4], const fr (&w8)[3]) {
    bfly(e[0], e[2]);
    bfly(e[1], e[3]);
    fr_mul_lazy(e[3], e[3], w8[1]);
    bfly(e[0], e[1]);  // y0, y2
    bfly(e[2], e[3]);  // y1, y3
    fr t = e[1];
    e[1] = e[2];
    e[2] = t;
}
template <>
__device__ __forceinline__ void dft_regs<3>(fr (&e)[8], const fr (&w8)[3]) {
    bfly(e[0], e[4]);
    bfly(e[1], e[5]);
    bfly(e[2], e[6]);
    bfly(e[3], e[7]);
    fr_mul_lazy(e[5], e[5], w8[0]);
    fr_mul_lazy(e[6], e[6], w8[1]);
    fr_mul_lazy(e[7], e[7], w8[2]);
    // two size-4 DFTs: e[0..3] -> even outputs, e[4..7] -> odd outputs
    static_for<0, 2>([&](auto hc) {
        constexpr int h = 4 * decltype(hc)::value;
        bfly(e[h + 0], e[h + 2]);
        bfly(e[h + 1], e[h + 3]);
        fr_mul_lazy(e[h + 3], e[h + 3], w8[1]);
        bfly(e[h + 0], e[h + 1]);  // z0, z2
        bfly(e[h + 2], e[h + 3]);  // z1, z3
    });
    // registers now hold (y0, y4, y2, y6, y1, y5, y3, y7)
    fr y1 = e[4], y2 = e[2], y3 = e[6], y4 = e[1], y5 = e[5], y6 = e[3];
    e[1] = y1; e[2] = y2; e[3] = y3; e[4] = y4; e[5] = y5; e[6] = y6;
}

// Radix plan: the remainder pass (radix 2 or 4) goes first, all later passes are radix 8.
template <int LOGK>
struct NttPlan {
    static constexpr int kRem = LOGK % 3;
    static constexpr int kFirstLogR = (LOGK < 3) ? LOGK : (kRem ? kRem : 3);
    static constexpr int kThreadsPerNtt = (LOGK <= 3) ? 1 : (1 << (LOGK - 3));
    static constexpr int kWgThreads = (kThreadsPerNtt > 256) ? kThreadsPerNtt : 256;
    static constexpr int kNttsPerWg = kWgThreads / kThreadsPerNtt;
    static constexpr int kLdsBytes = kNttsPerWg * lds_slots(1 << LOGK) * 32;
};

// digit reversal of the in-place DIF with the plan above: natural output index j -> LDS position
template <int LOGK>
__device__ __forceinline__ int dif_position(int j) {
    int pos = 0;
    int logs = LOGK;
    int logr = NttPlan<LOGK>::kFirstLogR;
#pragma unroll
    for (int guard = 0; guard < 6; guard++) {
        if (logs <= 0) break;
        logs -= logr;
        pos += (j & ((1 << logr) - 1)) << logs;
        j >>= logr;
        logr = 3;
    }
    return pos;
}

// One DIF pass over an LDS-resident row.  LOGS = log2 of the current sub-transform size.
// Launch-invariant operands, copied out of the kernel argument block once so that they stay
// in scalar registers (taking references into the by-value argument struct would force a
// private-memory copy of it).
struct NttConsts {
    const fr* tw;
    fr w8[3];
};

template <int LOGK, int LOGS, int LOGR, bool FIRST, bool EVALUATE>
__device__ __forceinline__ void dif_pass(const LdsRow& row, int t, bool active, const NttConsts& a,
                                         const fr* __restrict__ gin, const fr* __restrict__ pre_tw, uint32_t coset,
                                         fr* __restrict__ canon_out) {
    constexpr int K = 1 << LOGK;
    constexpr int R = 1 << LOGR;
    constexpr int LOGSUB = LOGS - LOGR;
    constexpr int SUB = 1 << LOGSUB;
    constexpr int T = NttPlan<LOGK>::kThreadsPerNtt;
    if (!active) return;
    for (int u = t; u < (K >> LOGR); u += T) {
        const int blk = u >> LOGSUB;
        const int i0 = u & (SUB - 1);
        const int base = (blk << LOGS) + i0;
        fr e[R];
        if constexpr (FIRST) {
            static_for<0, R>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int d = base + (q << LOGSUB);
                e[q] = fr_load(gin + d);
                if constexpr (EVALUATE) {
                    fr w = fr_load(pre_tw + (size_t)coset * d);
                    fr_mul_lazy(e[q], e[q], w);
                } else {
                    if (canon_out != nullptr) {
                        fr c;
                        fr_from_mont(c, e[q]);
                        fr_store(canon_out + d, c);
                    }
                }
            });
        } else {
            static_for<0, R>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                e[q] = row.get(base + (q << LOGSUB));
            });
        }
        dft_regs<LOGR>(e, a.w8);
        if constexpr (LOGSUB > 0) {
            static_for<1, R>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                fr w = fr_load(a.tw + ((size_t)(i0 * m) << (LOGK - LOGS)));
                fr_mul_lazy(e[m], e[m], w);
            });
        }
        static_for<0, R>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            row.put(base + (m << LOGSUB), e[m]);
        });
    }
}

template <int LOGK, int LOGS, bool EVALUATE>
__device__ __forceinline__ void dif_rest(const LdsRow& row, int t, bool active, const NttConsts& a) {
    if constexpr (LOGS > 0) {
        __syncthreads();
        dif_pass<LOGK, LOGS, 3, false, EVALUATE>(row, t, active, a, nullptr, nullptr, 0, nullptr);
        dif_rest<LOGK, LOGS - 3, EVALUATE>(row, t, active, a);
    }
}

// grid: ceil(work / kNttsPerWg) workgroups; work = rows (interpolate) or rows * ncos (evaluate)
template <int LOGK, bool EVALUATE>
__global__ void __launch_bounds__(NttPlan<LOGK>::kWgThreads) ntt_rows_kernel(const NttArgs a) {
    using Plan = NttPlan<LOGK>;
    constexpr int K = 1 << LOGK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int slot = threadIdx.x / Plan::kThreadsPerNtt;
    const int t = threadIdx.x % Plan::kThreadsPerNtt;
    const uint32_t total = EVALUATE ? a.rows * a.ncos : a.rows;
    const uint32_t w = blockIdx.x * Plan::kNttsPerWg + slot;
    const bool active = w < total;
    uint32_t r = 0, coset = 0;
    if (active) {
        if constexpr (EVALUATE) {
            r = w / a.ncos;
            coset = a.cosets[w % a.ncos];
        } else {
            r = w;
        }
    }
    const size_t row_off = (size_t)(a.row0 + r) * K;
    LdsRow row;
    row.lo = reinterpret_cast<uint4*>(smem) + (size_t)slot * 2 * lds_slots(K);
    row.hi = row.lo + lds_slots(K);

    NttConsts cs;
    cs.tw = a.tw;
    cs.w8[0] = a.w8[0];
    cs.w8[1] = a.w8[1];
    cs.w8[2] = a.w8[2];
    fr* canon = (!EVALUATE && a.canon_out != nullptr) ? a.canon_out + row_off : nullptr;
    dif_pass<LOGK, LOGK, Plan::kFirstLogR, true, EVALUATE>(row, t, active, cs, a.in + row_off, a.coset_tw, coset, canon);
    dif_rest<LOGK, LOGK - Plan::kFirstLogR, EVALUATE>(row, t, active, cs);
    __syncthreads();
    if (!active) return;
    fr* gout = EVALUATE ? a.out + (size_t)coset * a.plane_stride + row_off : a.out + row_off;
    for (int j = t; j < K; j += Plan::kThreadsPerNtt) {
        fr x = row.get(dif_position<LOGK>(j));
        if constexpr (!EVALUATE) {
            const fr scale = a.scale;
            fr_mul_lazy(x, x, scale);
        }
        fr y;
        fr_reduce(y, x);
        fr_store(gout + j, y);
    }
}

}  // namespace lg
