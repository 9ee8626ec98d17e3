// Round-3 bounded experiments on the constant products of the transforms, gfx950, wall ns per wave-level
// product per SIMD, with a correctness leg against the kernels' own shoup29 / mul29_dot on random operands:
//   (a) wshift29: product by a LAUNCH-INVARIANT constant w (the omega_8 powers of every radix-8 butterfly) as
//       sum_i a_i * (w 2^(29 i) mod p), the nine pre-shifted residues in scalar registers (kernel arguments): 81 multiplier
//       instructions for the sum + 4 for the quotient of its 57-bit top + 18 for q * (2^261 - p) = 103 instead of shoup29's 143
//   (b) kara29_dot<2>: mul29_dot<2> with each full 9 x 9 product done as a 3-way Karatsuba over 3-limb blocks: 54 multiplier
//       instructions per product instead of 81, paid for with 64-bit column additions / subtractions
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fr29_gfx950.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace lg;

struct WShift {
    uint32_t r[9][9];   // r[i] = w * 2^(29 i) mod p, 29-bit limbs
    uint32_t m_lo, m_hi;   // floor(2^296 / p)
};

// a dirty (limbs 0..7 <= 6 * 2^29, a.v[8] < 2^29): r = a * w mod p (+ p), limbs 0..7 < 2^29
__device__ __forceinline__ void wshift29(f29& out, const f29& a, const WShift& W) {
    uint64_t col[9];
#pragma unroll
    for (int c = 0; c < 9; c++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) acc = mad29s(a.v[i], W.r[i][c], acc);
        col[c] = acc;
    }
    uint32_t lo[8];
    uint64_t carry = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const uint64_t t = col[c] + carry;
        lo[c] = (uint32_t)t & kM29;
        carry = t >> 29;
    }
    const uint64_t T = col[8] + carry;                 // everything from bit 232 up: < 2^57
    // q = floor(T * M / 2^64), M = floor(2^296 / p) < 2^43: floor(t / p) or one less
    const uint32_t t0 = (uint32_t)T, t1 = (uint32_t)(T >> 32);
    uint64_t mid = ((uint64_t)t0 * W.m_lo) >> 32;
    mid += (uint64_t)t1 * W.m_lo;
    mid += (uint64_t)t0 * W.m_hi;
    const uint64_t q = (uint64_t)t1 * W.m_hi + (mid >> 32);
    const uint32_t q0 = (uint32_t)q & kM29, q1 = (uint32_t)(q >> 29);
    // low 261 bits of t + q * (2^261 - p)
    uint64_t acc = 0;
#pragma unroll
    for (int c = 0; c < 9; c++) {
        acc += (c < 8) ? (uint64_t)lo[c] : (T & kM29);
        acc = mad29s(q0, np29(c), acc);
        if (c > 0) acc = mad29s(q1, np29(c - 1), acc);
        if (c == 8) keep64(acc);
        out.v[c] = (uint32_t)acc & kM29;
        acc >>= 29;
    }
}

// 5 columns of a 3 x 3 limb block product
__device__ __forceinline__ void blk3(uint64_t (&o)[5], const uint32_t* x, const uint32_t* y) {
    o[0] = (uint64_t)x[0] * y[0];
    o[1] = mad29(x[0], y[1], (uint64_t)x[1] * y[0]);
    o[2] = mad29(x[0], y[2], mad29(x[1], y[1], (uint64_t)x[2] * y[0]));
    o[3] = mad29(x[1], y[2], (uint64_t)x[2] * y[1]);
    o[4] = (uint64_t)x[2] * y[2];
}
// the 17 columns of a * b (both normalised, limbs < 2^29) added to P
__device__ __forceinline__ void kara_cols(uint64_t (&P)[17], const f29& a, const f29& b) {
    uint32_t sa[3][3], sb[3][3];   // A0+A1, A0+A2, A1+A2
#pragma unroll
    for (int i = 0; i < 3; i++) {
        sa[0][i] = a.v[i] + a.v[3 + i]; sa[1][i] = a.v[i] + a.v[6 + i]; sa[2][i] = a.v[3 + i] + a.v[6 + i];
        sb[0][i] = b.v[i] + b.v[3 + i]; sb[1][i] = b.v[i] + b.v[6 + i]; sb[2][i] = b.v[3 + i] + b.v[6 + i];
    }
    uint64_t p00[5], p11[5], p22[5], s01[5], s02[5], s12[5];
    blk3(p00, &a.v[0], &b.v[0]); blk3(p11, &a.v[3], &b.v[3]); blk3(p22, &a.v[6], &b.v[6]);
    blk3(s01, sa[0], sb[0]); blk3(s02, sa[1], sb[1]); blk3(s12, sa[2], sb[2]);
#pragma unroll
    for (int i = 0; i < 5; i++) {
        P[i] += p00[i];
        P[3 + i] += s01[i] - p00[i] - p11[i];
        P[6 + i] += s02[i] - p00[i] - p22[i] + p11[i];
        P[9 + i] += s12[i] - p11[i] - p22[i];
        P[12 + i] += p22[i];
    }
}
template <int O>
__device__ __forceinline__ void kara29_dot(f29& r, const f29 (&a)[O], const f29 (&b)[O]) {
    uint64_t P[17];
#pragma unroll
    for (int i = 0; i < 17; i++) P[i] = 0;
#pragma unroll
    for (int o = 0; o < O; o++) kara_cols(P, a[o], b[o]);
    uint64_t acc = 0;
    uint32_t q[9];
#pragma unroll
    for (int c = 0; c < 9; c++) {
        acc += P[c];
#pragma unroll
        for (int i = 0; i < c; i++) acc = mad29s(q[i], p29(c - i), acc);
        q[c] = ((uint32_t)acc * kPinv29) & kM29;
        acc = mad29s(q[c], p29(0), acc);
        acc >>= 29;
    }
#pragma unroll
    for (int c = 9; c < 17; c++) {
        acc += P[c];
#pragma unroll
        for (int i = c - 8; i <= 8; i++) acc = mad29s(q[i], p29(c - i), acc);
        r.v[c - 9] = (uint32_t)acc & kM29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
}

// MODE 0 shoup29, 1 wshift29, 2 mul29_dot<2>, 3 kara29_dot<2>
template <int U, int MODE>
__global__ void kern(uint32_t* io, int iters, WShift W) {
    const size_t base = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * U * 9;
    f29 x[U], w, wq;
#pragma unroll
    for (int i = 0; i < 9; i++) { w.v[i] = W.r[0][i]; wq.v[i] = io[9 + i] & kM29; }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int i = 0; i < 9; i++) x[u].v[i] = io[base + u * 9 + i] & kM29;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if constexpr (MODE == 0) shoup29(x[u], x[u], w, wq);
            else if constexpr (MODE == 1) wshift29(x[u], x[u], W);
            else {
                f29 aa[2] = {x[u], x[(u + 1) % U]}, bb[2] = {w, wq};
                if constexpr (MODE == 2) mul29_dot<2>(x[u], aa, bb);
                else kara29_dot<2>(x[u], aa, bb);
                x[u].v[8] &= kM29;   // (keeps the loop's operands inside the normalised range; same cost in both variants)
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int i = 0; i < 9; i++) io[base + u * 9 + i] = x[u].v[i];
}

// correctness: out[t] = (shoup, wshift, dot, kara) results, fully reduced
__global__ void check_kernel(const uint32_t* in, uint32_t* out, int n, WShift W, f29 w, f29 wq, f29 b1) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    f29 a, a2, r0, r1, r2, r3;
    for (int i = 0; i < 9; i++) { a.v[i] = in[t * 18 + i]; a2.v[i] = in[t * 18 + 9 + i] & kM29; }
    a2.v[8] &= 0x3fffff;
    shoup29(r0, a, w, wq);
    wshift29(r1, a, W);
    f29 an = a;
    for (int i = 0; i < 9; i++) an.v[i] &= kM29;
    f29 aa[2] = {an, a2}, bb[2] = {w, b1};
    mul29_dot<2>(r2, aa, bb);
    kara29_dot<2>(r3, aa, bb);
    const fr f0 = pack29_reduced(r0), f1 = pack29_reduced(r1), f2 = pack29_reduced(r2), f3 = pack29_reduced(r3);
    uint32_t bad = 0;
    for (int i = 0; i < 8; i++) bad |= (f0.v[i] ^ f1.v[i]) | (f2.v[i] ^ f3.v[i]);
    uint32_t top1 = r1.v[8];
    out[t * 2] = bad;
    out[t * 2 + 1] = top1;   // limb 8 of wshift29's result: < 2^23 means value < 2p-ish (checked on the host)
}

template <int U, int MODE>
int run(const char* name, int cus, uint32_t* buf, const WShift& W) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-24s", name);
    for (int Wv : {1, 2, 4}) {
        const int grid = cus * Wv, iters = 4000 / Wv / U;
        hipLaunchKernelGGL((kern<U, MODE>), dim3(grid), dim3(256), 0, 0, buf, 10, W);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((kern<U, MODE>), dim3(grid), dim3(256), 0, 0, buf, iters, W);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double ops = (double)iters * U * Wv;
        printf(" | W=%d %7.1f ns/op", Wv, ms * 1e6 / ops);
    }
    printf("\n");
    return 0;
}

// ---- host big-number helpers (256-bit + headroom as 5 x u64) for the constants
typedef unsigned __int128 u128;
struct Big { uint64_t l[6]; };
static const uint64_t PL[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static bool geq(const Big& a, const Big& b) { for (int i = 5; i >= 0; i--) { if (a.l[i] != b.l[i]) return a.l[i] > b.l[i]; } return true; }
static void sub(Big& a, const Big& b) { u128 br = 0; for (int i = 0; i < 6; i++) { u128 t = (u128)a.l[i] - b.l[i] - br; a.l[i] = (uint64_t)t; br = (t >> 64) & 1; } }
static void shl1(Big& a) { for (int i = 5; i > 0; i--) a.l[i] = (a.l[i] << 1) | (a.l[i - 1] >> 63); a.l[0] <<= 1; }
static Big P() { Big p{}; for (int i = 0; i < 4; i++) p.l[i] = PL[i]; return p; }
static void mod_shl(Big& a, int bits) { const Big p = P(); for (int i = 0; i < bits; i++) { shl1(a); if (geq(a, p)) sub(a, p); } }
static void to29(const Big& a, uint32_t* out) { for (int i = 0; i < 9; i++) { const int bit = 29 * i, w = bit >> 6, sh = bit & 63; uint64_t x = a.l[w] >> sh; if (sh > 35) x |= a.l[w + 1] << (64 - sh); out[i] = (uint32_t)(x & (i < 8 ? kM29 : 0xffffffffu)); } }

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // a constant w < p and its tables
    Big w{};
    w.l[0] = 0x123456789abcdef1ull; w.l[1] = 0x0fedcba987654321ull; w.l[2] = 0x1122334455667788ull; w.l[3] = 0x1064a1b2c3d4e5f6ull;
    WShift W;
    Big s = w;
    for (int i = 0; i < 9; i++) { to29(s, W.r[i]); mod_shl(s, 29); }
    {   // floor(2^296 / p) by long division
        Big rem{}; uint64_t q = 0;
        rem.l[0] = 1;
        const Big p = P();
        // 2^296 / p: shift-subtract 296 bits
        Big acc{};
        for (int bit = 296; bit >= 0; bit--) { shl1(acc); if (bit == 296) acc.l[0] |= 1; if (geq(acc, p)) { sub(acc, p); if (bit < 64) q |= 1ull << bit; } }
        W.m_lo = (uint32_t)q; W.m_hi = (uint32_t)(q >> 32);
        (void)rem;
    }
    f29 wv, wq, b1;
    for (int i = 0; i < 9; i++) wv.v[i] = W.r[0][i];
    {   // wq = floor(w 2^261 / p)
        Big acc = w; const Big p = P(); Big q{};
        for (int bit = 260; bit >= 0; bit--) { shl1(acc); if (geq(acc, p)) { sub(acc, p); q.l[bit >> 6] |= 1ull << (bit & 63); } }
        to29(q, wq.v);
        for (int i = 0; i < 9; i++) b1.v[i] = (W.r[3][i]);
    }
    // correctness leg
    const int n = 1 << 16;
    std::vector<uint32_t> in((size_t)n * 18), out((size_t)n * 2);
    srand(7);
    for (int t = 0; t < n; t++) {
        for (int i = 0; i < 8; i++) in[t * 18 + i] = (t & 1) ? (uint32_t)(((uint64_t)rand() * 2654435761u) % (6u << 29)) : ((uint32_t)rand() * 2654435761u) & kM29;
        in[t * 18 + 8] = ((uint32_t)rand() * 2654435761u) & ((t & 2) ? 0x0fffffffu : 0x3fffff);
        for (int i = 9; i < 18; i++) in[t * 18 + i] = (uint32_t)rand() * 2654435761u;
    }
    uint32_t *din, *dout;
    CK(hipMalloc((void**)&din, in.size() * 4));
    CK(hipMalloc((void**)&dout, out.size() * 4));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, din, dout, n, W, wv, wq, b1);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; uint32_t top = 0;
    for (int t = 0; t < n; t++) { bad += out[t * 2] != 0; if (out[t * 2 + 1] > top) top = out[t * 2 + 1]; }
    printf("correctness: %d of %d operand sets differ (wshift29 vs shoup29, kara29_dot<2> vs mul29_dot<2>); largest top limb of wshift29 %#x (p's is 0x30644e)\n", bad, n, top);
    uint32_t* buf;
    const size_t bytes = (size_t)cus * 4 * 256 * 4 * 9 * 4 + 1024;
    CK(hipMalloc((void**)&buf, bytes));
    CK(hipMemset(buf, 0x5a, bytes));
    printf("wall ns per wave-level op per SIMD (W = waves per SIMD)\n");
    run<1, 0>("shoup29        U=1", cus, buf, W);
    run<4, 0>("shoup29        U=4", cus, buf, W);
    run<1, 1>("wshift29       U=1", cus, buf, W);
    run<4, 1>("wshift29       U=4", cus, buf, W);
    run<2, 2>("mul29_dot<2>   U=2", cus, buf, W);
    run<2, 3>("kara29_dot<2>  U=2", cus, buf, W);
    return bad ? 3 : 0;
}
