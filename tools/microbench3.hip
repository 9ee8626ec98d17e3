// Throughput of candidate Montgomery multipliers on gfx950 (wall ns per wave-mulmod per SIMD):
//   sat32  : 8 x 32-bit saturated limbs, CIOS (fr_gfx950.h fr_mul_lazy)
//   uns29  : 9 x 29-bit unsaturated limbs, product scanning, R = 2^261 (no carry chains)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "fr_gfx950.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct f29 { uint32_t v[9]; };
__device__ __forceinline__ constexpr uint32_t p29(int i) {
    constexpr uint32_t P[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
    return P[i];
}
#define M29 0x1fffffffu
__device__ __forceinline__ void mul29(f29& r, const f29& a, const f29& b) {
    uint64_t acc = 0;
    uint32_t q[9];
#pragma unroll
    for (int c = 0; c < 9; c++) {
#pragma unroll
        for (int i = 0; i <= c; i++) acc += (uint64_t)a.v[i] * b.v[c - i];
#pragma unroll
        for (int i = 0; i < c; i++) acc += (uint64_t)q[i] * p29(c - i);
        q[c] = ((uint32_t)acc * 0x0fffffffu) & M29;
        acc += (uint64_t)q[c] * p29(0);
        acc >>= 29;
    }
#pragma unroll
    for (int c = 9; c < 17; c++) {
#pragma unroll
        for (int i = c - 8; i <= 8; i++) acc += (uint64_t)a.v[i] * b.v[c - i];
#pragma unroll
        for (int i = c - 8; i <= 8; i++) acc += (uint64_t)q[i] * p29(c - i);
        r.v[c - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
}
// lazy butterfly in unsaturated form: plain limb-wise adds, biased subtract, one normalisation
__device__ __forceinline__ void bfly29(f29& a, f29& b, const uint32_t (&bias)[9]) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint32_t s = a.v[i] + b.v[i];
        uint32_t d = a.v[i] - b.v[i] + bias[i];
        a.v[i] = s;
        b.v[i] = d;
    }
}
__device__ __forceinline__ void norm29(f29& a) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t t = a.v[i] + c;
        a.v[i] = t & M29;
        c = t >> 29;
    }
    a.v[8] += c;
}

template <int U, int MODE>
__global__ void kern(uint32_t* io, int iters, long long* cyc) {
    const size_t base = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * U * 9;
    long long t0, t1;
    if constexpr (MODE == 0) {
        lg::fr x[U], w;
#pragma unroll
        for (int i = 0; i < 8; i++) w.v[i] = io[i] | 1;
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) x[u].v[i] = io[base + u * 9 + i];
        t0 = clock64();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < U; u++) lg::fr_mul_lazy(x[u], x[u], w);
        }
        t1 = clock64();
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) io[base + u * 9 + i] = x[u].v[i];
    } else if constexpr (MODE == 1) {
        f29 x[U], w;
#pragma unroll
        for (int i = 0; i < 9; i++) w.v[i] = io[i] & M29;
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < 9; i++) x[u].v[i] = io[base + u * 9 + i] & M29;
        t0 = clock64();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < U; u++) mul29(x[u], x[u], w);
        }
        t1 = clock64();
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < 9; i++) io[base + u * 9 + i] = x[u].v[i];
    } else {  // butterfly (add + biased sub) + one norm each: the non-multiply part of a radix-2 step
        f29 x[2];
        uint32_t bias[9];
#pragma unroll
        for (int i = 0; i < 9; i++) { bias[i] = io[i] | 0x20000000u; x[0].v[i] = io[base + i] & M29; x[1].v[i] = io[base + 9 + i] & M29; }
        t0 = clock64();
        for (int it = 0; it < iters; it++) {
            bfly29(x[0], x[1], bias);
            norm29(x[0]);
            norm29(x[1]);
        }
        t1 = clock64();
#pragma unroll
        for (int i = 0; i < 9; i++) { io[base + i] = x[0].v[i]; io[base + 9 + i] = x[1].v[i]; }
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int U, int MODE>
int run(const char* name, int cus, uint32_t* buf, long long* dcyc) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-22s", name);
    for (int W : {1, 2, 4}) {
        const int grid = cus * W, iters = 4000 / W / U;
        hipLaunchKernelGGL((kern<U, MODE>), dim3(grid), dim3(256), 0, 0, buf, 10, dcyc);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((kern<U, MODE>), dim3(grid), dim3(256), 0, 0, buf, iters, dcyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double ops = (double)iters * U * W;  // wave-level ops per SIMD
        printf(" | W=%d %7.1f ns/op (%6.1f G lane-op/s chip)", W, ms * 1e6 / ops, cus * 4.0 * 64 * ops / (ms * 1e6));
    }
    printf("\n");
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* buf;
    long long* dcyc;
    CK(hipMalloc((void**)&buf, (size_t)cus * 4 * 256 * 4 * 9 * 4 + 1024));
    CK(hipMemset(buf, 0x5a, (size_t)cus * 4 * 256 * 4 * 9 * 4 + 1024));
    CK(hipMalloc((void**)&dcyc, (size_t)cus * 8 * 8));
    printf("wall ns per wave-level op per SIMD; chip rate in lane-ops\n");
    run<1, 0>("sat32 mulmod U=1", cus, buf, dcyc);
    run<2, 0>("sat32 mulmod U=2", cus, buf, dcyc);
    run<4, 0>("sat32 mulmod U=4", cus, buf, dcyc);
    run<1, 1>("uns29 mulmod U=1", cus, buf, dcyc);
    run<2, 1>("uns29 mulmod U=2", cus, buf, dcyc);
    run<4, 1>("uns29 mulmod U=4", cus, buf, dcyc);
    run<1, 2>("uns29 bfly+2norm", cus, buf, dcyc);
    return 0;
}
