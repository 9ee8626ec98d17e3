// Micro-benchmarks that set the VALU roofline of the field arithmetic on gfx950:
// issue cost (cycles per wave-instruction per SIMD) of v_mad_u64_u32, 32-bit adds,
// v_lshl_add_u64, and the throughput of fr_mul_lazy at several occupancies.
// Build: hipcc -O3 --offload-arch=gfx950 -I ligero_amd/csrc -o build/microbench tools/microbench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "fr_gfx950.h"
using namespace lg;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

template <int NACC>
__global__ void k_mad(uint64_t* out, uint32_t a0, uint32_t b0, int iters, long long* cyc) {
    uint64_t acc[NACC];
    uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = j + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) acc[j] = (uint64_t)a * (uint32_t)(b + j) + acc[j];
    }
    long long t1 = clock64();
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < NACC; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
__global__ void k_add(uint32_t* out, uint32_t a0, int iters, long long* cyc) {
    uint32_t acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = j + threadIdx.x;
    uint32_t a = a0 + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[j]) : "v"(a));
    }
    long long t1 = clock64();
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < NACC; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
__global__ void k_addc(uint32_t* out, uint32_t a0, int iters, long long* cyc) {
    uint32_t acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = j + threadIdx.x;
    uint32_t a = a0 + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(acc[j]) : "v"(a) : "vcc");
    }
    long long t1 = clock64();
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < NACC; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
__global__ void k_add64(uint64_t* out, uint32_t a0, int iters, long long* cyc) {
    uint64_t acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = j + threadIdx.x;
    uint64_t a = a0 + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[j]) : "v"(a));
    }
    long long t1 = clock64();
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < NACC; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
__global__ void k_mullo(uint32_t* out, uint32_t a0, int iters, long long* cyc) {
    uint32_t acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; j++) acc[j] = j + threadIdx.x + 3;
    uint32_t a = a0 + threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < NACC; j++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[j]) : "v"(a));
    }
    long long t1 = clock64();
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < NACC; j++) s ^= acc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// U independent Montgomery products per thread per iteration
template <int U>
__global__ void k_mulmod(fr* io, const fr* w, int iters, long long* cyc) {
    fr x[U], ww = fr_load(w + threadIdx.x % 64);
#pragma unroll
    for (int u = 0; u < U; u++) x[u] = fr_load(io + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * U + u);
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < U; u++) fr_mul_lazy(x[u], x[u], ww);
    }
    long long t1 = clock64();
#pragma unroll
    for (int u = 0; u < U; u++) fr_store(io + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * U + u, x[u]);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int U>
__global__ void k_bfly(fr* io, int iters, long long* cyc) {
    fr x[2 * U];
#pragma unroll
    for (int u = 0; u < 2 * U; u++) x[u] = fr_load(io + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 2 * U + u);
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            fr s, d;
            fr_add_lazy(s, x[2 * u], x[2 * u + 1]);
            fr_sub_lazy(d, x[2 * u], x[2 * u + 1]);
            x[2 * u] = s;
            x[2 * u + 1] = d;
        }
    }
    long long t1 = clock64();
#pragma unroll
    for (int u = 0; u < 2 * U; u++) fr_store(io + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 2 * U + u, x[u]);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int cus = prop.multiProcessorCount;
    void* buf;
    CK(hipMalloc(&buf, (size_t)cus * 8 * 1024 * 8 * 32));
    CK(hipMemset(buf, 1, (size_t)cus * 8 * 1024 * 8 * 32));
    long long* dcyc;
    CK(hipMalloc((void**)&dcyc, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    // waves per SIMD: 1, 2, 4 (block = 256 threads = 1 wave per SIMD; blocks per CU = W)
    for (int W : {1, 2, 4}) {
        const int grid = cus * W;
        auto run = [&](const char* name, auto launch, double ops_per_thread_iter) {
            launch(grid);  // warm
            hipEventRecord(e0);
            launch(grid);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long cyc;
            hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
            // per SIMD: W waves each doing iters*ops wave-instructions in `cyc` cycles
            double wave_ops = (double)iters * ops_per_thread_iter * W;
            printf("  W=%d %-18s cycles/wave-op/SIMD = %7.2f   (kernel %.3f ms, wave0 %lld cyc, eff clock %.2f GHz)\n", W, name,
                   cyc / wave_ops, ms, cyc, cyc / (ms * 1e6));
        };
        run("v_mad_u64_u32 x8", [&](int g) { hipLaunchKernelGGL(k_mad<8>, dim3(g), dim3(256), 0, 0, (uint64_t*)buf, 12345u, 777u, iters, dcyc); }, 8);
        run("v_mul_lo_u32 x8", [&](int g) { hipLaunchKernelGGL(k_mullo<8>, dim3(g), dim3(256), 0, 0, (uint32_t*)buf, 12345u, iters, dcyc); }, 8);
        run("v_add_u32 x8", [&](int g) { hipLaunchKernelGGL(k_add<8>, dim3(g), dim3(256), 0, 0, (uint32_t*)buf, 12345u, iters, dcyc); }, 8);
        run("v_addc_co_u32 x8", [&](int g) { hipLaunchKernelGGL(k_addc<8>, dim3(g), dim3(256), 0, 0, (uint32_t*)buf, 12345u, iters, dcyc); }, 8);
        run("v_lshl_add_u64 x8", [&](int g) { hipLaunchKernelGGL(k_add64<8>, dim3(g), dim3(256), 0, 0, (uint64_t*)buf, 12345u, iters, dcyc); }, 8);
        run("fr_mul_lazy U=1", [&](int g) { hipLaunchKernelGGL(k_mulmod<1>, dim3(g), dim3(256), 0, 0, (fr*)buf, (const fr*)buf, iters / 4, dcyc); }, 0.25);
        run("fr_mul_lazy U=2", [&](int g) { hipLaunchKernelGGL(k_mulmod<2>, dim3(g), dim3(256), 0, 0, (fr*)buf, (const fr*)buf, iters / 4, dcyc); }, 0.5);
        run("fr_mul_lazy U=4", [&](int g) { hipLaunchKernelGGL(k_mulmod<4>, dim3(g), dim3(256), 0, 0, (fr*)buf, (const fr*)buf, iters / 4, dcyc); }, 1.0);
        run("bfly(add+sub) U=2", [&](int g) { hipLaunchKernelGGL(k_bfly<2>, dim3(g), dim3(256), 0, 0, (fr*)buf, iters / 4, dcyc); }, 0.5);
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
