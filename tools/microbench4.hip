// Constant-operand modular product candidates on gfx950 (wall ns per wave-product per SIMD):
//   mont29  : mul29 of fr29_gfx950.h (Montgomery, R' = 2^261): 164 mads + 9 mul_lo
//   shoup29 : Barrett with a precomputed quotient constant w' = floor(w 2^261 / p):
//             q = hi(a w') from columns 7..16 (53 mads), r = lo261(a w + q (2^261 - p)) (90 mads)
// Also a correctness leg: reads tools/mb4_vec.bin (N x {a[9], w[9], wq[9]}), writes tools/mb4_out.bin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fr29_gfx950.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace lg;

template <int U, int MODE>
__global__ void kern(uint32_t* io, int iters) {
    const size_t base = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * U * 9;
    f29 x[U], w, wq;
#pragma unroll
    for (int i = 0; i < 9; i++) { w.v[i] = io[i] & kM29; wq.v[i] = io[9 + i] & kM29; }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int i = 0; i < 9; i++) x[u].v[i] = io[base + u * 9 + i] & kM29;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if constexpr (MODE == 0) mul29(x[u], x[u], w);
            else shoup29(x[u], x[u], w, wq);
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int i = 0; i < 9; i++) io[base + u * 9 + i] = x[u].v[i];
}

__global__ void check_kernel(const uint32_t* in, uint32_t* out, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    f29 a, w, wq, r;
    for (int i = 0; i < 9; i++) { a.v[i] = in[t * 27 + i]; w.v[i] = in[t * 27 + 9 + i]; wq.v[i] = in[t * 27 + 18 + i]; }
    shoup29(r, a, w, wq);
    for (int i = 0; i < 9; i++) out[t * 9 + i] = r.v[i];
}

template <int U, int MODE>
int run(const char* name, int cus, uint32_t* buf) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-22s", name);
    for (int W : {1, 2, 4}) {
        const int grid = cus * W, iters = 4000 / W / U;
        hipLaunchKernelGGL((kern<U, MODE>), dim3(grid), dim3(256), 0, 0, buf, 10);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((kern<U, MODE>), dim3(grid), dim3(256), 0, 0, buf, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double ops = (double)iters * U * W;
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
    const size_t bytes = (size_t)cus * 4 * 256 * 4 * 9 * 4 + 1024;
    CK(hipMalloc((void**)&buf, bytes));
    CK(hipMemset(buf, 0x5a, bytes));
    FILE* f = fopen("tools/mb4_vec.bin", "rb");
    if (f) {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        const int n = (int)(sz / (27 * 4));
        std::vector<uint32_t> in((size_t)n * 27), out((size_t)n * 9);
        if (fread(in.data(), 4, in.size(), f) != in.size()) return 2;
        fclose(f);
        uint32_t *din, *dout;
        CK(hipMalloc((void**)&din, in.size() * 4));
        CK(hipMalloc((void**)&dout, out.size() * 4));
        CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(check_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, din, dout, n);
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        FILE* g = fopen("tools/mb4_out.bin", "wb");
        fwrite(out.data(), 4, out.size(), g);
        fclose(g);
        printf("checked %d vectors -> tools/mb4_out.bin\n", n);
    }
    printf("wall ns per wave-level op per SIMD; chip rate in lane-ops\n");
    run<1, 0>("mont29  U=1", cus, buf);
    run<4, 0>("mont29  U=4", cus, buf);
    run<1, 1>("shoup29 U=1", cus, buf);
    run<4, 1>("shoup29 U=4", cus, buf);
    return 0;
}
