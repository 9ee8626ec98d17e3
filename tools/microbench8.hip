// Does a wave64 instruction whose upper 32 lanes are all inactive issue in one pass instead of two on gfx950?  (the
// four-lanes-per-column Blake2s is a latency chain of dependent 32-bit operations; if half-empty waves issued twice as fast, eight
// columns per wave instead of sixteen would halve the time per block.)  One wave per SIMD runs a dependent chain of the hash's
// instruction kinds with 64, 32 (lower half) and 16 active lanes.
//   hipcc -O3 --offload-arch=gfx950 -o microbench8 tools/microbench8.hip && ./microbench8
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

__device__ __forceinline__ uint32_t rotr(uint32_t x, int r) { return __builtin_amdgcn_alignbit(x, x, r); }

// the G function's dependent chain, `iters` times
template <int ACTIVE>
__global__ void __launch_bounds__(64) chain_kernel(uint32_t* out, int iters, uint32_t seed) {
    const int lane = threadIdx.x;
    uint32_t a = seed + lane, b = seed * 3 + lane, c = seed ^ lane, d = seed + 7 * lane, m0 = lane * 2654435761u, m1 = ~m0;
    if (lane < ACTIVE) {
#pragma unroll 1
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                a = a + b + m0; d = rotr(d ^ a, 16);
                c = c + d;      b = rotr(b ^ c, 12);
                a = a + b + m1; d = rotr(d ^ a, 8);
                c = c + d;      b = rotr(b ^ c, 7);
            }
        }
    }
    out[blockIdx.x * 64 + lane] = a ^ b ^ c ^ d;
}

template <int ACTIVE>
static int run(const char* name, uint32_t* d_out, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 20000;
    hipLaunchKernelGGL((chain_kernel<ACTIVE>), dim3(blocks), dim3(64), 0, 0, d_out, 100, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((chain_kernel<ACTIVE>), dim3(blocks), dim3(64), 0, 0, d_out, iters, 1u);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)iters * 8 * 12;   // dependent instructions per wave (v_add3, v_xor, v_alignbit)
    printf("%-34s %4d waves: %8.3f ms  %6.2f ns per dependent instruction\n", name, blocks, ms, ms * 1e6 / ops);
    return 0;
}

int main() {
    uint32_t* d_out = nullptr;
    CK(hipMalloc(reinterpret_cast<void**>(&d_out), 4096 * 64 * sizeof(uint32_t)));
    for (int blocks : {256, 1024, 4096}) {   // 1, 4 and 16 waves per CU (one wave per workgroup)
        if (run<64>("64 active lanes", d_out, blocks)) return 1;
        if (run<32>("32 active lanes (lower half)", d_out, blocks)) return 1;
        if (run<16>("16 active lanes", d_out, blocks)) return 1;
    }
    return 0;
}
