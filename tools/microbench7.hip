// Store-pattern microbenchmark: HBM write bandwidth of the ways a wave can put 32-byte field elements into a
// row.  The row kernels store one element per lane as two 16-byte stores (each instruction covers every other 16 bytes of a
// 2 KB span); the alternatives make every store instruction cover a contiguous 1 KB.
//   hipcc -O3 --offload-arch=gfx950 -o microbench7 tools/microbench7.hip && ./microbench7
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

// MODE 0: 16 B per lane, contiguous per instruction            (reference: what a plain copy does)
// MODE 1: 32 B per lane as two 16-B stores to its own element  (the row kernels' pattern)
// MODE 2: 32 B per lane, but instruction 0 writes chunk L of the wave's first 1 KB and instruction 1 chunk L of the second
template <int MODE, bool NT>
__global__ void __launch_bounds__(256) write_kernel(u32x4* out, size_t n16, uint32_t seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 v;
    v.x = seed + (uint32_t)tid; v.y = seed ^ 0x9e3779b9u; v.z = (uint32_t)tid * 2654435761u; v.w = seed;
    if constexpr (MODE == 0) {
        for (size_t i = tid; i < n16; i += stride) {
            if constexpr (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
            v.x += 1;
        }
    } else if constexpr (MODE == 1) {
        for (size_t e = tid; 2 * e + 1 < n16; e += stride) {
            if constexpr (NT) { __builtin_nontemporal_store(v, out + 2 * e); __builtin_nontemporal_store(v, out + 2 * e + 1); }
            else { out[2 * e] = v; out[2 * e + 1] = v; }
            v.x += 1;
        }
    } else {
        const size_t lane = tid & 63, wave = tid >> 6, nwaves = stride >> 6;
        for (size_t w = wave; (w + 1) * 128 <= n16; w += nwaves) {
            u32x4* base = out + w * 128;
            if constexpr (NT) { __builtin_nontemporal_store(v, base + lane); __builtin_nontemporal_store(v, base + 64 + lane); }
            else { base[lane] = v; base[64 + lane] = v; }
            v.x += 1;
        }
    }
}

// the same three patterns as a copy (read one element per lane the way the kernels do, write it back elsewhere)
template <int MODE, bool NT>
__global__ void __launch_bounds__(256) copy_kernel(const u32x4* in, u32x4* out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (MODE == 0) {
        for (size_t i = tid; i < n16; i += stride) {
            const u32x4 v = NT ? __builtin_nontemporal_load(in + i) : in[i];
            if constexpr (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
        }
    } else if constexpr (MODE == 1) {
        for (size_t e = tid; 2 * e + 1 < n16; e += stride) {
            const u32x4 a = NT ? __builtin_nontemporal_load(in + 2 * e) : in[2 * e];
            const u32x4 b = NT ? __builtin_nontemporal_load(in + 2 * e + 1) : in[2 * e + 1];
            if constexpr (NT) { __builtin_nontemporal_store(a, out + 2 * e); __builtin_nontemporal_store(b, out + 2 * e + 1); }
            else { out[2 * e] = a; out[2 * e + 1] = b; }
        }
    } else {
        const size_t lane = tid & 63, wave = tid >> 6, nwaves = stride >> 6;
        for (size_t w = wave; (w + 1) * 128 <= n16; w += nwaves) {
            const u32x4* src = in + w * 128;
            u32x4* base = out + w * 128;
            const u32x4 a = NT ? __builtin_nontemporal_load(src + lane) : src[lane];
            const u32x4 b = NT ? __builtin_nontemporal_load(src + 64 + lane) : src[64 + lane];
            if constexpr (NT) { __builtin_nontemporal_store(a, base + lane); __builtin_nontemporal_store(b, base + 64 + lane); }
            else { base[lane] = a; base[64 + lane] = b; }
        }
    }
}

template <typename F>
static int time_it(const char* name, size_t bytes, F launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-58s %8.3f ms  %8.1f GB/s\n", name, ms / reps, (double)bytes * reps / (ms * 1e-3) / 1e9);
    return 0;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)2 << 30;
    const size_t n16 = bytes / 16;
    u32x4 *a = nullptr, *b = nullptr;
    CK(hipMalloc(reinterpret_cast<void**>(&a), bytes));
    CK(hipMalloc(reinterpret_cast<void**>(&b), bytes));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    for (int blocks_per_cu : {2, 8}) {
        const int grid = 256 * blocks_per_cu;
        printf("-- %d workgroups of 256 threads per CU\n", blocks_per_cu);
#define W(MODE, NT, label) \
        if (time_it("write  " label, bytes, [&] { hipLaunchKernelGGL((write_kernel<MODE, NT>), dim3(grid), dim3(256), 0, 0, a, n16, 7u); })) return 1;
#define C(MODE, NT, label) \
        if (time_it("copy   " label " (read + write bytes)", 2 * bytes, [&] { hipLaunchKernelGGL((copy_kernel<MODE, NT>), dim3(grid), dim3(256), 0, 0, a, b, n16); })) return 1;
        W(0, false, "16 B / lane contiguous, plain")
        W(0, true, "16 B / lane contiguous, nontemporal")
        W(1, false, "32-B element per lane (2 x 16 B), plain")
        W(1, true, "32-B element per lane (2 x 16 B), nontemporal")
        W(2, false, "32 B / lane as two contiguous 1 KB instr, plain")
        W(2, true, "32 B / lane as two contiguous 1 KB instr, nontemporal")
        C(0, false, "16 B / lane contiguous, plain")
        C(0, true, "16 B / lane contiguous, nontemporal")
        C(1, false, "32-B element per lane, plain")
        C(1, true, "32-B element per lane, nontemporal")
        C(2, false, "two contiguous 1 KB instr, plain")
        C(2, true, "two contiguous 1 KB instr, nontemporal")
    }
    (void)argc; (void)argv;
    return 0;
}
