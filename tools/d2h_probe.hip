// Device-to-host copies beside compute (round 4).  Which engine does the runtime use for a large copy into
// page-locked memory -- an SDMA engine or a shader (blit) kernel -- does that depend on what the copy waits for, how fast is
// each, and what does each do to an HBM-bound kernel running at the same time on another stream?  Compared with a copy kernel
// of our own with a small grid (the ship_kernel of ligero_amd/csrc/batch_prover.hip).
//   hipcc -O2 --offload-arch=gfx950 tools/d2h_probe.hip -o tools/d2h_probe     (run under rocprofv3 --kernel-trace to see blits)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) triad(const u32x4* a, const u32x4* b, u32x4* c, size_t n) {   // HBM bound: 2 reads + 1 write
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c[i] = a[i] + b[i];
}
__global__ void __launch_bounds__(256) ship(const u32x4* src, u32x4* dst, size_t n) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    size_t i = tid;
    for (; i + 3 * nt < n; i += 4 * nt) {
        const u32x4 v0 = src[i], v1 = src[i + nt], v2 = src[i + 2 * nt], v3 = src[i + 3 * nt];
        __builtin_nontemporal_store(v0, dst + i); __builtin_nontemporal_store(v1, dst + i + nt);
        __builtin_nontemporal_store(v2, dst + i + 2 * nt); __builtin_nontemporal_store(v3, dst + i + 3 * nt);
    }
    for (; i < n; i += nt) __builtin_nontemporal_store(src[i], dst + i);
}
int main() {
    const size_t bytes = size_t{1} << 30, n16 = bytes / 16;
    void *d, *h, *ta, *tb, *tc;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&ta, bytes)); CK(hipMalloc(&tb, bytes)); CK(hipMalloc(&tc, bytes));
    CK(hipMemset(d, 7, bytes)); CK(hipMemset(ta, 1, bytes)); CK(hipMemset(tb, 2, bytes));
    h = aligned_alloc(4096, bytes);
    CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
    void* hd; CK(hipHostGetDevicePointer(&hd, h, 0));
    hipStream_t sk, sc; CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    hipEvent_t k0, k1, c0, c1, dep; CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1)); CK(hipEventCreate(&c0)); CK(hipEventCreate(&c1));
    CK(hipEventCreateWithFlags(&dep, hipEventDisableTiming));
    const int KREP = 12;   // 12 x 3 GB of HBM traffic ~ 10 ms: the compute that runs beside the copy
    auto run = [&](const char* what, int mode, int blocks) -> int {
        // mode 0: no copy; 1: hipMemcpyAsync; 2: hipMemcpyAsync behind an event of the kernel stream; 3: ship kernel with `blocks` workgroups
        for (int rep = 0; rep < 2; rep++) {
            if (mode == 2) { hipLaunchKernelGGL(triad, dim3(64), dim3(256), 0, sk, (u32x4*)ta, (u32x4*)tb, (u32x4*)tc, (size_t)4096); CK(hipEventRecord(dep, sk)); CK(hipStreamWaitEvent(sc, dep, 0)); }
            CK(hipEventRecord(c0, sc));
            if (mode == 1 || mode == 2) CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, sc));
            if (mode == 3) hipLaunchKernelGGL(ship, dim3(blocks), dim3(256), 0, sc, (const u32x4*)d, (u32x4*)hd, n16);
            CK(hipEventRecord(c1, sc));
            CK(hipEventRecord(k0, sk));
            for (int i = 0; i < KREP; i++) hipLaunchKernelGGL(triad, dim3(2048), dim3(256), 0, sk, (u32x4*)ta, (u32x4*)tb, (u32x4*)tc, n16);
            CK(hipEventRecord(k1, sk));
            CK(hipStreamSynchronize(sk)); CK(hipStreamSynchronize(sc));
            float km, cm; CK(hipEventElapsedTime(&km, k0, k1)); CK(hipEventElapsedTime(&cm, c0, c1));
            if (rep == 1) printf("%-44s copy %7.2f ms = %5.1f GB/s   |  %d triads beside it %7.2f ms = %6.0f GB/s of HBM traffic\n", what, cm,
                                 mode ? bytes / cm / 1e6 : 0.0, KREP, km, 3.0 * bytes * KREP / km / 1e6);
        }
        return 0;
    };
    {   // the destination as a prover has it: inside a registered malloc'ed block, not page aligned, several copies in a row
        char* hv = (char*)malloc(bytes + 4096);
        CK(hipHostRegister(hv, bytes + 4096, hipHostRegisterDefault));
        for (size_t off : {size_t{16}, size_t{16 + 600000 * 64}}) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(c0, sc));
                CK(hipMemcpyAsync(hv + off, d, 450u << 20, hipMemcpyDeviceToHost, sc));
                CK(hipMemcpyAsync(hv + off + (450u << 20), (char*)d + (450u << 20), 5u << 20, hipMemcpyDeviceToHost, sc));
                CK(hipEventRecord(c1, sc));
                CK(hipStreamSynchronize(sc));
                float cm; CK(hipEventElapsedTime(&cm, c0, c1));
                if (rep) printf("hipMemcpyAsync x2 into registered malloc + %zu: %.2f ms = %.1f GB/s\n", off, cm, (455u << 20) / cm / 1e6);
            }
        }
    }
    if (run("no copy", 0, 0)) return 1;
    if (run("hipMemcpyAsync", 1, 0)) return 1;
    if (run("hipMemcpyAsync behind another stream's event", 2, 0)) return 1;
    for (int blocks : {4, 8, 16, 64, 256}) { char nm[64]; snprintf(nm, sizeof(nm), "ship kernel, %d workgroups", blocks); if (run(nm, 3, blocks)) return 1; }
    return 0;
}
