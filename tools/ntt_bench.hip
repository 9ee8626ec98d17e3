// Stand-alone timing of the row-NTT kernels (ligero_amd/csrc/ntt_kernels.h) with synthetic
// tables, for ablation experiments: compile with -DLG_LOGK=<n> and optional -DLG_ABL_* macros.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ntt_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
using namespace lg;
int main(int argc, char** argv) {
    constexpr int LOGK = LG_LOGK;
    constexpr int K = 1 << LOGK;
    const uint32_t rows = argc > 1 ? atoi(argv[1]) : (LOGK >= 10 ? 2509 : 22016);
    using Plan = NttPlan<LOGK>;
    fr *in, *out;
    uint8_t *tw, *ctw;
    CK(hipMalloc((void**)&in, (size_t)rows * K * 32));
    CK(hipMalloc((void**)&out, (size_t)8 * rows * K * 32));
    const size_t ntw = pass_tw_total(LOGK) ? pass_tw_total(LOGK) : 1;
    uint8_t* f2;
    CK(hipMalloc((void**)&tw, ntw * 72));
    CK(hipMalloc((void**)&ctw, (size_t)8 * K * 72));
    CK(hipMalloc((void**)&f2, (size_t)8 * 2 * K * 36));
    CK(hipMemset(in, 0x11, (size_t)rows * K * 32));
    CK(hipMemset(tw, 0x05, ntw * 72));
    CK(hipMemset(ctw, 0x03, (size_t)8 * K * 72));
    CK(hipMemset(f2, 0x07, (size_t)8 * 2 * K * 36));
    NttArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.canon_out = nullptr;
    auto planes = [](const uint8_t* b, size_t n) { return Tw29{(const uint4*)b, (const uint4*)(b + 16 * n), (const uint32_t*)(b + 32 * n)}; };
    a.tw = Tw29q{planes(tw, ntw), planes(tw + 36 * ntw, ntw)};
    a.coset_tw = Tw29q{planes(ctw, (size_t)8 * K), planes(ctw + (size_t)36 * 8 * K, (size_t)8 * K)};
    a.first2 = planes(f2, (size_t)8 * 2 * K);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 9; j++) { a.w8[i].v[j] = 0x01234567u >> (j & 3); a.w8q[i].v[j] = 0x00765432u >> (j & 3); }
    for (int j = 0; j < 9; j++) { a.one.v[j] = 0x00abcdefu; a.oneq.v[j] = 0x00fedcbau; a.scale.v[j] = 0x00123456u; a.invk.v[j] = 0x00345678u; a.invkq.v[j] = 0x00876543u; }
    a.rows = rows; a.row0 = 0; a.ncos = 7; a.chunk_rows = rows; a.proof_stride = 0;
    for (int s = 0; s < 7; s++) a.cosets[s] = s + 1;
    a.plane_stride = (uint64_t)rows * K;
    auto kern = ntt_rows_kernel<LOGK, 0, true>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Plan::kLdsBytes));
    const uint64_t work = (uint64_t)rows * 7;
    uint32_t grid = (uint32_t)((work + Plan::kNttsPerWg - 1) / Plan::kNttsPerWg);
    if (argc > 2 && atoi(argv[2]) > 0 && grid > (uint32_t)atoi(argv[2])) grid = atoi(argv[2]);  // persistent workgroups
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 5; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(Plan::kWgThreads), Plan::kLdsBytes, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("logk=%d rows=%u evaluate(7 cosets): %.3f ms  (%.1f us per WG-slot on 256 CUs)\n", LOGK, rows, best, best * 1e3 * 256 / ((work + Plan::kNttsPerWg - 1) / Plan::kNttsPerWg));
    return 0;
}
