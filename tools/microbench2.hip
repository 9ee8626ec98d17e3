// Instruction-issue micro-benchmarks (gfx950): cycles per wave-instruction per SIMD for the
// instructions the field arithmetic is made of, at 1/2/4/8 waves per SIMD.  64 ops per loop
// trip so loop overhead is negligible; time is wall time via HIP events, and in-kernel
// s_memtime: the LONGEST wave lifetime / kernel time gives the clock actually held (the mean does not: see the printf).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

enum Op { MAD_IND, MAD_DEP, MAD_CARRYOUT, MULLO, MULHI, ADD, ADDCO_CHAIN, ADD64, CNDMASK, MOV, MAD_SGPR, ADD3, MAD_ADDC_PAIR, MAD_ADD_GROUPED };

template <int OP>
__global__ void kern(uint32_t* out, uint32_t a0, int iters, long long* cyc) {
    uint32_t a = a0 + threadIdx.x, b = a0 * 3 + threadIdx.x;
    uint64_t acc0 = threadIdx.x, acc1 = 1, acc2 = 2, acc3 = 3, acc4 = 4, acc5 = 5, acc6 = 6, acc7 = 7;
    uint32_t r0 = 1, r1 = 2, r2 = 3, r3 = 4, r4 = 5, r5 = 6, r6 = 7, r7 = 8;
    uint64_t cy;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if constexpr (OP == MAD_IND) {
            REP8(asm volatile("v_mad_u64_u32 %0, %8, %9, %10, %0\n v_mad_u64_u32 %1, %8, %9, %10, %1\n v_mad_u64_u32 %2, %8, %9, %10, %2\n v_mad_u64_u32 %3, %8, %9, %10, %3\n"
                              "v_mad_u64_u32 %4, %8, %9, %10, %4\n v_mad_u64_u32 %5, %8, %9, %10, %5\n v_mad_u64_u32 %6, %8, %9, %10, %6\n v_mad_u64_u32 %7, %8, %9, %10, %7\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7), "=s"(cy) : "v"(a), "v"(b));)
        } else if constexpr (OP == MAD_DEP) {
            REP64(asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc0), "=s"(cy) : "v"(a), "v"(b));)
        } else if constexpr (OP == MAD_CARRYOUT) {  // mad followed by addc consuming its carry
            REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_addc_co_u32 %4, vcc, 0, %4, vcc\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n"
                              "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_addc_co_u32 %6, vcc, 0, %6, vcc\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_addc_co_u32 %7, vcc, 0, %7, vcc\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b) : "vcc");)
        } else if constexpr (OP == MULLO) {
            REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));)
        } else if constexpr (OP == MULHI) {
            REP8(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));)
        } else if constexpr (OP == ADD) {
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));)
        } else if constexpr (OP == ADDCO_CHAIN) {  // an 8-limb carry chain
            REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");)
        } else if constexpr (OP == ADD64) {
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7) : "v"(acc0));)
        } else if constexpr (OP == CNDMASK) {
            REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");)
        } else if constexpr (OP == MOV) {
            REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));)
        } else if constexpr (OP == MAD_SGPR) {  // one multiplicand from an SGPR
            uint32_t sb = __builtin_amdgcn_readfirstlane(b);
            REP8(asm volatile("v_mad_u64_u32 %0, %8, %9, %10, %0\n v_mad_u64_u32 %1, %8, %9, %10, %1\n v_mad_u64_u32 %2, %8, %9, %10, %2\n v_mad_u64_u32 %3, %8, %9, %10, %3\n"
                              "v_mad_u64_u32 %4, %8, %9, %10, %4\n v_mad_u64_u32 %5, %8, %9, %10, %5\n v_mad_u64_u32 %6, %8, %9, %10, %6\n v_mad_u64_u32 %7, %8, %9, %10, %7\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(acc4), "+v"(acc5), "+v"(acc6), "+v"(acc7), "=s"(cy) : "v"(a), "s"(sb));)
        } else if constexpr (OP == ADD3) {
            REP8(asm volatile("v_add3_u32 %0, %0, %8, %1\n v_add3_u32 %1, %1, %8, %2\n v_add3_u32 %2, %2, %8, %3\n v_add3_u32 %3, %3, %8, %4\n v_add3_u32 %4, %4, %8, %5\n v_add3_u32 %5, %5, %8, %6\n v_add3_u32 %6, %6, %8, %7\n v_add3_u32 %7, %7, %8, %0\n"
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));)
        } else if constexpr (OP == MAD_ADDC_PAIR) {  // 1 mad : 1 independent add (the mulmod mix)
            REP8(asm volatile("v_mad_u64_u32 %0, %8, %9, %10, %0\n v_add_u32 %4, %4, %9\n v_mad_u64_u32 %1, %8, %9, %10, %1\n v_add_u32 %5, %5, %9\n"
                              "v_mad_u64_u32 %2, %8, %9, %10, %2\n v_add_u32 %6, %6, %9\n v_mad_u64_u32 %3, %8, %9, %10, %3\n v_add_u32 %7, %7, %9\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=s"(cy) : "v"(a), "v"(b));)
        } else if constexpr (OP == MAD_ADD_GROUPED) {  // the same 1 : 1 mix, four of a class in a row instead of alternating
            REP8(asm volatile("v_mad_u64_u32 %0, %8, %9, %10, %0\n v_mad_u64_u32 %1, %8, %9, %10, %1\n v_mad_u64_u32 %2, %8, %9, %10, %2\n v_mad_u64_u32 %3, %8, %9, %10, %3\n"
                              "v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %9\n v_add_u32 %7, %7, %9\n"
                              : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "=s"(cy) : "v"(a), "v"(b));)
        }
    }
    long long t1 = clock64();
    uint64_t s = acc0 ^ acc1 ^ acc2 ^ acc3 ^ acc4 ^ acc5 ^ acc6 ^ acc7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
int run(const char* name, int cus, uint32_t* buf, long long* dcyc, int ops_per_iter) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-16s", name);
    for (int W : {1, 2, 4, 8}) {
        const int grid = cus * W, iters = 20000 / W;
        hipLaunchKernelGGL(kern<OP>, dim3(grid), dim3(256), 0, 0, buf, 12345u, 100, dcyc);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern<OP>, dim3(grid), dim3(256), 0, 0, buf, 12345u, iters, dcyc);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> c(grid);
        CK(hipMemcpy(c.data(), dcyc, grid * 8, hipMemcpyDeviceToHost));
        double avg = 0, mx = 0;
        for (auto v : c) { avg += v; if (v > mx) mx = (double)v; }
        avg /= grid;
        const double waveops = (double)iters * ops_per_iter * W;  // per SIMD
        // "GHz" = MEAN wave lifetime / kernel time, which is ~(W + 1) / 2W of the clock under oldest-wave-first issue -- not a clock;
        // "clk" = LONGEST lifetime / kernel time is the clock the chip held (correction)
        printf(" | W=%d %6.2f cyc/op (%5.2f ns/op, %.2f GHz, clk %.2f)", W, avg / waveops, ms * 1e6 / waveops, avg / (ms * 1e6), mx / (ms * 1e6));
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
    CK(hipMalloc((void**)&buf, (size_t)cus * 8 * 256 * 4));
    CK(hipMalloc((void**)&dcyc, (size_t)cus * 8 * 8));
    printf("cycles per wave-instruction per SIMD (avg in-kernel cycles / ops issued on that SIMD), by waves per SIMD\n");
    run<MAD_IND>("mad_u64 indep", cus, buf, dcyc, 64);
    run<MAD_DEP>("mad_u64 dep", cus, buf, dcyc, 64);
    run<MAD_SGPR>("mad_u64 sgpr", cus, buf, dcyc, 64);
    run<MAD_CARRYOUT>("mad+addc(vcc)", cus, buf, dcyc, 64);
    run<MAD_ADDC_PAIR>("mad+add indep", cus, buf, dcyc, 64);
    run<MAD_ADD_GROUPED>("mad x4, add x4", cus, buf, dcyc, 64);
    run<MULLO>("mul_lo_u32", cus, buf, dcyc, 64);
    run<MULHI>("mul_hi_u32", cus, buf, dcyc, 64);
    run<ADD>("add_u32", cus, buf, dcyc, 64);
    run<ADDCO_CHAIN>("addc chain", cus, buf, dcyc, 64);
    run<ADD64>("lshl_add_u64", cus, buf, dcyc, 64);
    run<CNDMASK>("cndmask", cus, buf, dcyc, 64);
    run<MOV>("mov_b32", cus, buf, dcyc, 64);
    run<ADD3>("add3_u32", cus, buf, dcyc, 64);
    return 0;
}
