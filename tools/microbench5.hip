// Instruction-issue micro-benchmarks, second list (gfx950): the instructions the "non-multiplier" half of the NTT and
// hash kernels is made of, and candidates for replacing them.  Same harness as microbench2.hip: 64 independent ops per
// loop trip on 8 registers, W = 1/2/4/8 waves per SIMD on every CU, in-kernel clock64() + HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define REP8(x) x x x x x x x x

// one asm statement = the same instruction on 8 independent 32-bit registers r0..r7 (+ operand %8 = a, %9 = b)
#define OP8_32(INS) REP8(asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b), "s"(sm) : "vcc");)
#define OP8_64(INS) REP8(asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(a), "v"(b), "s"(sm) : "vcc");)

#define I_LSHR64(n) "v_lshrrev_b64 %" #n ", 3, %" #n "\n"
#define I_LSHL64(n) "v_lshlrev_b64 %" #n ", 1, %" #n "\n"
#define I_CND_VCC(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CND_SGPR(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, %10\n"
#define I_BFI(n) "v_bfi_b32 %" #n ", %9, %" #n ", %8\n"
#define I_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 29\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 3, %9\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %9\n"
#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_LSHR32(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define I_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_ADDLIT(n) "v_add_u32 %" #n ", 0x2b6d0301, %" #n "\n"
#define I_MOVDPP_QP(n) "v_mov_b32_dpp %" #n ", %" #n " quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
#define I_MOVDPP_ROR(n) "v_mov_b32_dpp %" #n ", %" #n " row_ror:8 row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP(n) "v_add_u32_dpp %" #n ", %8, %" #n " quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n"
#define I_XORDPP(n) "v_xor_b32_dpp %" #n ", %8, %" #n " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x96\n"
#define I_MADU24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_MULU24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_MULHIU24(n) "v_mul_hi_u32_u24 %" #n ", %" #n ", %8\n"
#define I_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %" #n ", %" #n "\n"
#define I_MUL64F(n) "v_mul_f64 %" #n ", %" #n ", %" #n "\n"
#define I_ADD64F(n) "v_add_f64 %" #n ", %" #n ", %" #n "\n"
#define I_FMA32(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PKFMA32(n) "v_pk_fma_f32 %" #n ", %" #n ", %" #n ", %" #n "\n"
#define I_MADI64(n) "v_mad_i64_i32 %" #n ", vcc, %8, %9, %" #n "\n"
#define I_SWZ(n) "ds_swizzle_b32 %" #n ", %" #n " offset:0x101f\n"          /* bitmask mode: xor 4 */
#define I_BPERM(n) "ds_bpermute_b32 %" #n ", %8, %" #n "\n"
#define I_PLANE16(n) "v_permlane16_swap_b32 %" #n ", %" #n "\n"
#define I_XORSDWA(n) "v_xor_b32_sdwa %" #n ", %" #n ", %8 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_MAD_SCONST(n) "v_mad_u64_u32 %" #n ", vcc, %8, 0x0f05360, %" #n "\n"

enum Op { LSHR64, LSHL64, CND_VCC, CND_SGPR, BFI, ALIGNBIT, ANDOR, LSHLOR, LSHLADD, AND, XOR, LSHR32, SUB, ADDLIT, MOVDPP_QP, MOVDPP_ROR, ADDDPP, XORDPP, PERM, BITOP3,
          MADU24, MULU24, MULHIU24, FMA64, MUL64F, ADD64F, FMA32, PKFMA32, MADI64, SWZ, BPERM, ADD3, MAD_LIT, XORSDWA };

template <int OP>
__global__ void kern(uint32_t* out, uint32_t a0, int iters, long long* cyc) {
    uint32_t a = a0 + threadIdx.x, b = a0 * 3 + threadIdx.x;
    uint64_t q0 = threadIdx.x + 0x3ff0000000000000ull, q1 = 0x3ff0000000000001ull, q2 = 0x3ff0000000000002ull, q3 = 0x3ff0000000000003ull,
             q4 = 0x3ff0000000000004ull, q5 = 0x3ff0000000000005ull, q6 = 0x3ff0000000000006ull, q7 = 0x3ff0000000000007ull;
    uint32_t r0 = 1, r1 = 2, r2 = 3, r3 = 4, r4 = 5, r5 = 6, r6 = 7, r7 = 8;
    uint64_t sm = 0x5555555555555555ull ^ a0;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if constexpr (OP == LSHR64) { OP8_64(I_LSHR64) }
        else if constexpr (OP == LSHL64) { OP8_64(I_LSHL64) }
        else if constexpr (OP == CND_VCC) { OP8_32(I_CND_VCC) }
        else if constexpr (OP == CND_SGPR) { OP8_32(I_CND_SGPR) }
        else if constexpr (OP == BFI) { OP8_32(I_BFI) }
        else if constexpr (OP == ALIGNBIT) { OP8_32(I_ALIGNBIT) }
        else if constexpr (OP == ANDOR) { OP8_32(I_ANDOR) }
        else if constexpr (OP == LSHLOR) { OP8_32(I_LSHLOR) }
        else if constexpr (OP == LSHLADD) { OP8_32(I_LSHLADD) }
        else if constexpr (OP == AND) { OP8_32(I_AND) }
        else if constexpr (OP == XOR) { OP8_32(I_XOR) }
        else if constexpr (OP == LSHR32) { OP8_32(I_LSHR32) }
        else if constexpr (OP == SUB) { OP8_32(I_SUB) }
        else if constexpr (OP == ADDLIT) { OP8_32(I_ADDLIT) }
        else if constexpr (OP == MOVDPP_QP) { OP8_32(I_MOVDPP_QP) }
        else if constexpr (OP == MOVDPP_ROR) { OP8_32(I_MOVDPP_ROR) }
        else if constexpr (OP == ADDDPP) { OP8_32(I_ADDDPP) }
        else if constexpr (OP == XORDPP) { OP8_32(I_XORDPP) }
        else if constexpr (OP == PERM) { OP8_32(I_PERM) }
        else if constexpr (OP == BITOP3) { OP8_32(I_BITOP3) }
        else if constexpr (OP == MADU24) { OP8_32(I_MADU24) }
        else if constexpr (OP == MULU24) { OP8_32(I_MULU24) }
        else if constexpr (OP == MULHIU24) { OP8_32(I_MULHIU24) }
        else if constexpr (OP == FMA64) { OP8_64(I_FMA64) }
        else if constexpr (OP == MUL64F) { OP8_64(I_MUL64F) }
        else if constexpr (OP == ADD64F) { OP8_64(I_ADD64F) }
        else if constexpr (OP == FMA32) { OP8_32(I_FMA32) }
        else if constexpr (OP == PKFMA32) { OP8_64(I_PKFMA32) }
        else if constexpr (OP == MADI64) { OP8_64(I_MADI64) }
        else if constexpr (OP == SWZ) { OP8_32(I_SWZ) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        else if constexpr (OP == BPERM) { OP8_32(I_BPERM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        else if constexpr (OP == ADD3) { OP8_32(I_ADD3) }
        else if constexpr (OP == XORSDWA) { OP8_32(I_XORSDWA) }

    }
    long long t1 = clock64();
    uint64_t s = q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
int run(const char* name, int cus, uint32_t* buf, long long* dcyc, int ops_per_iter) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-22s", name);
    for (int W : {1, 2, 4, 8}) {
        const int grid = cus * W, iters = 12000 / W;
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
        // "GHz" = MEAN wave lifetime / kernel time: the SIMD issues oldest-wave-first, so with W waves per SIMD the waves finish one
        // after the other and the mean lifetime is ~(W + 1) / 2W of the kernel -- NOT a clock.  "clk" = LONGEST lifetime / kernel
        // time is the clock the chip held (correction)
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
#define RUN(OP, NAME) run<OP>(NAME, cus, buf, dcyc, 64);
    RUN(LSHR64, "v_lshrrev_b64") RUN(LSHL64, "v_lshlrev_b64") RUN(CND_VCC, "v_cndmask vcc") RUN(CND_SGPR, "v_cndmask_e64 sgpr")
    RUN(BFI, "v_bfi_b32") RUN(ALIGNBIT, "v_alignbit_b32") RUN(ANDOR, "v_and_or_b32") RUN(LSHLOR, "v_lshl_or_b32") RUN(LSHLADD, "v_lshl_add_u32")
    RUN(XORSDWA, "v_xor_b32_sdwa word") RUN(ADD3, "v_add3_u32") RUN(AND, "v_and_b32") RUN(XOR, "v_xor_b32") RUN(LSHR32, "v_lshrrev_b32") RUN(SUB, "v_sub_u32") RUN(ADDLIT, "v_add_u32 literal")
    RUN(MOVDPP_QP, "v_mov_dpp quad_perm") RUN(MOVDPP_ROR, "v_mov_dpp row_ror:8") RUN(ADDDPP, "v_add_u32_dpp qp") RUN(XORDPP, "v_xor_b32_dpp qp")
    RUN(PERM, "v_perm_b32") RUN(BITOP3, "v_bitop3_b32") RUN(MADU24, "v_mad_u32_u24") RUN(MULU24, "v_mul_u32_u24") RUN(MULHIU24, "v_mul_hi_u32_u24")
    RUN(MADI64, "v_mad_i64_i32")
    RUN(FMA64, "v_fma_f64") RUN(MUL64F, "v_mul_f64") RUN(ADD64F, "v_add_f64") RUN(FMA32, "v_fma_f32") RUN(PKFMA32, "v_pk_fma_f32")
    RUN(SWZ, "ds_swizzle_b32 xor4") RUN(BPERM, "ds_bpermute_b32")
    return 0;
}
