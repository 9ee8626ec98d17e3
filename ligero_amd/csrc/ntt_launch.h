// Host-side entry points of the per-size row-NTT translation units (ntt_inst.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_kernels.h"

namespace lg {
// logki = log2 of the LDS-resident transform size, logo = log2 of the outer radix (0 unless logki == 12)
#define LG_DECL_NTT(N) hipError_t launch_ntt_logk_##N(int logo, bool evaluate, hipStream_t st, const NttArgs& a);
LG_DECL_NTT(1) LG_DECL_NTT(2) LG_DECL_NTT(3) LG_DECL_NTT(4) LG_DECL_NTT(5) LG_DECL_NTT(6)
LG_DECL_NTT(7) LG_DECL_NTT(8) LG_DECL_NTT(9) LG_DECL_NTT(10) LG_DECL_NTT(11) LG_DECL_NTT(12)
#undef LG_DECL_NTT

inline hipError_t launch_ntt(int logki, int logo, bool evaluate, hipStream_t st, const NttArgs& a) {
    switch (logki) {
        case 1: return launch_ntt_logk_1(logo, evaluate, st, a);
        case 2: return launch_ntt_logk_2(logo, evaluate, st, a);
        case 3: return launch_ntt_logk_3(logo, evaluate, st, a);
        case 4: return launch_ntt_logk_4(logo, evaluate, st, a);
        case 5: return launch_ntt_logk_5(logo, evaluate, st, a);
        case 6: return launch_ntt_logk_6(logo, evaluate, st, a);
        case 7: return launch_ntt_logk_7(logo, evaluate, st, a);
        case 8: return launch_ntt_logk_8(logo, evaluate, st, a);
        case 9: return launch_ntt_logk_9(logo, evaluate, st, a);
        case 10: return launch_ntt_logk_10(logo, evaluate, st, a);
        case 11: return launch_ntt_logk_11(logo, evaluate, st, a);
        case 12: return launch_ntt_logk_12(logo, evaluate, st, a);
        default: return hipErrorInvalidValue;
    }
}
}  // namespace lg
