// f3 on the device: the kernels of the evaluation trace and the check of a trace program, shared by the commitment contexts
// (witness.hip: lg_encode_commit_from_inputs, lg_prove_batch_queue_inputs) and the stand-alone tracer of sharded provers (tracer.hip).
#pragma once
#include <algorithm>
#include <vector>

#include "fr_gfx950.h"
#include "trace_plan.h"

namespace lg {

constexpr uint32_t kGateNone = 0xffffffffu, kGateConst = 0x80000000u;

// f3 on the device (arithmetic_circuit/mod.rs:325-358): the evaluation trace, one launch per dependency level.  w of every proof
// sits in the W block of d_preenc; a gate reads its operands (positions of w written by earlier levels or by the scatter of the
// assignment, or constants that have no position) and writes its own position.  Values stay fully reduced Montgomery words, as the
// host's evaluation leaves them, so the W block is the same bytes.
constexpr uint8_t kTraceInput = 0, kTraceAdd = 1, kTraceMul = 2, kTraceOne = 3;
struct TraceLevelArgs {
    fr* pre;                 // [batch][4 m][k]
    const uint8_t* op;       // [npos]
    const uint32_t* left;    // [npos]
    const uint32_t* right;
    const fr* consts;
    const uint32_t* order;   // positions of the gates, level by level
    uint64_t begin, end;     // this level = order[begin, end)
    uint64_t mk;
    uint32_t batch;
};
__device__ __forceinline__ void trace_gate(const TraceLevelArgs& a, fr* w, uint32_t pos) {
    const uint32_t l = a.left[pos], r = a.right[pos];
    const fr x = (l & kGateConst) ? fr_load(a.consts + (l & ~kGateConst)) : fr_load(w + l);
    const fr y = (r & kGateConst) ? fr_load(a.consts + (r & ~kGateConst)) : fr_load(w + r);
    fr t, z;
    if (a.op[pos] == kTraceMul) fr_mul_lazy(t, x, y);
    else fr_add_raw(t, x, y);                    // both < p: the sum < 2p
    fr_reduce(z, t);
    fr_store(w + pos, z);
}
static __global__ void __launch_bounds__(256) trace_level_kernel(const TraceLevelArgs a) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t span = a.end - a.begin;
    if (gid >= span * a.batch) return;
    trace_gate(a, a.pre + (gid / span) * 4 * a.mk + 3 * a.mk, a.order[a.begin + gid % span]);
}
// A run of NARROW levels (a few hundred gates each: the tail of a Poseidon circuit is sixty of them, a builder-made chain hundreds) in
// ONE launch: a workgroup per proof walks the levels with a barrier between them -- a proof's gates read that proof's values only, and
// the waves of a workgroup share their CU's vector cache, so what one wave stored before the barrier the others load after it.
struct TraceFusedArgs {
    TraceLevelArgs lv;            // begin / end unused
    const uint64_t* level_off;    // device copy
    uint32_t level0, level1;      // levels [level0, level1) (0-based index into level_off)
};
static __global__ void __launch_bounds__(256) trace_fused_kernel(const TraceFusedArgs a) {
    fr* w = a.lv.pre + (uint64_t)blockIdx.x * 4 * a.lv.mk + 3 * a.lv.mk;
    for (uint32_t l = a.level0; l < a.level1; l++) {
        const uint64_t b = a.level_off[l], e = a.level_off[l + 1];
        for (uint64_t i = b + threadIdx.x; i < e; i += 256) trace_gate(a.lv, w, a.lv.order[i]);
        __syncthreads();
    }
}

struct TraceScatterArgs {
    fr* pre;
    const uint32_t* in_pos;  // [nin] positions of the assigned variables (the same for every proof)
    const fr* in_vals;       // [batch][nin]
    uint64_t nin, mk;
    uint32_t batch;
    uint32_t has_one;        // position 0 is the leading constant one
    fr one;
};
static __global__ void __launch_bounds__(256) trace_scatter_kernel(const TraceScatterArgs a) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per = a.nin + 1;              // slot nin of every proof writes the one
    if (gid >= per * a.batch) return;
    const uint64_t b = gid / per, i = gid % per;
    fr* w = a.pre + b * 4 * a.mk + 3 * a.mk;
    if (i == a.nin) { if (a.has_one) fr_store(w, a.one); return; }
    fr_store(w + a.in_pos[i], fr_load(a.in_vals + b * a.nin + i));
}

struct TraceOutputsArgs {
    const fr* pre;
    const uint32_t* outputs; // [nout] positions
    uint32_t* ok;            // [batch]
    uint64_t mk;
    uint32_t nout, batch;
    fr one;
};
static __global__ void __launch_bounds__(256) trace_outputs_kernel(const TraceOutputsArgs a) {      // ok[] preset to 1; grid (slices, batch)
    const uint32_t b = blockIdx.y;
    const fr* w = a.pre + (uint64_t)b * 4 * a.mk + 3 * a.mk;
    uint32_t bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.nout; i += (uint64_t)gridDim.x * blockDim.x) {
        const fr v = fr_load(w + a.outputs[i]);
#pragma unroll
        for (int j = 0; j < 8; j++) bad |= v.v[j] ^ a.one.v[j];
    }
    if (__ballot(bad != 0) && (threadIdx.x & 63) == 0) atomicExch(a.ok + b, 0u);
}


// The device trusts nothing of a program it has not seen checked: every gate exactly once in `order`, operands in range, and every
// operand in an EARLIER level than its gate (inputs and the one are level 0) -- which is also what makes the launches race free.
inline bool trace_program_ok(uint64_t npos, const uint8_t* op, const uint32_t* left, const uint32_t* right, uint32_t nconst, const uint32_t* order, uint64_t ngates,
                             const uint64_t* level_off, uint32_t nlevels, const uint32_t* outputs, uint32_t nout, uint64_t* inputs_out) {
    if (npos >= kGateConst) return false;
    if (level_off[0] != 0 || level_off[nlevels] != ngates) return false;
    for (uint32_t l = 0; l < nlevels; l++)
        if (level_off[l + 1] < level_off[l]) return false;
    std::vector<uint32_t> lev(npos, 0);
    uint64_t gates = 0, inputs = 0;
    for (uint64_t p = 0; p < npos; p++) {
        if (op[p] == kTraceAdd || op[p] == kTraceMul) { gates++; lev[p] = 0xffffffffu; }
        else if (op[p] == kTraceInput) inputs++;
        else if (op[p] != kTraceOne || p != 0) return false;
    }
    if (gates != ngates) return false;
    for (uint32_t l = 0; l < nlevels; l++)
        for (uint64_t i = level_off[l]; i < level_off[l + 1]; i++) {
            const uint32_t p = order[i];
            if (p >= npos || lev[p] != 0xffffffffu) return false;      // not a gate, or listed twice
            lev[p] = l + 1;
        }
    for (uint64_t p = 0; p < npos; p++) {
        if (op[p] != kTraceAdd && op[p] != kTraceMul) continue;
        for (uint32_t s : {left[p], right[p]}) {
            if (s & kGateConst) { if ((s & ~kGateConst) >= nconst) return false; }
            else if (s >= npos || lev[s] >= lev[p]) return false;
        }
    }
    for (uint32_t i = 0; i < nout; i++)
        if (outputs[i] >= npos) return false;
    *inputs_out = inputs;
    return true;
}

}  // namespace lg
