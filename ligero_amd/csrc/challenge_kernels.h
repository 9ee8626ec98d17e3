// Linear-test challenges on the device (src/ligero/mod.rs:719-722):
//   r_linear = get_field_elements_from_prng(4 m k, seed)   src/utils.rs:23-29  (ChaCha20Rng + F::rand)
//   r_a      = self.a.row_mul(&r_linear)                    src/matrices/mod.rs:100-110
// so that neither the 4mk-element challenge vector nor r_a crosses PCIe (90 MB per Poseidon batch) and the
// host does not spend 1.6 ms per proof on them.
//
// F::rand (ark-ff 0.4) is rejection sampling: 32 stream bytes per attempt, top two bits masked, accepted when
// below p (75.6 % of attempts), the limbs ARE the Montgomery representation.  Element i is therefore the i-th
// ACCEPTED 32-byte chunk of the ChaCha20 stream: a stream compaction.  Three launches, no stored candidates:
//   chacha_count_kernel     accepted chunks per workgroup (256 blocks = 512 chunks each)
//   chacha_scan_kernel      exclusive prefix sums of those counts (one workgroup per proof)
//   chacha_scatter_kernel   recomputes the blocks, prefix-sums the flags (wave ballots + LDS), writes element i
// The same restatement as ligero_amd/host/transcript.hpp (PARITY UNPINNED against the Rust crates, see there);
// tests compare the two bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fr_gfx950.h"

namespace lg {

struct ChaChaArgs {
    const uint32_t* seeds;   // [batch][8] key words (little-endian words of the 32-byte seed)
    fr* out;                 // [batch][n] accepted elements in stream order
    uint32_t* counts;        // [batch][wgs] accepted chunks per workgroup
    uint32_t* short_flag;    // set to 1 if a proof's candidates did not yield n elements
    uint32_t n;              // elements wanted per proof
    uint32_t blocks;         // ChaCha blocks generated per proof (2 chunks each)
    uint32_t wgs;            // workgroups per proof = ceil(blocks / 256)
};

__device__ __forceinline__ uint32_t rotl32(uint32_t v, int n) { return __builtin_amdgcn_alignbit(v, v, 32 - n); }

// one ChaCha20 block: 64-bit block counter in words 12-13, stream id 0 (rand_chacha's layout)
__device__ __forceinline__ void chacha20_block(const uint32_t* key, uint64_t counter, uint32_t (&x)[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define LG_QR(a, b, c, d)                                         \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);                 \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);                 \
    x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);                  \
    x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        LG_QR(0, 4, 8, 12) LG_QR(1, 5, 9, 13) LG_QR(2, 6, 10, 14) LG_QR(3, 7, 11, 15)
        LG_QR(0, 5, 10, 15) LG_QR(1, 6, 11, 12) LG_QR(2, 7, 8, 13) LG_QR(3, 4, 9, 14)
    }
#undef LG_QR
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] += s[i];
}

// chunk h (0 / 1) of a block as a candidate element; returns whether F::rand accepts it
__device__ __forceinline__ bool chacha_candidate(const uint32_t (&x)[16], int h, fr& e) {
#pragma unroll
    for (int i = 0; i < 8; i++) e.v[i] = x[8 * h + i];
    e.v[7] &= 0x3fffffffu;  // 256 - 254 bits shaved
    bool lt = false, decided = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (!decided && e.v[i] != fr_p(i)) { lt = e.v[i] < fr_p(i); decided = true; }
    }
    return lt;  // equal to p: rejected (is_geq_modulus)
}

// accepted chunks of this workgroup; flags returned per thread, total in every thread
__device__ __forceinline__ uint32_t wg_accept_prefix(bool a0, bool a1, uint32_t& before0, uint32_t& before1) {
    __shared__ uint32_t wave_tot[4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t m0 = __ballot(a0), m1 = __ballot(a1);
    const uint64_t below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // chunks are ordered (thread 0: c0, c1), (thread 1: c0, c1), ...
    const uint32_t in_wave_before = __popcll(m0 & below) + __popcll(m1 & below);
    if (lane == 0) wave_tot[wave] = __popcll(m0) + __popcll(m1);
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if ((uint32_t)w < wave) base += wave_tot[w];
        total += wave_tot[w];
    }
    before0 = base + in_wave_before;
    before1 = before0 + (a0 ? 1u : 0u);
    __syncthreads();
    return total;
}

static __global__ void __launch_bounds__(256) chacha_count_kernel(ChaChaArgs a) {
    const uint32_t proof = blockIdx.y, blk = blockIdx.x * 256 + threadIdx.x;
    uint32_t x[16];
    fr e;
    bool a0 = false, a1 = false;
    if (blk < a.blocks) {
        chacha20_block(a.seeds + 8 * proof, blk, x);
        a0 = chacha_candidate(x, 0, e);
        a1 = chacha_candidate(x, 1, e);
    }
    uint32_t b0, b1;
    const uint32_t total = wg_accept_prefix(a0, a1, b0, b1);
    if (threadIdx.x == 0) a.counts[(uint64_t)proof * a.wgs + blockIdx.x] = total;
}

// counts[proof][w] -> accepted chunks in the workgroups BEFORE w (exclusive prefix sums, in place): one workgroup per proof, every
// thread a contiguous run.  (Until round 4 every scatter workgroup summed the counts in front of it itself: quadratic in the length of
// the challenge vector -- 22 GB of L2 reads at 2^20 constraints, 360 GB and 20 of the kernel's 21 ms at 2^22.)
static __global__ void __launch_bounds__(1024) chacha_scan_kernel(ChaChaArgs a) {
    __shared__ uint32_t part[1024];
    uint32_t* cnt = a.counts + (uint64_t)blockIdx.x * a.wgs;
    uint32_t carry = 0;                                   // accepted chunks in the tiles before this one
    for (uint32_t base = 0; base < a.wgs; base += 4096) {       // tiles of 4096 counts, four consecutive ones per thread: coalesced
        const uint32_t i0 = base + 4 * threadIdx.x;
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = i0 + j < a.wgs ? cnt[i0 + j] : 0;
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        part[threadIdx.x] = mine;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {              // inclusive scan of the threads' sums
            const uint32_t u = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += u;
            __syncthreads();
        }
        uint32_t run = carry + part[threadIdx.x] - mine;
        const uint32_t tile_total = part[1023];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (i0 + j < a.wgs) cnt[i0 + j] = run;
            run += v[j];
        }
        carry += tile_total;
        __syncthreads();                                  // part[] is rewritten by the next tile
    }
}

static __global__ void __launch_bounds__(256) chacha_scatter_kernel(ChaChaArgs a) {      // counts = what chacha_scan_kernel left
    const uint32_t proof = blockIdx.y, blk = blockIdx.x * 256 + threadIdx.x;
    const uint32_t offset = a.counts[(uint64_t)proof * a.wgs + blockIdx.x];
    uint32_t x[16];
    fr e0, e1;
    bool a0 = false, a1 = false;
    if (blk < a.blocks) {
        chacha20_block(a.seeds + 8 * proof, blk, x);
        a0 = chacha_candidate(x, 0, e0);
        a1 = chacha_candidate(x, 1, e1);
    }
    uint32_t b0, b1;
    const uint32_t total = wg_accept_prefix(a0, a1, b0, b1);
    fr* out = a.out + (uint64_t)proof * a.n;
    if (a0 && offset + b0 < a.n) fr_store(out + offset + b0, e0);
    if (a1 && offset + b1 < a.n) fr_store(out + offset + b1, e1);
    if (blockIdx.x + 1 == a.wgs && threadIdx.x == 0 && offset + total < a.n) atomicExch(a.short_flag, 1u);
}

// r_a[p][col] = sum over the entries (row, value) of column col of r[p][row] * value   (A in CSC form).
// Columns are very uneven: the column of the constant one collects an entry for every constant of the circuit
// (4237 of 54051 on the Poseidon instance, the rest have <= 69), so columns above kHeavyColumn entries are left
// to a second launch that gives each of them a whole workgroup.
constexpr uint32_t kHeavyColumn = 128;
struct SparseRowMulArgs {
    const uint32_t* col_ptr;  // [cols + 1]
    const uint32_t* ent_row;  // [nnz]
    const fr* ent_val;        // [nnz] Montgomery
    const fr* r;              // [batch][rows_in] Montgomery
    fr* out;                  // [batch][cols] Montgomery, fully reduced
    const uint32_t* heavy;    // column ids with more than kHeavyColumn entries
    uint32_t cols, rows_in;
};
__device__ __forceinline__ void sparse_accumulate(const SparseRowMulArgs& a, const fr* r, uint32_t e, fr& acc) {
    const uint32_t row = a.ent_row[e];
    if (row >= a.rows_in) return;  // row_mul zips the challenge with the rows: rows beyond it do not contribute
    fr t;
    fr_mul_lazy(t, fr_load(r + row), fr_load(a.ent_val + e));
    fr_add_lazy(acc, acc, t);
}
static __global__ void __launch_bounds__(256) sparse_row_mul_kernel(SparseRowMulArgs a) {
    const uint32_t col = blockIdx.x * 256 + threadIdx.x, proof = blockIdx.y;
    if (col >= a.cols) return;
    const uint32_t e0 = a.col_ptr[col], e1 = a.col_ptr[col + 1];
    if (e1 - e0 > kHeavyColumn) return;
    const fr* r = a.r + (uint64_t)proof * a.rows_in;
    fr acc;
#pragma unroll
    for (int l = 0; l < 8; l++) acc.v[l] = 0;
    for (uint32_t e = e0; e < e1; e++) sparse_accumulate(a, r, e, acc);
    fr red;
    fr_reduce(red, acc);
    fr_store(a.out + (uint64_t)proof * a.cols + col, red);
}
// Heavy columns are cut into segments of kHeavySegment entries so that one column with a million entries (the constant-one
// column of a circuit with a million outputs: every "+ 1" refers to it) spreads over the chip instead of one workgroup.
//   1. grid (segments, batch): the workgroup strides over its segment and tree-reduces in LDS -> seg_partial[proof][segment]
//   2. grid (heavy columns, batch): the same over the column's partial sums -> out
constexpr uint32_t kHeavySegment = 2048;
struct HeavySegArgs {
    SparseRowMulArgs m;
    const uint32_t* seg_begin;     // [nseg + 1] entry range of segment i = [seg_begin[i], seg_end[i])
    const uint32_t* seg_end;
    const uint32_t* heavy_seg_ptr; // [nheavy + 1] segments of heavy column h = [heavy_seg_ptr[h], heavy_seg_ptr[h + 1])
    fr* seg_partial;               // [batch][nseg] lazy sums
    uint32_t nseg;
};
__device__ __forceinline__ fr block_sum(fr acc, fr* part) {
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            fr x = part[threadIdx.x], y = part[threadIdx.x + d];
            fr_add_lazy(x, x, y);
            part[threadIdx.x] = x;
        }
        __syncthreads();
    }
    return part[0];
}
static __global__ void __launch_bounds__(256) sparse_row_mul_heavy_segments_kernel(HeavySegArgs a) {
    __shared__ fr part[256];
    const uint32_t seg = blockIdx.x, proof = blockIdx.y;
    const fr* r = a.m.r + (uint64_t)proof * a.m.rows_in;
    fr acc;
#pragma unroll
    for (int l = 0; l < 8; l++) acc.v[l] = 0;
    for (uint32_t e = a.seg_begin[seg] + threadIdx.x; e < a.seg_end[seg]; e += 256) sparse_accumulate(a.m, r, e, acc);
    const fr total = block_sum(acc, part);
    if (threadIdx.x == 0) fr_store(a.seg_partial + (uint64_t)proof * a.nseg + seg, total);
}
static __global__ void __launch_bounds__(256) sparse_row_mul_heavy_finish_kernel(HeavySegArgs a) {
    __shared__ fr part[256];
    const uint32_t h = blockIdx.x, proof = blockIdx.y;
    const fr* partial = a.seg_partial + (uint64_t)proof * a.nseg;
    fr acc;
#pragma unroll
    for (int l = 0; l < 8; l++) acc.v[l] = 0;
    for (uint32_t i = a.heavy_seg_ptr[h] + threadIdx.x; i < a.heavy_seg_ptr[h + 1]; i += 256) {
        const fr t = fr_load(partial + i);
        fr_add_lazy(acc, acc, t);
    }
    const fr total = block_sum(acc, part);
    if (threadIdx.x == 0) {
        fr red;
        fr_reduce(red, total);
        fr_store(a.m.out + (uint64_t)proof * a.m.cols + a.m.heavy[h], red);
    }
}

}  // namespace lg
