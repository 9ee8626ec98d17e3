// Column hashing (Blake2s-256) and Merkle tree (SHA-256) kernels for gfx950.
//
// Replaces, from NP-Eng/ligero:
//   * u.columns() + H::evaluate per column        src/ligero/mod.rs:536-542, src/matrices/mod.rs:163-167
//     with H = FieldToBytesColHasher<F, Blake2s256> (src/ligero/types.rs:18): digest_j =
//     Blake2s-256( LE64(rows) || canonicalLE32(U[0][j]) || ... || canonicalLE32(U[rows-1][j]) )
//   * create_merkle_tree + root                   src/ligero/mod.rs:544-551
//     with TestMerkleTreeParams (types.rs:6-8,25-26): identity leaf hash; bottom inner level
//     SHA-256( LE64(32)||L || LE64(32)||R ); upper levels SHA-256( L || R ); heap order.
//
// There is no transpose: the codeword matrix is stored as np = 8 O coset planes [s][row][q]
// (column j = np q + s; O = 1 up to k = 2048) holding canonical integers, so lanes = adjacent q
// read adjacent 32-byte elements of one row -- coalesced row-major streaming.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace lg {

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }

// ---------------------------------------------------------------- Blake2s
__device__ __forceinline__ constexpr uint32_t b2s_iv(int i) {
    constexpr uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
    return IV[i];
}
__device__ __forceinline__ constexpr int b2s_sigma(int r, int i) {
    constexpr uint8_t S[10][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
    return S[r][i];
}

#define LG_B2S_G(a, b, c, d, x, y)        \
    a += b + (x); d = rotr32(d ^ a, 16);  \
    c += d;       b = rotr32(b ^ c, 12);  \
    a += b + (y); d = rotr32(d ^ a, 8);   \
    c += d;       b = rotr32(b ^ c, 7);

// one compression; t = byte counter after this block (fits 64 bits, high word passed separately)
__device__ __forceinline__ void b2s_compress(uint32_t (&h)[8], const uint32_t (&m)[16], uint32_t t_lo, uint32_t t_hi, bool last) {
    uint32_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    uint32_t v8 = b2s_iv(0), v9 = b2s_iv(1), v10 = b2s_iv(2), v11 = b2s_iv(3);
    uint32_t v12 = b2s_iv(4) ^ t_lo, v13 = b2s_iv(5) ^ t_hi, v14 = last ? ~b2s_iv(6) : b2s_iv(6), v15 = b2s_iv(7);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        LG_B2S_G(v0, v4, v8, v12, m[b2s_sigma(r, 0)], m[b2s_sigma(r, 1)])
        LG_B2S_G(v1, v5, v9, v13, m[b2s_sigma(r, 2)], m[b2s_sigma(r, 3)])
        LG_B2S_G(v2, v6, v10, v14, m[b2s_sigma(r, 4)], m[b2s_sigma(r, 5)])
        LG_B2S_G(v3, v7, v11, v15, m[b2s_sigma(r, 6)], m[b2s_sigma(r, 7)])
        LG_B2S_G(v0, v5, v10, v15, m[b2s_sigma(r, 8)], m[b2s_sigma(r, 9)])
        LG_B2S_G(v1, v6, v11, v12, m[b2s_sigma(r, 10)], m[b2s_sigma(r, 11)])
        LG_B2S_G(v2, v7, v8, v13, m[b2s_sigma(r, 12)], m[b2s_sigma(r, 13)])
        LG_B2S_G(v3, v4, v9, v14, m[b2s_sigma(r, 14)], m[b2s_sigma(r, 15)])
    }
    h[0] ^= v0 ^ v8;  h[1] ^= v1 ^ v9;  h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

// Parked state of one column between two launches: chaining value + the bytes of the current, still incomplete 64-byte block.
// After an EVEN number of rows those are 8 bytes (the tail of the last row, or the length prefix), after an odd number 40
// (8 + one whole row): kColStateVec uint4 per column = h[8] | m[0..9] | unused.  The same record travels between GPUs when a
// column is hashed by several ranks in turn (row-relay commit, lg_stage_hash_rows).
constexpr int kColStateVec = 5;

struct ColHashArgs {
    const uint4* u;         // coset planes [8][total_rows][k], canonical integers, 2 x uint4 per element
    uint8_t* leaves;        // [batch][n][32]
    uint4* state;           // [batch][np][k][kColStateVec]: chaining value + carried bytes between row chunks
    uint32_t rows;          // rows per proof (4m) of THIS matrix (addressing)
    uint32_t k;             // elements per plane row (ki)
    uint32_t lognp;         // log2 of the number of planes np (n = np * k)
    uint32_t proof_begin;   // first proof hashed by this launch
    uint32_t proof_count;   // proofs hashed by this launch
    uint32_t row_begin;     // rows [row_begin, row_end) of each proof are absorbed
    uint32_t row_end;
    uint32_t plane_begin;   // planes [plane_begin, plane_begin + plane_count) are hashed (all of them in a
    uint32_t plane_count;   // single-GPU commit; the planes a rank owns when a proof is coset-sharded)
    uint32_t first;         // 1: start from the initial state; 0: resume from `state`
    uint32_t last;          // 1: finalise and write the leaf digest; 0: save `state`
    uint64_t plane_stride;  // in elements
    // position of row_begin inside the COLUMN and the column's length: the same as row_begin / rows unless this matrix is a
    // row shard of a larger one (row-relay commit: a rank holds rows [col_pos, col_pos + nrows) of a column of col_rows)
    uint64_t col_pos;
    uint64_t col_rows;
};

// One lane per column.  Thread id -> (proof b, coset s, q) with q fastest, so a wave reads
// 64 adjacent elements (2 KiB contiguous) of one row per step.  A column can be absorbed in
// several launches over consecutive row ranges (the commit pipeline hashes a chunk of rows
// while the next chunk is still being encoded; the row-relay commit hands a column from GPU to
// GPU): the chaining value and the bytes of the incomplete block are parked in `state` in
// between.  Row ranges may start and end anywhere -- a 64-byte block is 8 carried bytes + one row +
// 24 bytes of the next, so an odd position carries 40 bytes instead of 8.
static __global__ void __launch_bounds__(256, 8) blake2s_columns_kernel(const ColHashArgs a) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = (uint64_t)a.proof_count * a.plane_count * a.k;
    if (gid >= total) return;
    const uint32_t q = (uint32_t)(gid % a.k);
    const uint32_t s = a.plane_begin + (uint32_t)((gid / a.k) % a.plane_count);
    const uint32_t b = a.proof_begin + (uint32_t)((gid / a.k) / a.plane_count);
    // element (row i) = p[i * 2k], p[i * 2k + 1]
    const uint4* p = a.u + 2 * ((uint64_t)s * a.plane_stride + ((uint64_t)b * a.rows + a.row_begin) * a.k + q);
#ifdef LG_AB_HASH_ALIASED_ROWS
    // ABLATION (tools/ab_hash_reads.sh, EXPERIMENTS.md section I): every row read of a column hits the column's FIRST row -- the same
    // instruction stream with the hash's pass over U served from cache instead of HBM (digests are wrong): the upper bound of what an
    // evaluate + hash fusion could save, without any of its costs.  (col_rows < 2^58: a zero the compiler cannot see.)
    const uint64_t step = 2 * (uint64_t)a.k * (a.col_rows >> 62);
#else
    const uint64_t step = 2 * (uint64_t)a.k;
#endif
    uint4* st = a.state + kColStateVec * ((((uint64_t)b << a.lognp) + s) * a.k + q);

    uint32_t h[8];
    uint32_t m[16];
    const bool odd_start = (a.col_pos & 1) != 0;   // wave-uniform
    if (a.first) {
#pragma unroll
        for (int i = 0; i < 8; i++) h[i] = b2s_iv(i);
        h[0] ^= 0x01010020u;
        m[0] = (uint32_t)a.col_rows;  // LE64(rows): serialize_compressed length prefix of Vec<F>
        m[1] = (uint32_t)(a.col_rows >> 32);
    } else {
        const uint4 s0 = st[0], s1 = st[1], s2 = st[2];
        h[0] = s0.x; h[1] = s0.y; h[2] = s0.z; h[3] = s0.w;
        h[4] = s1.x; h[5] = s1.y; h[6] = s1.z; h[7] = s1.w;
        m[0] = s2.x; m[1] = s2.y;
        if (odd_start) {
            const uint4 s3 = st[3], s4 = st[4];
            m[2] = s2.z; m[3] = s2.w; m[4] = s3.x; m[5] = s3.y; m[6] = s3.z; m[7] = s3.w; m[8] = s4.x; m[9] = s4.y;
        }
    }
    uint64_t t = (a.col_pos >> 1) * 64;  // bytes compressed so far (whole blocks)
    uint32_t nrows = a.row_end - a.row_begin;
    uint4 a0, a1, b0, b1;
    if (odd_start && nrows) {
        // the block in progress already holds 8 + 32 bytes: the first 24 bytes of this row complete it
        a0 = p[0]; a1 = p[1];
        m[10] = a0.x; m[11] = a0.y; m[12] = a0.z; m[13] = a0.w; m[14] = a1.x; m[15] = a1.y;
        t += 64;
        b2s_compress(h, m, (uint32_t)t, (uint32_t)(t >> 32), false);
        m[0] = a1.z;
        m[1] = a1.w;
        p += step;
        nrows--;
    }
    const uint32_t pairs = nrows >> 1;
    if (pairs) { a0 = p[0]; a1 = p[1]; b0 = p[step]; b1 = p[step + 1]; }
    for (uint32_t i = 0; i < pairs; i++) {
        m[2] = a0.x; m[3] = a0.y; m[4] = a0.z; m[5] = a0.w; m[6] = a1.x; m[7] = a1.y; m[8] = a1.z; m[9] = a1.w;
        m[10] = b0.x; m[11] = b0.y; m[12] = b0.z; m[13] = b0.w; m[14] = b1.x; m[15] = b1.y;
        const uint32_t c0 = b1.z, c1 = b1.w;
        p += 2 * step;
        if (i + 1 < pairs) { a0 = p[0]; a1 = p[1]; b0 = p[step]; b1 = p[step + 1]; }  // prefetch next pair
        t += 64;
        b2s_compress(h, m, (uint32_t)t, (uint32_t)(t >> 32), false);
        m[0] = c0;
        m[1] = c1;
    }
    const bool odd_end = (nrows & 1) != 0;   // one more row, at an even position: it joins the carried bytes
    if (odd_end) {
        a0 = p[0]; a1 = p[1];
        m[2] = a0.x; m[3] = a0.y; m[4] = a0.z; m[5] = a0.w; m[6] = a1.x; m[7] = a1.y; m[8] = a1.z; m[9] = a1.w;
    }
    if (!a.last) {
        st[0] = make_uint4(h[0], h[1], h[2], h[3]);
        st[1] = make_uint4(h[4], h[5], h[6], h[7]);
        if (odd_end) {
            st[2] = make_uint4(m[0], m[1], m[2], m[3]);
            st[3] = make_uint4(m[4], m[5], m[6], m[7]);
            st[4] = make_uint4(m[8], m[9], 0, 0);
        } else {
            st[2] = make_uint4(m[0], m[1], 0, 0);
        }
        return;
    }
    if (odd_end) {
#pragma unroll
        for (int i = 10; i < 16; i++) m[i] = 0;
        t += 40;
    } else {
#pragma unroll
        for (int i = 2; i < 16; i++) m[i] = 0;
        t += 8;
    }
    b2s_compress(h, m, (uint32_t)t, (uint32_t)(t >> 32), true);
    uint4* out = reinterpret_cast<uint4*>(a.leaves + 32 * ((((uint64_t)b * a.k + q) << a.lognp) + s));
    out[0] = make_uint4(h[0], h[1], h[2], h[3]);
    out[1] = make_uint4(h[4], h[5], h[6], h[7]);
}

// ---------------------------------------------------------------- Blake2s, four lanes per column
// For FEW columns (one Poseidon proof: 1024 columns = 16 waves on 1024 SIMDs) the one-lane-per-column kernel above is
// a pure latency chain: 173 blocks x 996 dependent-ish instructions at one wave per SIMD, 0.33 of the 0.39 ms of a single
// commitment.  Here a QUAD of lanes shares one column: lane l holds column l of the 4 x 4 state (v[l], v[4+l], v[8+l],
// v[12+l]), so the four G functions of a half-round run side by side.  The diagonal half-round mixes the columns:
// rotations inside the quad, done as DPP quad_perm operands of the instructions that consume them (and the rotation back
// rides on the first uses of the next column half-round), so no lane exchange is a separate instruction.  The 64-byte block is staged through LDS (each lane loads 16 bytes of the two rows it covers) and
// every lane fetches the two message words of its G with ds_read_b32 from per-lane addresses that are loop invariant
// (sigma is applied once, when the 40 addresses are built): 240 VALU + 40 LDS reads per lane and block instead of 996 VALU.
// Throughput per column is lower than the one-lane kernel's (four lanes do 1.7x the instructions of one), so the host
// picks this kernel only when the one-lane form cannot fill the machine.
struct B2sQuadSigma {
    uint8_t w[10][4][4];   // [round][lane][column x, column y, diagonal x, diagonal y] -> message word index
};
__host__ __device__ constexpr B2sQuadSigma b2s_quad_sigma() {
    B2sQuadSigma t{};
    for (int r = 0; r < 10; r++)
        for (int l = 0; l < 4; l++) {
            t.w[r][l][0] = (uint8_t)b2s_sigma(r, 2 * l);
            t.w[r][l][1] = (uint8_t)b2s_sigma(r, 2 * l + 1);
            t.w[r][l][2] = (uint8_t)b2s_sigma(r, 8 + 2 * ((l + 3) % 4));   // lane l works on diagonal l - 1 (see the kernel)
            t.w[r][l][3] = (uint8_t)b2s_sigma(r, 9 + 2 * ((l + 3) % 4));
        }
    return t;
}
__constant__ const B2sQuadSigma kB2sQuadSigma = b2s_quad_sigma();
__constant__ const uint32_t kB2sIv[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};

// quad_perm:[1,2,3,0] / [2,3,0,1] / [3,0,1,2]: lane l reads lane (l + 1, 2, 3) % 4 of its quad
template <int CTRL>
__device__ __forceinline__ uint32_t quad_rot(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true);
}
constexpr int kQuadRot1 = 0x39, kQuadRot2 = 0x4E, kQuadRot3 = 0x93;

// a wave's LDS operations execute in order; this only stops the compiler from moving reads across the writes
__device__ __forceinline__ void quad_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
constexpr int kQuadSlotWords = 20;    // per slot: 2 pad | 2 carried words | 16 words of two rows; the block is words 2..17
constexpr int kQuadStrideWords = 44;  // two slots + pad: 44 c mod 32 is distinct for the 8 quads of a 32-lane group

struct ColHashQuadArgs {
    const uint4* u;         // coset planes, canonical integers, 2 x uint4 per element
    uint8_t* leaves;        // [batch][n][32]
    uint32_t rows, k, lognp;   // rows: rows per proof of THIS matrix (addressing)
    uint32_t proof_begin, proof_count;
    uint32_t plane_begin, plane_count;
    uint64_t plane_stride;  // elements
    // Row ranges (round 3), with the same parked state as the one-lane kernel (kColStateVec uint4 per column): rows
    // [row_begin, row_end) are absorbed; they are rows [col_pos, ...) of columns of col_rows rows.  col_pos must be EVEN and,
    // unless `last`, so must the number of rows (the host falls back to the one-lane kernel otherwise).  A whole-column launch
    // is row_begin = 0, row_end = rows, col_pos = 0, col_rows = rows, first = last = 1.
    uint4* state;
    uint32_t row_begin, row_end;
    uint32_t first, last;
    uint64_t col_pos, col_rows;
};

static __global__ void __launch_bounds__(256) blake2s_columns_quad_kernel(const ColHashQuadArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * kQuadStrideWords];
    const uint32_t lane = threadIdx.x & 3, quad = threadIdx.x >> 2;
    const uint64_t col = (uint64_t)blockIdx.x * 64 + quad;
    const uint64_t total = (uint64_t)a.proof_count * a.plane_count * a.k;
    const bool active = col < total;                       // whole quads are active or not
    const uint64_t cc = active ? col : 0;
    const uint32_t q = (uint32_t)(cc % a.k);
    const uint32_t s = a.plane_begin + (uint32_t)((cc / a.k) % a.plane_count);
    const uint32_t b = a.proof_begin + (uint32_t)((cc / a.k) / a.plane_count);
    // lane l fetches 16 bytes: rows (2 blk + l / 2), half (l % 2)
    const uint4* p = a.u + 2 * ((uint64_t)s * a.plane_stride + ((uint64_t)b * a.rows + a.row_begin + (lane >> 1)) * a.k + q) + (lane & 1);
    uint32_t* st = reinterpret_cast<uint32_t*>(a.state + kColStateVec * ((((uint64_t)b << a.lognp) + s) * a.k + q));
    const uint64_t step = 4 * (uint64_t)a.k;               // two rows, in uint4
    uint32_t* slot0 = lds + quad * kQuadStrideWords;
    // loop-invariant LDS addresses of this lane's message words (slot 0; slot 1 = + kQuadSlotWords words)
    const uint32_t* mw[10][4];
#pragma unroll
    for (int r = 0; r < 10; r++)
#pragma unroll
        for (int j = 0; j < 4; j++) mw[r][j] = slot0 + 2 + kB2sQuadSigma.w[r][lane][j];
    const uint32_t iv_lo = kB2sIv[lane], iv_hi = kB2sIv[4 + lane];
    uint32_t h_lo = iv_lo ^ (lane == 0 ? 0x01010020u : 0u), h_hi = iv_hi;
    if (!a.first && active) { h_lo = st[lane]; h_hi = st[4 + lane]; }     // resume: lane l holds h[l] and h[4 + l]

    const uint32_t rows = a.row_end - a.row_begin;
    // finalising launch: rows even -> one more block with only the 8 carried bytes; otherwise whole blocks of two rows
    const uint32_t nblk = a.last ? ((rows & 1) ? (rows + 1) / 2 : rows / 2 + 1) : rows / 2;
    const uint64_t total_len = 8 + 32 * a.col_rows;
    const uint64_t blk0 = a.col_pos >> 1;                                   // blocks compressed before this launch
    auto fetch = [&](uint32_t blk) -> uint4 {
        const uint32_t row = 2 * blk + (lane >> 1);
        return (active && row < rows) ? p[(uint64_t)blk * step] : make_uint4(0, 0, 0, 0);
    };
    // store the rows of block blk into slot blk & 1 and the 8 bytes they carry over into the other slot
    auto stage = [&](uint32_t blk, const uint4& v) {
        uint32_t* sl = slot0 + (blk & 1) * kQuadSlotWords;
        *reinterpret_cast<uint4*>(sl + 4 + 4 * lane) = v;
        if (lane == 3) {
            uint32_t* other = slot0 + ((blk & 1) ^ 1) * kQuadSlotWords;
            other[2] = v.z;
            other[3] = v.w;
        }
    };
    if (lane == 0) {
        if (a.first) { slot0[2] = (uint32_t)a.col_rows; slot0[3] = (uint32_t)(a.col_rows >> 32); }   // LE64(rows): serialize_compressed length prefix of Vec<F>
        else if (active) { slot0[2] = st[8]; slot0[3] = st[9]; }                                     // the 8 bytes carried over from the last row absorbed
    }
    uint4 cur = fetch(0);
    stage(0, cur);
    uint4 nxt = fetch(1);
    auto block = [&](uint32_t blk, auto slot_c) {
        constexpr int so = decltype(slot_c)::value * kQuadSlotWords;   // compile-time slot: the 40 reads use immediate offsets
        const bool last = a.last && blk + 1 == nblk;
        const uint64_t t = last ? total_len : 64 * (blk0 + blk + 1);
        uint32_t va = h_lo, vb = h_hi, vc = iv_lo;
        const uint32_t tw = lane == 0 ? (uint32_t)t : (lane == 1 ? (uint32_t)(t >> 32) : ((lane == 2 && last) ? 0xffffffffu : 0u));
        uint32_t vd = iv_hi ^ tw;
        quad_lds_order();
        // all 40 message words of this lane up front: the reads are independent of the chain below, so their latency
        // hides behind the first rounds instead of being paid four words at a time in every round
        uint32_t mr[10][4];
#pragma unroll
        for (int r = 0; r < 10; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) mr[r][j] = mw[r][j][so];
#pragma unroll
        for (int r = 0; r < 10; r++) {
            const uint32_t x0 = mr[r][0], y0 = mr[r][1], x1 = mr[r][2], y1 = mr[r][3];
            // column half-round.  Between the half-rounds a, c, d travel and b stays: lane l then works on diagonal l - 1,
            // G(v[l-1], v[4+l], v[8+l+1], v[12+l+2]) -- b is the LAST value a G produces and a, d, c the oldest, so the DPP
            // operands below are never the result of the instruction just before them (a DPP read of a fresh VALU result costs
            // two wait states).  From the second round on va, vc, vd arrive rotated by the previous diagonal half-round and are
            // rotated back on their first use.
            if (r > 0) {
                va = quad_rot<kQuadRot1>(va) + x0 + vb;              // v[l] sits in lane l + 1
                vd = rotr32(quad_rot<kQuadRot2>(vd) ^ va, 16);       // v[12 + l] in lane l - 2
                vc = quad_rot<kQuadRot3>(vc) + vd;                   // v[8 + l] in lane l - 1
            } else {
                va = va + x0 + vb;
                vd = rotr32(vd ^ va, 16);
                vc = vc + vd;
            }
            vb = rotr32(vb ^ vc, 12);
            va = va + vb + y0;
            vd = rotr32(vd ^ va, 8);
            vc = vc + vd;
            vb = rotr32(vb ^ vc, 7);
            // diagonal half-round: lane l takes a from lane l - 1, c from lane l + 1, d from lane l + 2
            va = quad_rot<kQuadRot3>(va) + x1 + vb;
            vd = rotr32(quad_rot<kQuadRot2>(vd) ^ va, 16);
            vc = quad_rot<kQuadRot1>(vc) + vd;
            vb = rotr32(vb ^ vc, 12);
            va = va + vb + y1;
            vd = rotr32(vd ^ va, 8);
            vc = vc + vd;
            vb = rotr32(vb ^ vc, 7);
        }
        // undo the last diagonal rotation and feed forward: h[l] ^= v[l] ^ v[8 + l], h[4 + l] ^= v[4 + l] ^ v[12 + l]
        h_lo ^= quad_rot<kQuadRot1>(va) ^ quad_rot<kQuadRot3>(vc);
        h_hi ^= vb ^ quad_rot<kQuadRot2>(vd);
        if (blk + 1 < nblk) {
            quad_lds_order();
            stage(blk + 1, nxt);
            nxt = fetch(blk + 2);
        }
    };
    for (uint32_t blk = 0; blk < nblk; blk += 2) {
        block(blk, std::integral_constant<int, 0>{});
        if (blk + 1 < nblk) block(blk + 1, std::integral_constant<int, 1>{});
    }
    if (!active) return;
    if (!a.last) {
        // park: h[l] / h[4 + l] from lane l, the 8 carried bytes (staged into the slot the next block would have used) from lane 0
        st[lane] = h_lo;
        st[4 + lane] = h_hi;
        quad_lds_order();
        if (lane == 0) {
            const uint32_t* nx = slot0 + (nblk & 1) * kQuadSlotWords;
            st[8] = nx[2];
            st[9] = nx[3];
        }
        return;
    }
    uint32_t* out = reinterpret_cast<uint32_t*>(a.leaves + 32 * ((((uint64_t)b * a.k + q) << a.lognp) + s));
    out[lane] = h_lo;
    out[4 + lane] = h_hi;
}

// ---------------------------------------------------------------- SHA-256
__device__ __forceinline__ constexpr uint32_t sha_k(int i) {
    constexpr uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    return K[i];
}
__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// one SHA-256 block; w[] holds the 16 big-endian message words and is clobbered
__device__ __forceinline__ void sha256_block(uint32_t (&st)[8], uint32_t (&w)[16]) {
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + sha_k(i) + w[i & 15];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

__device__ __forceinline__ void sha256_init(uint32_t (&st)[8]) {
    st[0] = 0x6a09e667u; st[1] = 0xbb67ae85u; st[2] = 0x3c6ef372u; st[3] = 0xa54ff53au;
    st[4] = 0x510e527fu; st[5] = 0x9b05688cu; st[6] = 0x1f83d9abu; st[7] = 0x5be0cd19u;
}

// parent = SHA-256 over two 32-byte children (LEAF: each child prefixed by LE64(32))
template <bool LEAF>
__device__ __forceinline__ void sha256_two_to_one(const uint4* left, const uint4* right, uint4* out) {
    uint4 l0 = left[0], l1 = left[1], r0 = right[0], r1 = right[1];
    uint32_t L[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
    uint32_t R[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
    uint32_t st[8], w[16];
    sha256_init(st);
    if constexpr (LEAF) {
        // 80-byte message: 20 00 00 00 00 00 00 00 | L | 20 00 .. | R
        w[0] = 0x20000000u; w[1] = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) w[2 + i] = bswap32(L[i]);
        w[10] = 0x20000000u; w[11] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) w[12 + i] = bswap32(R[i]);
        sha256_block(st, w);
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = bswap32(R[4 + i]);
        w[4] = 0x80000000u;
#pragma unroll
        for (int i = 5; i < 15; i++) w[i] = 0;
        w[15] = 80 * 8;
        sha256_block(st, w);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) { w[i] = bswap32(L[i]); w[8 + i] = bswap32(R[i]); }
        sha256_block(st, w);
        w[0] = 0x80000000u;
#pragma unroll
        for (int i = 1; i < 15; i++) w[i] = 0;
        w[15] = 64 * 8;
        sha256_block(st, w);
    }
    out[0] = make_uint4(bswap32(st[0]), bswap32(st[1]), bswap32(st[2]), bswap32(st[3]));
    out[1] = make_uint4(bswap32(st[4]), bswap32(st[5]), bswap32(st[6]), bswap32(st[7]));
}

// sibling leaf + authentication path (root side first) of opened columns: MerkleTree::generate_proof pieces (mod.rs:951);
// one thread per 32-byte digest, blockIdx.y = proof within the call
struct GatherPathArgs {
    const uint8_t* leaves;   // [batch][n][32]
    const uint8_t* nodes;    // [batch][n-1][32]
    const uint32_t* idx;     // [proofs][t]
    uint8_t* sib;            // [proofs][t][32]
    uint8_t* paths;          // [proofs][t][logn-1][32]
    uint32_t n, logn, t, proof0;
};
static __global__ void __launch_bounds__(256) gather_paths_kernel(GatherPathArgs a) {
    const uint32_t p = blockIdx.y, plen = a.logn - 1;
    const uint64_t h = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= (uint64_t)a.t * (plen + 1)) return;
    const uint8_t* leaves = a.leaves + 32 * (uint64_t)(a.proof0 + p) * a.n;
    const uint8_t* nodes = a.nodes + 32 * (uint64_t)(a.proof0 + p) * (a.n - 1);
    const uint32_t c = (uint32_t)(h / (plen + 1)), lvl = (uint32_t)(h % (plen + 1));
    const uint32_t j = a.idx[(uint64_t)p * a.t + c];
    const uint4* src;
    uint4* dst;
    if (lvl == plen) {  // leaf sibling
        src = reinterpret_cast<const uint4*>(leaves + 32 * (uint64_t)(j ^ 1));
        dst = reinterpret_cast<uint4*>(a.sib + 32 * ((uint64_t)p * a.t + c));
    } else {            // auth_path[lvl]: sibling of the ancestor at depth lvl + 1
        const uint32_t depth = lvl + 1;
        const uint32_t anc = j >> (a.logn - depth);
        const uint32_t node = ((1u << depth) - 1) + (anc ^ 1);
        src = reinterpret_cast<const uint4*>(nodes + 32 * (uint64_t)node);
        dst = reinterpret_cast<uint4*>(a.paths + 32 * (((uint64_t)p * a.t + c) * plen + lvl));
    }
    dst[0] = src[0];
    dst[1] = src[1];
}

struct MerkleArgs {
    const uint8_t* leaves;  // [batch][n][32]
    uint8_t* nodes;         // [batch][n-1][32], heap order, root at 0
    uint32_t n;             // leaves per tree (power of two >= 2)
    uint32_t logn;
    uint32_t batch;
    uint32_t in_depth;      // depth of the input level (logn: the leaves; otherwise an inner level already in `nodes`)
    uint32_t chunks;        // workgroups per tree = max(1, 2^in_depth / 512)
};

// Each workgroup takes 512 consecutive nodes of the input level (all of it when the level is
// smaller) and reduces them as far as they go -- up to nine levels -- through LDS, writing every
// produced node to its heap slot.  A 2^15-leaf tree is two launches instead of fifteen.
template <bool LEAF>
__global__ void __launch_bounds__(256) merkle_subtree_kernel(const MerkleArgs a) {
    __shared__ uint4 buf[2][256][2];
    const uint32_t b = blockIdx.x / a.chunks, chunk = blockIdx.x % a.chunks;
    if (b >= a.batch) return;
    uint8_t* tree = a.nodes + 32 * (uint64_t)b * (a.n - 1);
    const uint32_t in_count = (a.in_depth >= 9) ? 512u : (1u << a.in_depth);  // inputs handled by this workgroup
    uint32_t depth = a.in_depth - 1;                                         // depth of the level being produced
    uint32_t count = in_count >> 1;                                          // nodes this workgroup produces at `depth`
    const uint32_t t = threadIdx.x;
    // first level: children come from global memory
    if (t < count) {
        const uint32_t i = chunk * count + t;  // index within the level
        const uint4 *l, *r;
        if constexpr (LEAF) {
            const uint8_t* lv = a.leaves + 32 * ((uint64_t)b * a.n + 2 * (uint64_t)i);
            l = reinterpret_cast<const uint4*>(lv);
            r = reinterpret_cast<const uint4*>(lv + 32);
        } else {
            const uint32_t child = (1u << a.in_depth) - 1 + 2 * i;
            l = reinterpret_cast<const uint4*>(tree + 32 * (uint64_t)child);
            r = reinterpret_cast<const uint4*>(tree + 32 * (uint64_t)(child + 1));
        }
        uint4 out[2];
        sha256_two_to_one<LEAF>(l, r, out);
        uint4* dst = reinterpret_cast<uint4*>(tree + 32 * (uint64_t)((1u << depth) - 1 + i));
        dst[0] = out[0];
        dst[1] = out[1];
        buf[0][t][0] = out[0];
        buf[0][t][1] = out[1];
    }
    int cur = 0;
    while (count > 1) {
        __syncthreads();
        count >>= 1;
        depth -= 1;
        if (t < count) {
            uint4 out[2];
            sha256_two_to_one<false>(&buf[cur][2 * t][0], &buf[cur][2 * t + 1][0], out);
            const uint32_t i = chunk * count + t;
            uint4* dst = reinterpret_cast<uint4*>(tree + 32 * (uint64_t)((1u << depth) - 1 + i));
            dst[0] = out[0];
            dst[1] = out[1];
            buf[cur ^ 1][t][0] = out[0];
            buf[cur ^ 1][t][1] = out[1];
        }
        cur ^= 1;
    }
}

}  // namespace lg
