// Row-combination kernels behind the three sub-proof polynomials of NP-Eng/ligero
// (SURVEY.md section 8f #1-2), working on the data the commitment left resident in HBM:
//
//   prove_interleaved            src/ligero/mod.rs:658      preenc_u.row_mul(r)   (src/matrices/mod.rs:138-149)
//   prove_linear_constraints     src/ligero/mod.rs:731-736  sum_i u_i * r_i       (polynomial products)
//   prove_quadratic_constraints  src/ligero/mod.rs:845-848  sum_i r_i (p_x_i p_y_i - p_z_i)
//
// The polynomial products are done in the evaluation domain of size 2k.  Its points are
// omega_n^(4j), i.e. codeword indices 4j: those evaluations of every u_i are ALREADY in the
// codeword planes s = 0 (mod 4) the commitment keeps (canonical integers), so nothing is
// re-encoded for u; each output point is a sum over rows -- a column-wise reduction, lanes =
// adjacent slots, coalesced -- followed by one size-2k inverse NTT.  Exact field arithmetic
// makes the coefficients identical to the reference's.
#pragma once
#include "fr_gfx950.h"

namespace lg {

// All kernels take the proof index from blockIdx.z: one launch serves every proof of a batch.
struct RowSumArgs {
    const fr* a;            // operand A: element (proof p, row i, col c) at a[p * a_proof + i * a_row + c * a_col]
    const fr* b;            // operand B likewise; null => b = r[p * rows + i] (per-row scalar)
    const fr* r;            // per-row scalars (Montgomery), used when b == null
    fr* partial;            // [batch][nchunks][cols] lazy sums
    uint64_t a_proof, a_row, a_col, b_proof, b_row, b_col;
    uint32_t rows, cols, rows_per_chunk, nchunks;
};

// partial[p][chunk][c] = sum_{i in chunk} A[p][i][c] (*) B[p][i][c]     ((*) = Montgomery product)
static __global__ void __launch_bounds__(256) rowsum_mul_kernel(RowSumArgs a) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t chunk = blockIdx.y, p = blockIdx.z;
    if (c >= a.cols) return;
    a.a += (uint64_t)p * a.a_proof;
    if (a.b != nullptr) a.b += (uint64_t)p * a.b_proof;
    a.r += (uint64_t)p * a.rows;
    a.partial += (uint64_t)p * a.nchunks * a.cols;
    const uint32_t i0 = chunk * a.rows_per_chunk;
    const uint32_t i1 = (i0 + a.rows_per_chunk < a.rows) ? i0 + a.rows_per_chunk : a.rows;
    fr acc;
#pragma unroll
    for (int l = 0; l < 8; l++) acc.v[l] = 0;
    for (uint32_t i = i0; i < i1; i++) {
        const fr x = fr_load(a.a + (uint64_t)i * a.a_row + (uint64_t)c * a.a_col);
        const fr y = (a.b != nullptr) ? fr_load(a.b + (uint64_t)i * a.b_row + (uint64_t)c * a.b_col) : fr_load(a.r + i);
        fr t;
        fr_mul_lazy(t, x, y);
        fr_add_lazy(acc, acc, t);
    }
    fr_store(a.partial + (uint64_t)chunk * a.cols + c, acc);
}

struct QuadSumArgs {
    const fr* u;            // one codeword plane [batch * 4m][ki], canonical; each proof's rows are [X; Y; Z; W]
    const fr* r;            // [batch][m] challenges, Montgomery
    fr* partial;            // [batch][nchunks][ki]
    fr r2;                  // 2^512 mod p
    uint32_t m, ki, rows_per_chunk, nchunks;
};

// partial[p][chunk][q] = sum_{i in chunk} r_i (x_i y_i - z_i) as a plain integer (not Montgomery)
static __global__ void __launch_bounds__(256) quadsum_kernel(QuadSumArgs a) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t chunk = blockIdx.y, p = blockIdx.z;
    if (q >= a.ki) return;
    a.u += (uint64_t)p * 4 * a.m * a.ki;
    a.r += (uint64_t)p * a.m;
    a.partial += (uint64_t)p * a.nchunks * a.ki;
    const uint32_t i0 = chunk * a.rows_per_chunk;
    const uint32_t i1 = (i0 + a.rows_per_chunk < a.m) ? i0 + a.rows_per_chunk : a.m;
    fr acc;
#pragma unroll
    for (int l = 0; l < 8; l++) acc.v[l] = 0;
    for (uint32_t i = i0; i < i1; i++) {
        const fr x = fr_load(a.u + (uint64_t)i * a.ki + q);
        const fr y = fr_load(a.u + ((uint64_t)a.m + i) * a.ki + q);
        const fr z = fr_load(a.u + (2 * (uint64_t)a.m + i) * a.ki + q);
        const fr ri = fr_load(a.r + i);  // r * 2^256
        fr rr, xy, t1, t2, d;
        fr_mul_lazy(rr, ri, a.r2);       // r * 2^512
        fr_mul_lazy(xy, x, y);           // x y / 2^256
        fr_mul_lazy(t1, xy, rr);         // x y r
        fr_mul_lazy(t2, z, ri);          // z r
        fr_sub_lazy(d, t1, t2);
        fr_add_lazy(acc, acc, d);
    }
    fr_store(a.partial + (uint64_t)chunk * a.ki + q, acc);
}

// out[p * out_proof + c * out_stride + out_off] = (sum_chunks partial[p][chunk][c]) (*) post, fully reduced
static __global__ void __launch_bounds__(256) rowsum_finish_kernel(const fr* partial, uint32_t nchunks, uint32_t cols, fr post, fr* out,
                                                           uint32_t out_stride, uint32_t out_off, uint64_t out_proof) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    partial += (uint64_t)blockIdx.z * nchunks * cols;
    out += (uint64_t)blockIdx.z * out_proof;
    fr acc;
#pragma unroll
    for (int l = 0; l < 8; l++) acc.v[l] = 0;
    for (uint32_t j = 0; j < nchunks; j++) {
        const fr t = fr_load(partial + (uint64_t)j * cols + c);
        fr_add_lazy(acc, acc, t);
    }
    fr m, z;
    fr_mul_lazy(m, acc, post);
    fr_reduce(z, m);
    fr_store(out + (uint64_t)c * out_stride + out_off, z);
}

}  // namespace lg
