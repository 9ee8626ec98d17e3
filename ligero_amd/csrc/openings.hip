// Reads of the resident commitment: open_columns (src/ligero/mod.rs:935-955: u.column(i) + generate_proof), codeword rows,
// and the row operators reed_solomon / reed_solomon_interpolate / reed_solomon_evaluate (mod.rs:998-1012) on scratch buffers.
#include "lg_context.h"

namespace lg {

struct GatherArgs {
    const fr* u;             // coset planes, canonical
    const uint8_t* leaves;   // [n][32] of the proof
    const uint8_t* nodes;    // [n-1][32] of the proof
    const uint32_t* idx;     // [t]
    fr* cols;                // [t][rows] Montgomery
    uint8_t* sib;            // [t][32]
    uint8_t* paths;          // [t][logn-1][32]
    fr r2;
    uint64_t plane_stride;
    uint64_t row_base;       // proof * rows
    uint32_t rows, k, n, logn, t;  // k = plane row length ki
    uint32_t lognp;                // log2 of the number of planes
    uint32_t proof0;               // blockIdx.y = p serves proof proof0 + p: inputs/outputs advance by one proof each
    const uint32_t* slot;          // null: column c of proof p goes to cols[p][c]; else [proofs][t]: to cols[slot] of ONE region shared by
                                   // the call's proofs, 0xffffffff = not gathered (the throughput prover's compact openings)
};

// u.column(i) for the opened indices (src/matrices/mod.rs:169-171) + generate_proof pieces
__global__ void __launch_bounds__(256) gather_columns_kernel(GatherArgs a) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t ncol_elems = (uint64_t)a.t * a.rows;
    {
        const uint32_t p = blockIdx.y, plen = a.logn - 1;
        a.leaves += 32 * (uint64_t)(a.proof0 + p) * a.n;
        a.nodes += 32 * (uint64_t)(a.proof0 + p) * (a.n - 1);
        a.row_base = (uint64_t)(a.proof0 + p) * a.rows;
        a.idx += (uint64_t)p * a.t;
        if (a.slot) a.slot += (uint64_t)p * a.t;
        else a.cols += (uint64_t)p * ncol_elems;
        a.sib += 32 * (uint64_t)p * a.t;
        a.paths += 32 * (uint64_t)p * a.t * plen;
    }
    if (gid < ncol_elems) {
        const uint32_t c = (uint32_t)(gid / a.rows), i = (uint32_t)(gid % a.rows);
        uint64_t dst = gid;
        if (a.slot) {
            const uint32_t sl = a.slot[c];
            if (sl == 0xffffffffu) return;
            dst = (uint64_t)sl * a.rows + i;
        }
        const uint32_t j = a.idx[c];
        const uint32_t s = j & ((1u << a.lognp) - 1), q = j >> a.lognp;
        fr x = fr_load(a.u + (uint64_t)s * a.plane_stride + (a.row_base + i) * a.k + q);
        fr y, z;
        fr_mul_lazy(y, x, a.r2);
        fr_reduce(z, y);
        fr_store(a.cols + dst, z);
        return;
    }
    const uint64_t h = gid - ncol_elems;
    const uint32_t plen = a.logn - 1;
    if (h >= (uint64_t)a.t * (plen + 1)) return;
    const uint32_t c = (uint32_t)(h / (plen + 1)), lvl = (uint32_t)(h % (plen + 1));
    const uint32_t j = a.idx[c];
    const uint4* src;
    uint4* dst;
    if (lvl == plen) {  // leaf sibling
        src = reinterpret_cast<const uint4*>(a.leaves + 32 * (uint64_t)(j ^ 1));
        dst = reinterpret_cast<uint4*>(a.sib + 32 * (uint64_t)c);
    } else {  // auth_path[lvl], root side first: sibling of the ancestor at depth lvl+1
        const uint32_t depth = lvl + 1;
        const uint32_t anc = j >> (a.logn - depth);
        const uint32_t node = ((1u << depth) - 1) + (anc ^ 1);
        src = reinterpret_cast<const uint4*>(a.nodes + 32 * (uint64_t)node);
        dst = reinterpret_cast<uint4*>(a.paths + 32 * ((uint64_t)c * plen + lvl));
    }
    dst[0] = src[0];
    dst[1] = src[1];
}

// planes (canonical) -> natural column order rows (Montgomery): out[i][np q + s]; k = plane row length
__global__ void __launch_bounds__(256) planes_to_rows_kernel(const fr* u, uint64_t plane_stride, uint64_t row_base,
                                                            uint32_t nrows, uint32_t k, uint32_t lognp, fr r2, fr* out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = ((uint64_t)nrows * k) << lognp;
    if (gid >= total) return;
    const uint32_t q = (uint32_t)(gid % k);
    const uint32_t s = (uint32_t)((gid / k) & ((1u << lognp) - 1));
    const uint64_t i = (gid / k) >> lognp;
    fr x = fr_load(u + (uint64_t)s * plane_stride + (row_base + i) * k + q);
    fr y, z;
    fr_mul_lazy(y, x, r2);
    fr_reduce(z, y);
    fr_store(out + ((i * (uint64_t)k) << lognp) + ((uint64_t)q << lognp) + s, z);
}

}  // namespace lg

extern "C" {

int lg_read_codeword_rows(lg_ctx* c, uint32_t proof, uint32_t row0, uint32_t nrows, uint64_t* out) {
    if (!c || !out) return LG_ERR_BAD_ARG;
    if (c->gf) {
        if (!gf_committed(c->gf)) return LG_ERR_STATE;
        if (proof >= c->batch || (uint64_t)row0 + nrows > c->rows) return LG_ERR_BAD_ARG;
        LG_HIP(c, hipSetDevice(c->device));
        return gf_read_codeword_rows(c->gf, proof, row0, nrows, out);
    }
    if (!c->held.committed) return LG_ERR_STATE;
    if (proof >= c->batch || (uint64_t)row0 + nrows > c->rows) return LG_ERR_BAD_ARG;
    if (nrows == 0) return LG_OK;
    { const int rc_ = need_planes(c, all_planes_mask(c), "lg_read_codeword_rows"); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    const size_t elems = (size_t)nrows * c->n;
    { const int rc_ = settle_open_copy(c); if (rc_ != LG_OK) return rc_; }
    int rc = grow(c, &c->scr.c, &c->scr.c_elems, elems);
    if (rc != LG_OK) return rc;
    const uint64_t threads = elems;
    hipLaunchKernelGGL(lg::planes_to_rows_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->st.main, c->d_u,
                       c->total_rows * c->ki, (uint64_t)proof * c->rows + row0, nrows, c->ki, (uint32_t)c->lognp, c->tab.r2, c->scr.c);
    LG_HIP(c, hipGetLastError());
    return read_back(c, out, c->scr.c, elems * sizeof(fr));
}

}  // extern "C"

// the gather itself: t columns (+ sibling digests and paths) of nproofs consecutive proofs, indices and outputs on the device
int gather_columns_launch(lg_ctx* c, uint32_t proof0, uint32_t nproofs, const uint32_t* d_idx, uint32_t t, fr* d_cols, uint8_t* d_sib, uint8_t* d_paths,
                          const uint32_t* d_slot) {
    lg::GatherArgs g;
    memset(&g, 0, sizeof(g));
    g.u = c->d_u;
    g.leaves = c->d_leaves;
    g.nodes = c->d_nodes;
    g.idx = d_idx;
    g.cols = d_cols;
    g.sib = d_sib;
    g.paths = d_paths;
    g.slot = d_slot;
    g.r2 = c->tab.r2;
    g.plane_stride = c->total_rows * c->ki;
    g.lognp = (uint32_t)c->lognp;
    g.proof0 = proof0;
    g.rows = c->rows; g.k = c->ki; g.n = c->n; g.logn = (uint32_t)c->logn; g.t = t;
    const uint32_t plen = (uint32_t)c->logn - 1;
    const uint64_t threads = (uint64_t)t * c->rows + (uint64_t)t * (plen + 1);
    LG_LAUNCH(c, lg::gather_columns_kernel, dim3((uint32_t)((threads + 255) / 256), nproofs), dim3(256), 0, c->st.main, g);
    return LG_OK;
}

// opens t columns of each of `nproofs` consecutive proofs starting at `proof0` (one launch)
static int open_columns_impl(lg_ctx* c, uint32_t proof0, uint32_t nproofs, const uint32_t* idx, uint32_t t, uint64_t* cols_out, uint8_t* sib_out,
                             uint8_t* paths_out, bool wait = true) {
    if (!c || !idx || !cols_out || !sib_out || (!paths_out && c->logn > 1)) return LG_ERR_BAD_ARG;
    if (c->gf) {
        if (!gf_committed(c->gf)) return LG_ERR_STATE;
        if ((uint64_t)proof0 + nproofs > c->batch) return LG_ERR_BAD_ARG;
        for (size_t i = 0; i < (size_t)nproofs * t; i++)
            if (idx[i] >= c->n) return LG_ERR_BAD_ARG;
        if ((size_t)nproofs * t == 0) return LG_OK;
        LG_HIP(c, hipSetDevice(c->device));
        return gf_open_columns(c->gf, proof0, nproofs, idx, t, cols_out, sib_out, paths_out);
    }
    if (!c->held.committed) return LG_ERR_STATE;
    if ((uint64_t)proof0 + nproofs > c->batch) return LG_ERR_BAD_ARG;
    const size_t nidx = (size_t)nproofs * t;
    uint32_t touched = 0;
    for (size_t i = 0; i < nidx; i++) {
        if (idx[i] >= c->n) return LG_ERR_BAD_ARG;
        touched |= 1u << (idx[i] & (c->nplanes - 1));
    }
    if (nidx == 0) return LG_OK;
    { const int rc_ = need_planes(c, touched, "lg_open_columns"); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    { const int rc_ = settle_open_copy(c); if (rc_ != LG_OK) return rc_; }
    const uint32_t plen = (uint32_t)c->logn - 1;
    if (c->scr.idx_cap < nidx) {
        if (c->scr.d_idx) LG_HIP(c, hipFree(c->scr.d_idx));
        c->scr.d_idx = nullptr; c->scr.idx_cap = 0;
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->scr.d_idx), nidx * sizeof(uint32_t)));
        c->scr.idx_cap = nidx;
    }
    const size_t path_bytes = nidx * (plen + 1) * 32;
    if (c->scr.path_cap < path_bytes) {
        if (c->scr.d_path) LG_HIP(c, hipFree(c->scr.d_path));
        c->scr.d_path = nullptr; c->scr.path_cap = 0;
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->scr.d_path), path_bytes));
        c->scr.path_cap = path_bytes;
    }
    int rc = grow(c, &c->scr.c, &c->scr.c_elems, nidx * c->rows);
    if (rc != LG_OK) return rc;
    LG_HIP(c, hipMemcpyAsync(c->scr.d_idx, idx, nidx * sizeof(uint32_t), hipMemcpyHostToDevice, c->st.main));
    uint8_t* d_sib = c->scr.d_path;
    uint8_t* d_paths = c->scr.d_path + nidx * 32;
    rc = gather_columns_launch(c, proof0, nproofs, c->scr.d_idx, t, c->scr.c, d_sib, d_paths);
    if (rc != LG_OK) return rc;
    // queued openings travel on the download stream: 50 MB per opening at 2^20 constraints would otherwise sit in front of the next
    // sub-proof's kernels on the encode stream (0.9 ms each)
    hipStream_t cs = c->st.main;
    if (!wait) {
        if (!c->scr.ev_gathered) LG_HIP(c, hipEventCreateWithFlags(&c->scr.ev_gathered, lg_event_flags()));
        if (!c->scr.ev_copied) LG_HIP(c, hipEventCreateWithFlags(&c->scr.ev_copied, hipEventDisableTiming));
        LG_HIP(c, hipEventRecord(c->scr.ev_gathered, c->st.main));
        LG_HIP(c, hipStreamWaitEvent(c->st.dn, c->scr.ev_gathered, 0));
        cs = c->st.dn;
    }
    LG_HIP(c, hipMemcpyAsync(cols_out, c->scr.c, nidx * c->rows * sizeof(fr), hipMemcpyDeviceToHost, cs));
    LG_HIP(c, hipMemcpyAsync(sib_out, d_sib, nidx * 32, hipMemcpyDeviceToHost, cs));
    if (plen) LG_HIP(c, hipMemcpyAsync(paths_out, d_paths, nidx * plen * 32, hipMemcpyDeviceToHost, cs));
    if (wait) {
        LG_HIP(c, hipStreamSynchronize(c->st.main));
    } else {
        LG_HIP(c, hipEventRecord(c->scr.ev_copied, c->st.dn));
        c->scr.copy_pending = true;
    }
    return LG_OK;
}

extern "C" {

int lg_open_columns(lg_ctx* c, uint32_t proof, const uint32_t* idx, uint32_t t, uint64_t* cols_out, uint8_t* sib_out, uint8_t* paths_out) {
    return open_columns_impl(c, proof, 1, idx, t, cols_out, sib_out, paths_out);
}

int lg_open_columns_async(lg_ctx* c, uint32_t proof, const uint32_t* idx, uint32_t t, uint64_t* cols_out, uint8_t* sib_out, uint8_t* paths_out) {
    return open_columns_impl(c, proof, 1, idx, t, cols_out, sib_out, paths_out, false);
}

int lg_open_columns_wait(lg_ctx* c) {
    if (!c) return LG_ERR_BAD_ARG;
    if (c->gf || !c->scr.ev_copied) return LG_OK;
    LG_HIP(c, hipSetDevice(c->device));
    LG_HIP(c, hipEventSynchronize(c->scr.ev_copied));      // (the latest one: the download stream runs the openings in order)
    c->scr.copy_pending = false;
    return LG_OK;
}

int lg_open_columns_batch(lg_ctx* c, const uint32_t* idx, uint32_t t, uint64_t* cols_out, uint8_t* sib_out, uint8_t* paths_out) {
    if (!c) return LG_ERR_BAD_ARG;
    return open_columns_impl(c, 0, c->batch, idx, t, cols_out, sib_out, paths_out);
}

}  // extern "C"

// row operators on scratch buffers: a = input rows / coefficients, b = coset planes, c = natural-order output
static int rs_common(lg_ctx* c, const uint64_t* in, uint32_t nrows, uint64_t* out, bool do_interp, bool do_eval) {
    if (!c || !in || !out) return LG_ERR_BAD_ARG;
    if (c->gf) {
        if ((uint64_t)nrows > c->total_rows) return LG_ERR_BAD_ARG;
        LG_HIP(c, hipSetDevice(c->device));
        return gf_reed_solomon(c->gf, in, nrows, out, do_interp, do_eval);
    }
    if (nrows == 0) return LG_OK;
    if ((uint64_t)nrows > c->total_rows) return LG_ERR_BAD_ARG;
    LG_HIP(c, hipSetDevice(c->device));
    const size_t mat = (size_t)nrows * c->k;
    int rc = grow(c, &c->scr.a, &c->scr.a_elems, 2 * mat);
    if (rc != LG_OK) return rc;
    fr* d_in = c->scr.a;
    fr* d_co = c->scr.a + mat;
    LG_HIP(c, hipMemcpyAsync(d_in, in, mat * sizeof(fr), hipMemcpyHostToDevice, c->st.main));
    const fr* coeffs = d_in;
    if (do_interp) {
        lg::NttArgs a = interp_args(c, d_in, d_co, nullptr, 0, nrows);
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->st.main, a));
        coeffs = d_co;
    }
    if (!do_eval) return read_back(c, out, coeffs, mat * sizeof(fr));
    rc = grow(c, &c->scr.b, &c->scr.b_elems, 8 * mat);
    if (rc != LG_OK) return rc;
    { const int rc_ = settle_open_copy(c); if (rc_ != LG_OK) return rc_; }
    rc = grow(c, &c->scr.c, &c->scr.c_elems, 8 * mat);
    if (rc != LG_OK) return rc;
    const uint64_t sstride = (uint64_t)nrows * c->ki;
    lg::NttArgs a = eval_args(c, coeffs, c->scr.b, sstride, 0, nrows, true);
    LG_HIP(c, lg::launch_ntt(c->logki, c->logo, true, c->st.main, a));
    const uint64_t threads = 8 * (uint64_t)mat;
    hipLaunchKernelGGL(lg::planes_to_rows_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->st.main, c->scr.b,
                       sstride, (uint64_t)0, nrows, c->ki, (uint32_t)c->lognp, c->tab.r2, c->scr.c);
    LG_HIP(c, hipGetLastError());
    return read_back(c, out, c->scr.c, 8 * mat * sizeof(fr));
}
extern "C" {

int lg_reed_solomon_interpolate(lg_ctx* c, const uint64_t* msg, uint32_t nrows, uint64_t* coeffs_out) {
    return rs_common(c, msg, nrows, coeffs_out, true, false);
}
int lg_reed_solomon_evaluate(lg_ctx* c, const uint64_t* coeffs, uint32_t nrows, uint64_t* codeword_out) {
    return rs_common(c, coeffs, nrows, codeword_out, false, true);
}
int lg_reed_solomon(lg_ctx* c, const uint64_t* msg, uint32_t nrows, uint64_t* codeword_out) {
    return rs_common(c, msg, nrows, codeword_out, true, true);
}

}  // extern "C"
