// One proof over several GPUs: the staged calls (lg_stage_*), the one-call commits with the collectives
// as callbacks on the library's streams (lg_commit_sharded: coset-sharded; lg_commit_row_relay: rows end to end), stage times.
#include "lg_context.h"

// ---- staged commit for one proof sharded over several GPUs ----------------
// Coset-sharded: rank g interpolates its row shard, the host layer all-gathers the coefficient rows (RCCL), then rank g
// evaluates and hashes the planes it owns for ALL rows, the host layer all-gathers the leaf digests, and every rank builds
// the (replicated) tree.  Row-relay: rank g keeps its rows end to end (all planes) and the columns' Blake2s states travel
// from rank to rank (lg_stage_hash_rows).
static uint32_t message_planes_mask(const lg_ctx* c) {   // planes s = 0 (mod 8): they hold the canonical message itself
    return all_planes_mask(c) & 0x01010101u;
}
static void add_canon_range(lg_ctx* c, uint32_t r0, uint32_t r1) {
    auto& v = c->held.canon_ranges;
    v.emplace_back(r0, r1);
    std::sort(v.begin(), v.end());
    size_t w = 0;
    for (size_t i = 1; i < v.size(); i++) {
        if (v[i].first <= v[w].second) v[w].second = std::max(v[w].second, v[i].second);
        else v[++w] = v[i];
    }
    v.resize(w + 1);
}

extern "C" {

int lg_stage_interpolate(lg_ctx* c, const uint64_t* preenc_rows, uint32_t row0, uint32_t nrows) {
    if (!c) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;   // generic-field contexts serve the hot path only
    if (c->batch != 1) return LG_ERR_STATE;
    if ((uint64_t)row0 + nrows > c->rows) return LG_ERR_BAD_ARG;
    if (nrows == 0) return LG_OK;
    LG_HIP(c, hipSetDevice(c->device));
    // the interpolation rewrites message planes of U: nothing of an earlier commitment may still be reading them
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    if (c->shard.on) {
        // the message rows of a sharded proof exist only shard by shard: hold exactly the range handed over
        const bool inside = c->shard.d_preenc_alloc && row0 >= c->shard.pre_row0 && (uint64_t)row0 + nrows <= (uint64_t)c->shard.pre_row0 + c->shard.pre_rows;
        if (!inside) {
            if (!preenc_rows) return LG_ERR_STATE;   // "already resident" rows that were never uploaded
            LG_HIP(c, hipStreamSynchronize(c->st.main));
            if (c->shard.d_preenc_alloc) LG_HIP(c, hipFree(c->shard.d_preenc_alloc));
            c->shard.d_preenc_alloc = nullptr; c->shard.pre_rows = 0; c->shard.alloc_rows = 0;
            c->shard.forget_layout();   // the rows lg_commit_sharded left here are gone: its preenc_rows = NULL path must not trust them
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->shard.d_preenc_alloc), (size_t)nrows * c->k * sizeof(fr)));
            c->shard.pre_row0 = row0; c->shard.pre_rows = nrows; c->shard.alloc_rows = nrows;
            c->d_preenc = c->shard.d_preenc_alloc - (size_t)row0 * c->k;   // virtual base: indexed by absolute row
        }
    }
    if (preenc_rows)
        LG_HIP(c, hipMemcpyAsync(c->d_preenc + (size_t)row0 * c->k, preenc_rows, (size_t)nrows * c->k * sizeof(fr), hipMemcpyDefault, c->st.main));   // (host or device rows: lg_tracer_rows)
    // The code is systematic: the message planes (s = 0 mod 8) of these rows ARE the message, so the interpolation writes their
    // canonical copy into the ones this context holds and the evaluation skips them for these rows (a rank of a coset-sharded
    // proof still has to evaluate them for the rows it only receives coefficients of)
    const uint32_t msg_held = message_planes_mask(c) & own_planes_mask(c);
    lg::NttArgs a = interp_args(c, c->d_preenc, c->d_coeffs, msg_held ? c->d_u : nullptr, row0, nrows);
    if (msg_held) {
        a.plane_stride = c->total_rows * c->ki;
        a.canon_mask = 0;
        for (uint32_t cc = 0; cc < (1u << c->logo); cc++)
            if (msg_held & (1u << (8 * cc))) a.canon_mask |= 1u << cc;
    }
    LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->st.main, a));
    if (!c->held.staging) c->held.canon_ranges.clear();
    if (msg_held) add_canon_range(c, row0, row0 + nrows);
    // a staged commit starts (or grows by an adjacent row range); what an earlier commitment left in U is void
    if (c->held.staging && row0 == c->held.row1) c->held.row1 = row0 + nrows;
    else if (c->held.staging && row0 + nrows == c->held.row0) c->held.row0 = row0;
    else if (!(c->held.staging && row0 >= c->held.row0 && row0 + nrows <= c->held.row1)) { c->held.row0 = row0; c->held.row1 = row0 + nrows; }
    c->held.staging = true;
    c->held.planes = 0;
    c->held.committed = false;
    return LG_OK;
}

}  // extern "C"

static int stage_plane_args(lg_ctx* c, uint32_t plane_mask) {
    if (c->gf) return LG_ERR_UNSUPPORTED;   // generic-field contexts serve the hot path only
    if (c->batch != 1) return LG_ERR_STATE;
    if (c->nplanes < 32 && (plane_mask >> c->nplanes) != 0) return LG_ERR_BAD_ARG;
    if (plane_mask & ~own_planes_mask(c)) {
        snprintf(c->err, sizeof(c->err), "plane mask 0x%x reaches outside the planes [%u, %u) this sharded context holds", plane_mask, c->shard.plane0,
                 c->shard.plane0 + c->shard.planes);
        return LG_ERR_BAD_ARG;
    }
    return LG_OK;
}

// evaluation of the planes of plane_mask for rows [r0, r1) from LG_BUF_COEFFS.  Rows this context interpolated itself during
// the staged commit in progress (canon_ranges) already have their message planes (lg_stage_interpolate).
static int stage_evaluate_launch(lg_ctx* c, uint32_t mask, uint32_t r0, uint32_t r1) {
    if (r1 <= r0) return LG_OK;
    lg::NttArgs a = eval_args(c, c->d_coeffs, c->d_u, c->total_rows * c->ki, r0, r1 - r0, true);
    a.ncos = 0;
    for (uint32_t s = 0; s < c->nplanes; s++)
        if (mask & (1u << s)) a.cosets[a.ncos++] = (uint8_t)s;
    if (a.ncos == 0) return LG_OK;
    a.chunk_rows = a.rows;
    LG_HIP(c, lg::launch_ntt(c->logki, c->logo, true, c->st.main, a));
    return LG_OK;
}
static int stage_evaluate_range(lg_ctx* c, uint32_t plane_mask, uint32_t r0, uint32_t r1) {
    if (r1 <= r0 || plane_mask == 0) return LG_OK;
    const uint32_t nomsg = plane_mask & ~message_planes_mask(c);
    uint32_t at = r0;
    if (c->held.staging && nomsg != plane_mask)
        for (const auto& cr : c->held.canon_ranges) {
            const uint32_t a0 = std::max(at, cr.first), a1 = std::min(r1, cr.second);
            if (a1 <= a0) continue;
            { const int rc_ = stage_evaluate_launch(c, plane_mask, at, a0); if (rc_ != LG_OK) return rc_; }
            { const int rc_ = stage_evaluate_launch(c, nomsg, a0, a1); if (rc_ != LG_OK) return rc_; }
            at = a1;
        }
    return stage_evaluate_launch(c, plane_mask, at, r1);
}

// column hashes of the planes of plane_mask over rows [row0, row0 + nrows) of this context's U, which are rows
// [col_pos, col_pos + nrows) of columns of col_rows rows; one launch per run of consecutive planes, on stream `hs`
static int stage_hash_launch(lg_ctx* c, hipStream_t hs, uint32_t plane_mask, uint32_t row0, uint32_t nrows, uint64_t col_pos, uint64_t col_rows) {
    const uint64_t plane = c->total_rows * c->ki;
    for (uint32_t s = 0; s < c->nplanes;) {
        if (!(plane_mask & (1u << s))) { s++; continue; }
        uint32_t e2 = s;
        while (e2 + 1 < c->nplanes && (plane_mask & (1u << (e2 + 1)))) e2++;
        lg::ColHashArgs h;
        memset(&h, 0, sizeof(h));
        h.u = reinterpret_cast<const uint4*>(c->d_u);
        h.leaves = c->d_leaves;
        h.state = c->d_hstate;
        h.rows = c->rows; h.k = c->ki; h.lognp = (uint32_t)c->lognp;
        h.proof_begin = 0; h.proof_count = 1;
        h.row_begin = row0; h.row_end = row0 + nrows;
        h.first = col_pos == 0;
        h.last = col_pos + nrows == col_rows;
        h.plane_begin = s; h.plane_count = e2 - s + 1;
        h.plane_stride = plane;
        h.col_pos = col_pos; h.col_rows = col_rows;
        // few columns (a rank's planes of a coset-sharded proof: n / G of them): the one-lane kernel would be a latency chain on a
        // fraction of the SIMDs; four lanes per column shorten it, also across row ranges (the parked state is the same)
        { const int rc_ = colhash_launch(c, hs, h, true); if (rc_ != LG_OK) return rc_; }
        s = e2 + 1;
    }
    return LG_OK;
}

extern "C" {

int lg_stage_evaluate_hash(lg_ctx* c, uint32_t plane_mask) {
    if (!c) return LG_ERR_BAD_ARG;
    { const int rc_ = stage_plane_args(c, plane_mask); if (rc_ != LG_OK) return rc_; }
    c->held.touch_staged();   // re-evaluating over a finished commitment voids it
    if (plane_mask == 0) return LG_OK;
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    // Row chunks, as in commit_core: while the encode stream evaluates chunk i + 1 of the owned planes, the hash stream absorbs
    // chunk i into the column states (S22 on one rank: 104 -> 92 ms; on 8 ranks each rank's share of it)
    Chunk chunks[lg_ctx::kMaxChunks];
    const int nchunks = plan_chunks(c, chunks);
    hipStream_t hs = nchunks > 1 ? c->st.hash : c->st.main;
    if (nchunks > 1) {
        LG_HIP(c, hipEventRecord(c->evt.done, c->st.main));     // earlier work on the encode stream (the previous tree) may read the leaves
        LG_HIP(c, hipStreamWaitEvent(hs, c->evt.done, 0));
    }
    for (int i = 0; i < nchunks; i++) {
        const Chunk& ch = chunks[i];
        { const int rc_ = stage_evaluate_range(c, plane_mask, ch.row_begin, ch.row_end); if (rc_ != LG_OK) return rc_; }
        if (nchunks > 1) {
            LG_HIP(c, hipEventRecord(c->evt.chunk[i], c->st.main));
            LG_HIP(c, hipStreamWaitEvent(hs, c->evt.chunk[i], 0));
        }
        { const int rc_ = stage_hash_launch(c, hs, plane_mask, ch.row_begin, ch.row_end - ch.row_begin, ch.row_begin, c->rows); if (rc_ != LG_OK) return rc_; }
    }
    if (nchunks > 1) {   // later work on the encode stream (lg_stage_merkle, the caller's all-gather after lg_sync) sees the leaves
        LG_HIP(c, hipEventRecord(c->evt.done, hs));
        LG_HIP(c, hipStreamWaitEvent(c->st.main, c->evt.done, 0));
    }
    c->held.planes |= plane_mask;
    return LG_OK;
}

// Split form of lg_stage_evaluate_hash for a caller that receives the coefficient rows piece by piece (an all-gather cut into pieces
// that arrive while earlier pieces are being evaluated): evaluate ANY rows that are there, in any order, then hash once all are done.
int lg_stage_evaluate_rows(lg_ctx* c, uint32_t plane_mask, uint32_t row0, uint32_t nrows) {
    if (!c) return LG_ERR_BAD_ARG;
    { const int rc_ = stage_plane_args(c, plane_mask); if (rc_ != LG_OK) return rc_; }
    if ((uint64_t)row0 + nrows > c->rows) return LG_ERR_BAD_ARG;
    c->held.touch_staged();   // re-evaluating over a finished commitment voids it
    if (nrows == 0 || plane_mask == 0) return LG_OK;
    LG_HIP(c, hipSetDevice(c->device));
    return stage_evaluate_range(c, plane_mask, row0, row0 + nrows);
}

// Column hashes of a ROW RANGE: rows [row0, row0 + nrows) of this context are rows [col_pos, col_pos + nrows) of columns
// that are col_rows rows long (mod.rs:536-542: the length prefix is col_rows).  col_pos = 0 starts the columns, otherwise
// their Blake2s states are resumed from LG_BUF_HSTATE; col_pos + nrows = col_rows finalises them into LG_BUF_LEAVES,
// otherwise the states go back to LG_BUF_HSTATE.  The launch is queued on the hash stream behind everything issued so far.
int lg_stage_hash_rows(lg_ctx* c, uint32_t plane_mask, uint32_t row0, uint32_t nrows, uint64_t col_pos, uint64_t col_rows) {
    if (!c) return LG_ERR_BAD_ARG;
    { const int rc_ = stage_plane_args(c, plane_mask); if (rc_ != LG_OK) return rc_; }
    // (the column's byte length 8 + 32 col_rows is a 64-bit Blake2s counter)
    if ((uint64_t)row0 + nrows > c->rows || nrows == 0 || col_rows > (1ull << 58) || col_pos > col_rows || nrows > col_rows - col_pos) return LG_ERR_BAD_ARG;
    c->held.touch_staged();
    if (plane_mask == 0) return LG_OK;
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }   // the previous tree may still read the leaves
    LG_HIP(c, hipEventRecord(c->evt.stage_in, c->st.main));               // the rows just evaluated, a state just received
    LG_HIP(c, hipStreamWaitEvent(c->st.hash, c->evt.stage_in, 0));
    { const int rc_ = stage_hash_launch(c, c->st.hash, plane_mask, row0, nrows, col_pos, col_rows); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipEventRecord(c->evt.stage_hash, c->st.hash));
    c->held.hash_pending = true;
    c->held.planes |= plane_mask;
    return LG_OK;
}

int lg_stage_hash(lg_ctx* c, uint32_t plane_mask) {
    if (!c) return LG_ERR_BAD_ARG;
    return lg_stage_hash_rows(c, plane_mask, 0, c->rows, 0, c->rows);
}

int lg_stage_merkle(lg_ctx* c) {
    if (!c) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;   // generic-field contexts serve the hot path only
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    lg::MerkleArgs m;
    m.leaves = c->d_leaves; m.nodes = c->d_nodes; m.n = c->n; m.logn = (uint32_t)c->logn; m.batch = c->batch;
    uint32_t depth = (uint32_t)c->logn;
    bool leaf = true;
    while (depth > 0) {
        m.in_depth = depth;
        m.chunks = depth > 9 ? (1u << (depth - 9)) : 1u;
        const dim3 grid(c->batch * m.chunks);
        if (leaf)
            LG_LAUNCH(c, lg::merkle_subtree_kernel<true>, grid, dim3(256), 0, c->st.main, m);
        else
            LG_LAUNCH(c, lg::merkle_subtree_kernel<false>, grid, dim3(256), 0, c->st.main, m);
        leaf = false;
        depth = depth > 9 ? depth - 9 : 0;
    }
    LG_HIP(c, hipGetLastError());
    c->held.committed = true;
    c->held.staging = false;
    return LG_OK;
}

}  // extern "C"

// Digest exchange of the sharded commit without a host-side layout pass: pack copies the leaf digests of this rank's planes into
// block `rank` of a staging buffer of `world` equal blocks ([q][plane of the run][32] each), the host layer all-gathers the
// buffer in place, unpack scatters every block back into leaf order j = np q + s.
static int digest_run(lg_ctx* c, uint32_t world, uint32_t* per_out) {
    if (world == 0 || c->nplanes % world != 0) {
        snprintf(c->err, sizeof(c->err), "%u coset planes cannot be dealt to %u ranks in equal runs", c->nplanes, world);
        return LG_ERR_BAD_ARG;
    }
    *per_out = c->nplanes / world;
    return LG_OK;
}
extern "C" {

int lg_stage_digests_pack(lg_ctx* c, uint32_t world, uint32_t rank, void** dptr_out, size_t* bytes_per_rank_out) {
    if (!c || !dptr_out || !bytes_per_rank_out || rank >= world) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    if (c->batch != 1) return LG_ERR_STATE;
    uint32_t per;
    { const int rc_ = digest_run(c, world, &per); if (rc_ != LG_OK) return rc_; }
    uint32_t run = 0;
    for (uint32_t s = rank * per; s < (rank + 1) * per; s++) run |= 1u << s;
    { const int rc_ = need_planes(c, run, "lg_stage_digests_pack"); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    const size_t block = (size_t)c->ki * per * 32;
    if (!c->shard.d_digest_xchg) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->shard.d_digest_xchg), (size_t)c->n * 32));
    LG_HIP(c, hipMemcpy2DAsync(c->shard.d_digest_xchg + (size_t)rank * block, (size_t)per * 32, c->d_leaves + (size_t)rank * per * 32, (size_t)c->nplanes * 32,
                               (size_t)per * 32, c->ki, hipMemcpyDeviceToDevice, c->st.main));
    *dptr_out = c->shard.d_digest_xchg;
    *bytes_per_rank_out = block;
    return LG_OK;
}
int lg_stage_digests_unpack(lg_ctx* c, uint32_t world) {
    if (!c) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    if (c->batch != 1 || !c->shard.d_digest_xchg) return LG_ERR_STATE;
    uint32_t per;
    { const int rc_ = digest_run(c, world, &per); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    const size_t block = (size_t)c->ki * per * 32;
    for (uint32_t o = 0; o < world; o++)
        LG_HIP(c, hipMemcpy2DAsync(c->d_leaves + (size_t)o * per * 32, (size_t)c->nplanes * 32, c->shard.d_digest_xchg + (size_t)o * block, (size_t)per * 32,
                                   (size_t)per * 32, c->ki, hipMemcpyDeviceToDevice, c->st.main));
    return LG_OK;
}

int lg_device_buffer(lg_ctx* c, int which, void** dptr_out, size_t* bytes_out) {
    if (!c || !dptr_out || !bytes_out) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;   // generic-field contexts serve the hot path only
    if (which == LG_BUF_LEAVES || which == LG_BUF_NODES || which == LG_BUF_HSTATE) {   // the caller will touch them outside our streams' order
        LG_HIP(c, hipSetDevice(c->device));
        { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
        { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    }
    switch (which) {
        case LG_BUF_PREENC:   // sharded: the allocated row range [pre_row0, pre_row0 + pre_rows) only
            // (a getter changes no state: a zero-copy producer that filled the matrix says so with lg_preenc_mark_filled)
            *dptr_out = c->shard.on ? c->shard.d_preenc_alloc : c->d_preenc;
            *bytes_out = (size_t)(c->shard.on ? c->shard.pre_rows : c->total_rows) * c->k * sizeof(fr);
            break;
        case LG_BUF_COEFFS: *dptr_out = c->d_coeffs; *bytes_out = (size_t)c->shard.coeff_rows_alloc * c->k * sizeof(fr); break;
        case LG_BUF_LEAVES: *dptr_out = c->d_leaves; *bytes_out = (size_t)c->batch * c->n * 32; break;
        case LG_BUF_NODES: *dptr_out = c->d_nodes; *bytes_out = (size_t)c->batch * (c->n - 1) * 32; break;
        case LG_BUF_HSTATE: *dptr_out = c->d_hstate; *bytes_out = (size_t)c->batch * c->n * LG_HSTATE_BYTES; break;
        default: return LG_ERR_BAD_ARG;
    }
    return LG_OK;
}

}  // extern "C"

// ---- one call per commit for a proof sharded over several GPUs: the stages above in one stream-ordered sequence, the exchanges
// through the caller's callbacks (include/ligero_hip.h: lg_comm).  Nothing here waits on the host.
static int comm_fail(lg_ctx* c, const char* what, int rc) {
    snprintf(c->err, sizeof(c->err), "%s: the communication callback returned %d", what, rc);
    return LG_ERR_COMM;
}
static int shard_events(lg_ctx* c, hipEvent_t** ev_out) {
    *ev_out = nullptr;
    if (!c->prof.on) return LG_OK;
    if (!c->shard.ev_valid) {
        for (auto& set : c->shard.ev)
            for (auto& e : set) LG_HIP(c, hipEventCreate(&e));
        c->shard.ev_valid = true;
    }
    *ev_out = c->shard.ev[c->shard.commits % lg_ctx::kShardProfRing];
    return LG_OK;
}
// ownership rule of lg_commit_sharded: the rows are cut into `pieces` pieces of world * sub rows, rank g owns sub-block g of
// every piece -- so that piece c of the all-gather is ONE in-place collective on rows [c world sub, (c + 1) world sub) of
// LG_BUF_COEFFS and complete row prefixes arrive in order (the column hash can follow the evaluation piece by piece)
static void shard_plan(uint32_t rows, uint32_t world, uint32_t pieces, uint32_t* sub_out, uint32_t* pieces_out) {
    if (pieces < 1) pieces = 1;
    if (pieces > (uint32_t)lg_ctx::kMaxChunks) pieces = lg_ctx::kMaxChunks;
    uint32_t per_piece = (rows + pieces - 1) / pieces;                 // rows per piece before rounding up to whole sub-blocks
    uint32_t sub = (per_piece + world - 1) / world;
    if (sub == 0) sub = 1;
    // pieces start on even rows: a piece is hashed on its own and two rows share a Blake2s block (the four-lanes-per-column kernel
    // resumes at block boundaries only)
    if (pieces > 1 && (sub & 1)) sub++;
    *sub_out = sub;
    *pieces_out = (rows + world * sub - 1) / (world * sub);            // pieces that hold at least one row
}

extern "C" {

int lg_shard_row_ranges(uint32_t rows, uint32_t world, uint32_t rank, uint32_t pieces, uint32_t* ranges_out, uint32_t* nranges_out) {
    if (!ranges_out || !nranges_out || world == 0 || rank >= world || rows == 0) return LG_ERR_BAD_ARG;
    uint32_t sub, np;
    shard_plan(rows, world, pieces, &sub, &np);
    uint32_t n = 0;
    for (uint32_t p = 0; p < np; p++) {
        const uint64_t a = (uint64_t)p * world * sub + (uint64_t)rank * sub;
        const uint64_t b = std::min<uint64_t>(rows, a + sub);
        if (b > a) { ranges_out[2 * n] = (uint32_t)a; ranges_out[2 * n + 1] = (uint32_t)(b - a); n++; }
    }
    *nranges_out = n;
    return LG_OK;
}

// A commit that fails after it began queueing work (a callback's error, a HIP error) leaves NO commitment and no half-staged
// state behind: the flags a later lg_stage_* call would extend are cleared and the encode stream is ordered behind whatever
// the exchange and hash streams still hold.  (Peers already inside a collective are the communicator's to unblock.)
static int abandon_staged(lg_ctx* c, int rc) {
    c->held.drop();
    c->held.row0 = c->held.row1 = 0;
    c->held.hash_pending = false;
    for (hipStream_t s : {c->st.xchg, c->st.hash})
        if (s && hipEventRecord(c->evt.done, s) == hipSuccess) (void)hipStreamWaitEvent(c->st.main, c->evt.done, 0);
    return rc;
}

static int commit_sharded_body(lg_ctx* c, const lg_comm* comm, const uint64_t* preenc_rows, uint32_t pieces, bool* began) {
    if (!c || !comm) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    if (c->batch != 1) return LG_ERR_STATE;
    const uint32_t world = comm->world, rank = comm->rank;
    if (world == 0 || rank >= world) return LG_ERR_BAD_ARG;
    const bool exchange = world > 1 || (comm->flags & LG_COMM_EXCHANGE_AT_WORLD_1);
    if (exchange && !comm->all_gather) return LG_ERR_BAD_ARG;
    if (c->nplanes % world != 0 || c->shard.planes != c->nplanes / world || c->shard.plane0 != rank * (c->nplanes / world)) {
        snprintf(c->err, sizeof(c->err), "lg_commit_sharded: rank %u of %u must hold planes [%u, %u) of %u (lg_ctx_create_sharded); this context holds [%u, %u)", rank, world,
                 rank * (c->nplanes / world), (rank + 1) * (c->nplanes / world), c->nplanes, c->shard.plane0, c->shard.plane0 + c->shard.planes);
        return LG_ERR_STATE;
    }
    LG_HIP(c, hipSetDevice(c->device));
    *began = true;
    uint32_t sub, np;
    shard_plan(c->rows, world, pieces, &sub, &np);
    const uint32_t piece_rows = world * sub;
    uint32_t ranges[2 * lg_ctx::kMaxChunks], nranges = 0;
    lg_shard_row_ranges(c->rows, world, rank, pieces, ranges, &nranges);
    uint32_t own = 0;
    for (uint32_t i = 0; i < nranges; i++) own += ranges[2 * i + 1];
    // earlier work that reads what is about to be rewritten
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    // the coefficient buffer holds np whole pieces (padding rows of the last one are exchanged but never read)
    if ((uint64_t)np * piece_rows > c->shard.coeff_rows_alloc) {
        LG_HIP(c, hipStreamSynchronize(c->st.main));
        if (c->st.xchg) LG_HIP(c, hipStreamSynchronize(c->st.xchg));
        LG_HIP(c, hipFree(c->d_coeffs));
        c->d_coeffs = nullptr; c->shard.coeff_rows_alloc = 0;
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_coeffs), (size_t)np * piece_rows * c->k * sizeof(fr)));
        c->shard.coeff_rows_alloc = np * piece_rows;
    }
    // this rank's message rows, compact and piece-major
    // (the note is void when the allocation changed hands since -- lg_stage_interpolate -- and never trusted beyond the rows allocated)
    const bool same_layout = c->shard.pieces == np && c->shard.world == world && c->shard.rank == rank && c->shard.compact_rows == own &&
                             (!c->shard.on || c->shard.alloc_rows == own);
    if (!preenc_rows && own && !(same_layout && (c->shard.on ? c->shard.d_preenc_alloc != nullptr : true))) {
        snprintf(c->err, sizeof(c->err), "lg_commit_sharded: no resident rows of this layout (pass this rank's %u rows)", own);
        return LG_ERR_STATE;
    }
    fr* compact = nullptr;
    if (c->shard.on) {
        if (own && (!c->shard.d_preenc_alloc || !same_layout)) {
            LG_HIP(c, hipStreamSynchronize(c->st.main));
            if (c->shard.d_preenc_alloc) LG_HIP(c, hipFree(c->shard.d_preenc_alloc));
            c->shard.d_preenc_alloc = nullptr; c->shard.pre_rows = 0; c->shard.alloc_rows = 0;
            c->shard.forget_layout();
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->shard.d_preenc_alloc), (size_t)own * c->k * sizeof(fr)));
            c->shard.alloc_rows = own;
        }
        compact = c->shard.d_preenc_alloc;
        // (ranges that follow one another -- a single range, or one rank owning every sub-block -- keep lg_stage_interpolate's "rows
        // [pre_row0, pre_row0 + pre_rows) are resident" view of the same buffer: the compact order is then the matrix order)
        bool one_run = nranges >= 1;
        for (uint32_t i = 1; i < nranges; i++) one_run = one_run && ranges[2 * i] == ranges[2 * (i - 1)] + ranges[2 * (i - 1) + 1];
        c->shard.pre_row0 = one_run ? ranges[0] : 0;
        c->shard.pre_rows = one_run ? own : 0;
        c->d_preenc = one_run ? c->shard.d_preenc_alloc - (size_t)ranges[0] * c->k : nullptr;
    } else {
        // an unsharded context (world 1): the matrix has its own full-size buffer; the rows sit at their own positions
        compact = c->d_preenc;
    }
    c->shard.pieces = np; c->shard.world = world; c->shard.rank = rank; c->shard.compact_rows = own;
    if (preenc_rows && own) LG_HIP(c, hipMemcpyAsync(compact, preenc_rows, (size_t)own * c->k * sizeof(fr), hipMemcpyDefault, c->st.main));   // (host or device rows)
    hipEvent_t* ev = nullptr;
    { const int rc_ = shard_events(c, &ev); if (rc_ != LG_OK) return rc_; }
    if (ev) LG_HIP(c, hipEventRecord(ev[0], c->st.main));
    if (exchange && np > 1 && !c->st.xchg) LG_HIP(c, hipStreamCreateWithFlags(&c->st.xchg, hipStreamNonBlocking));
    // a new staged commit: what an earlier commitment left in U is void
    c->held.begin_staged();
    // the message rows this context holds as ONE run of the matrix: all of its ranges if they follow one another (one rank: the
    // whole matrix), else none -- the entry points that read preenc_u rows (lg_interleaved_row_mul, lg_commit_resident) index them
    // by matrix row, which a compact buffer of scattered ranges does not support
    bool rows_one_run = nranges >= 1;
    for (uint32_t i = 1; i < nranges; i++) rows_one_run = rows_one_run && ranges[2 * i] == ranges[2 * (i - 1)] + ranges[2 * (i - 1) + 1];
    c->held.row0 = rows_one_run ? ranges[0] : 0; c->held.row1 = rows_one_run ? ranges[0] + own : 0;
    const uint32_t msg_held = message_planes_mask(c) & own_planes_mask(c);
    const uint64_t plane = c->total_rows * c->ki;
    const uint32_t mask = own_planes_mask(c);
    // 1. interpolate this rank's rows piece by piece; piece p of the all-gather follows on the exchange stream
    uint32_t compact_off = 0;
    for (uint32_t i = 0; i < nranges; i++) {
        const uint32_t r0 = ranges[2 * i], nr = ranges[2 * i + 1], p = r0 / piece_rows;
        // `in` is indexed by the absolute row like `out`: bias the compact buffer's base accordingly
        const fr* in = c->shard.on ? compact + ((int64_t)compact_off - (int64_t)r0) * (int64_t)c->k : compact;
        lg::NttArgs a = interp_args(c, in, c->d_coeffs, msg_held ? c->d_u : nullptr, r0, nr);
        if (msg_held) {
            a.plane_stride = plane;
            a.canon_mask = 0;
            for (uint32_t cc = 0; cc < (1u << c->logo); cc++)
                if (msg_held & (1u << (8 * cc))) a.canon_mask |= 1u << cc;
            add_canon_range(c, r0, r0 + nr);
        }
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->st.main, a));
        compact_off += nr;
        (void)p;
    }
    if (ev) LG_HIP(c, hipEventRecord(ev[1], c->st.main));
    // 2. + 3. the pieces: all-gather (in place on whole pieces of LG_BUF_COEFFS), evaluate, hash -- piece p + 1 is on the wire while
    // piece p is evaluated, the hash of piece p runs on the hash stream beside the evaluation of piece p + 1
    hipStream_t xs = (exchange && np > 1) ? c->st.xchg : c->st.main;
    if (exchange) {
        if (xs != c->st.main) {
            LG_HIP(c, hipEventRecord(c->evt.done, c->st.main));            // every own row is interpolated
            LG_HIP(c, hipStreamWaitEvent(xs, c->evt.done, 0));
        }
        for (uint32_t p = 0; p < np; p++) {
            uint8_t* base = reinterpret_cast<uint8_t*>(c->d_coeffs) + (size_t)p * piece_rows * c->k * sizeof(fr);
            const int rc_ = comm->all_gather(comm->user, base, (uint64_t)sub * c->k * sizeof(fr), static_cast<void*>(xs));
            if (rc_ != 0) return comm_fail(c, "all-gather of the coefficient rows", rc_);
            if (xs != c->st.main) LG_HIP(c, hipEventRecord(c->evt.up[p], xs));
        }
    }
    if (ev) LG_HIP(c, hipEventRecord(ev[2], c->st.main));
    for (uint32_t p = 0; p < np; p++) {
        // (the time this stream stands still waiting for piece p is measured by an event on either side of the wait)
        if (ev) LG_HIP(c, hipEventRecord(ev[lg_ctx::kShardStages + 1 + 2 * p], c->st.main));
        if (exchange && xs != c->st.main) LG_HIP(c, hipStreamWaitEvent(c->st.main, c->evt.up[p], 0));
        if (ev) LG_HIP(c, hipEventRecord(ev[lg_ctx::kShardStages + 2 + 2 * p], c->st.main));
        const uint32_t r0 = p * piece_rows, r1 = std::min(c->rows, (p + 1) * piece_rows);
        // (a single piece of a large commit is still cut into row chunks, as lg_stage_evaluate_hash does)
        Chunk chunks[lg_ctx::kMaxChunks];
        int nchunks = np > 1 ? 1 : plan_chunks(c, chunks);
        if (np > 1) chunks[0] = Chunk{0, 1, r0, r1};
        for (int i = 0; i < nchunks; i++) {
            { const int rc_ = stage_evaluate_range(c, mask, chunks[i].row_begin, chunks[i].row_end); if (rc_ != LG_OK) return rc_; }
            LG_HIP(c, hipEventRecord(c->evt.stage_in, c->st.main));
            LG_HIP(c, hipStreamWaitEvent(c->st.hash, c->evt.stage_in, 0));
            { const int rc_ = stage_hash_launch(c, c->st.hash, mask, chunks[i].row_begin, chunks[i].row_end - chunks[i].row_begin, chunks[i].row_begin, c->rows); if (rc_ != LG_OK) return rc_; }
            LG_HIP(c, hipEventRecord(c->evt.stage_hash, c->st.hash));
            c->held.hash_pending = true;
        }
    }
    c->held.planes |= mask;
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    if (ev) LG_HIP(c, hipEventRecord(ev[3], c->st.main));
    // 4. the digests
    if (exchange) {
        void* d = nullptr; size_t bytes = 0;
        { const int rc_ = lg_stage_digests_pack(c, world, rank, &d, &bytes); if (rc_ != LG_OK) return rc_; }
        const int rc_ = comm->all_gather(comm->user, d, (uint64_t)bytes, static_cast<void*>(c->st.main));
        if (rc_ != 0) return comm_fail(c, "all-gather of the leaf digests", rc_);
        { const int rc2 = lg_stage_digests_unpack(c, world); if (rc2 != LG_OK) return rc2; }
    }
    if (ev) LG_HIP(c, hipEventRecord(ev[4], c->st.main));
    // 5. the tree
    { const int rc_ = lg_stage_merkle(c); if (rc_ != LG_OK) return rc_; }
    if (ev) { LG_HIP(c, hipEventRecord(ev[5], c->st.main)); c->shard.wait_pairs[c->shard.commits % lg_ctx::kShardProfRing] = np; c->shard.commits++; }
    return LG_OK;
}

int lg_commit_sharded(lg_ctx* c, const lg_comm* comm, const uint64_t* preenc_rows, uint32_t pieces) {
    bool began = false;
    const int rc = commit_sharded_body(c, comm, preenc_rows, pieces, &began);
    return (rc != LG_OK && began) ? abandon_staged(c, rc) : rc;
}

// Row-relay commit (lg_stage_hash_rows): this context holds the rank's OWN rows -- the ranges of lg_relay_row_ranges, concatenated
// in column order -- and all coset planes of them.
int lg_relay_row_ranges(uint64_t col_rows, uint32_t world, uint32_t rank, int layout, uint64_t* ranges_out, uint32_t* nranges_out) {
    if (!ranges_out || !nranges_out || world == 0 || rank >= world || col_rows == 0) return LG_ERR_BAD_ARG;
    uint32_t n = 0;
    if (layout == LG_RELAY_CONTIGUOUS) {
        // balanced, every boundary on an even row (two rows share a Blake2s block: the four-lanes-per-column kernel hands a column
        // over at block boundaries only); the last rank takes the odd row
        const uint64_t half = col_rows / 2;
        const uint64_t a = 2 * (half * rank / world), b = rank + 1 == world ? col_rows : 2 * (half * (rank + 1) / world);
        if (b > a) { ranges_out[0] = a; ranges_out[1] = b - a; n = 1; }
    } else if (layout == LG_RELAY_BLOCKS) {
        if (col_rows % 4) return LG_ERR_BAD_ARG;
        const uint64_t m = col_rows / 4, a = m * rank / world, b = m * (rank + 1) / world;
        if (b > a)
            for (uint32_t blk = 0; blk < 4; blk++) { ranges_out[2 * n] = blk * m + a; ranges_out[2 * n + 1] = b - a; n++; }
    } else if (layout > LG_RELAY_ROUND_ROBIN_BASE && layout <= LG_RELAY_ROUND_ROBIN_BASE + 8) {
        // the rows are cut into C * world balanced ranges (even boundaries, the last range takes the odd row) dealt to the ranks in
        // turn: rank g keeps ranges g, g + world, g + 2 world, ... -- it can encode its next range while the column states of the
        // current one are somewhere else on the ring
        const uint64_t C = (uint64_t)(layout - LG_RELAY_ROUND_ROBIN_BASE), R = C * world, half = col_rows / 2;
        for (uint64_t cidx = 0; cidx < C; cidx++) {
            const uint64_t i = cidx * world + rank;
            const uint64_t a = 2 * (half * i / R), b = i + 1 == R ? col_rows : 2 * (half * (i + 1) / R);
            if (b > a) { ranges_out[2 * n] = a; ranges_out[2 * n + 1] = b - a; n++; }
        }
    } else {
        return LG_ERR_BAD_ARG;
    }
    *nranges_out = n;
    return LG_OK;
}

static int commit_row_relay_body(lg_ctx* c, const lg_comm* comm, uint64_t col_rows, int layout, uint32_t plane_groups, const uint64_t* preenc_rows, bool* began) {
    if (!c || !comm) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    if (c->batch != 1 || c->shard.on) return LG_ERR_STATE;
    const uint32_t world = comm->world, rank = comm->rank;
    if (world == 0 || rank >= world) return LG_ERR_BAD_ARG;
    const bool exchange = world > 1;
    if (exchange && (!comm->send || !comm->recv || !comm->broadcast)) return LG_ERR_BAD_ARG;
    // the chain: every range of every rank, in column order
    struct Link { uint64_t pos, n; uint32_t owner, local; };
    std::vector<Link> chain;
    uint32_t local_rows = 0;
    for (uint32_t r = 0; r < world; r++) {
        uint64_t rg[16]; uint32_t nr = 0;
        { const int rc_ = lg_relay_row_ranges(col_rows, world, r, layout, rg, &nr); if (rc_ != LG_OK) return rc_; }
        uint32_t local = 0;
        for (uint32_t i = 0; i < nr; i++) {
            chain.push_back(Link{rg[2 * i], rg[2 * i + 1], r, local});
            local += (uint32_t)rg[2 * i + 1];
        }
        if (r == rank) local_rows = local;
    }
    std::sort(chain.begin(), chain.end(), [](const Link& a, const Link& b) { return a.pos < b.pos; });
    if (c->rows != std::max<uint32_t>(1, local_rows)) {
        snprintf(c->err, sizeof(c->err), "lg_commit_row_relay: rank %u of %u keeps %u of the %llu rows; this context has %u", rank, world, local_rows,
                 (unsigned long long)col_rows, c->rows);
        return LG_ERR_STATE;
    }
    LG_HIP(c, hipSetDevice(c->device));
    *began = true;
    hipEvent_t* ev = nullptr;
    { const int rc_ = shard_events(c, &ev); if (rc_ != LG_OK) return rc_; }
    if (ev) LG_HIP(c, hipEventRecord(ev[0], c->st.main));
    const uint32_t all = all_planes_mask(c);
    bool head_done = false;
    const bool round_robin = layout > LG_RELAY_ROUND_ROBIN_BASE;
    if (round_robin) {
        // Round robin: this rank keeps several ranges that are far apart in the column.  Encode stream: interpolate everything,
        // then evaluate range after range, an event behind each.  Hash stream: for every range in column order -- wait for ITS
        // evaluation, receive the column states from the rank before, absorb the rows, hand the states on.  The evaluation of the
        // next range runs while the states of this one are still on their way round the ring: the relay's serial chain (G hops of
        // one rank's hash each) and the encoding overlap instead of adding up.  Plane groups as in the contiguous layout (P runs of
        // planes per hop: G C + P - 1 steps of one group each, small enough for the four-lanes-per-column kernel), but the ring
        // wraps round -- rank G - 1 hands range c to rank 0's range c + 1 -- and its transfers are rendezvous: with every rank
        // posting its operations in order, P groups in flight stay free of a cycle only while P <= G - 1.
        { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
        { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
        if (local_rows) { const int rc_ = lg_stage_interpolate(c, preenc_rows, 0, local_rows); if (rc_ != LG_OK) return rc_; }
        else c->held.touch_staged();
        uint32_t ci = 0;
        for (const Link& l : chain) {
            if (l.owner != rank) continue;
            { const int rc_ = lg_stage_evaluate_rows(c, all, l.local, (uint32_t)l.n); if (rc_ != LG_OK) return rc_; }
            LG_HIP(c, hipEventRecord(c->evt.chunk[ci++ % lg_ctx::kMaxChunks], c->st.main));
        }
        if (ev) LG_HIP(c, hipEventRecord(ev[1], c->st.main));
        if (ev) LG_HIP(c, hipEventRecord(ev[2], c->st.main));
        uint32_t P = plane_groups ? plane_groups : (world <= 2 ? 1 : (world <= 4 ? 2 : 4));
        while (P & (P - 1)) P &= P - 1;                                       // a power of two (the planes are)
        while (P > 1 && (P > c->nplanes || P > world - 1 || (!plane_groups && (c->n / P) < 8192))) P >>= 1;
        if (P < 1) P = 1;
        const uint32_t per = c->nplanes / P;
        const size_t group_bytes = (size_t)per * c->ki * LG_HSTATE_BYTES;
        ci = 0;
        for (size_t i = 0; i < chain.size(); i++) {
            const Link& l = chain[i];
            if (l.owner != rank) continue;
            LG_HIP(c, hipStreamWaitEvent(c->st.hash, c->evt.chunk[ci++ % lg_ctx::kMaxChunks], 0));
            for (uint32_t g = 0; g < P; g++) {
                uint8_t* gstate = reinterpret_cast<uint8_t*>(c->d_hstate) + (size_t)g * group_bytes;
                const uint32_t gmask = (per >= 32 ? 0xffffffffu : ((1u << per) - 1u)) << (g * per);
                if (exchange && i > 0 && chain[i - 1].owner != rank) {
                    const int rc_ = comm->recv(comm->user, gstate, group_bytes, chain[i - 1].owner, static_cast<void*>(c->st.hash));
                    if (rc_ != 0) return comm_fail(c, "receive of the column states", rc_);
                }
                { const int rc_ = stage_hash_launch(c, c->st.hash, gmask, l.local, (uint32_t)l.n, l.pos, col_rows); if (rc_ != LG_OK) return rc_; }
                if (exchange && i + 1 < chain.size() && chain[i + 1].owner != rank) {
                    const int rc_ = comm->send(comm->user, gstate, group_bytes, chain[i + 1].owner, static_cast<void*>(c->st.hash));
                    if (rc_ != 0) return comm_fail(c, "send of the column states", rc_);
                }
            }
        }
        LG_HIP(c, hipEventRecord(c->evt.stage_hash, c->st.hash));
        c->held.hash_pending = true;
        c->held.planes |= all;
    } else if (local_rows) {
        { const int rc_ = lg_stage_interpolate(c, preenc_rows, 0, local_rows); if (rc_ != LG_OK) return rc_; }
        // evaluate in row chunks; the rank that holds the first rows of the columns hashes each chunk as soon as it is evaluated
        // (on the hash stream, beside the evaluation of the next chunk)
        const Link* first = nullptr;
        for (const Link& l : chain)
            if (l.owner == rank) { first = &l; break; }
        Chunk chunks[lg_ctx::kMaxChunks];
        const int planned = plan_chunks(c, chunks);
        const uint32_t n0 = (uint32_t)first->n, nch = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)planned, n0));
        const bool head = first->pos == 0;
        for (uint32_t i = 0; i < nch; i++) {
            // (even cuts, as plan_chunks makes them)
            const uint32_t a = i == 0 ? 0 : 2 * (uint32_t)((uint64_t)(n0 / 2) * i / nch), b = i + 1 == nch ? n0 : 2 * (uint32_t)((uint64_t)(n0 / 2) * (i + 1) / nch);
            if (b <= a) continue;
            { const int rc_ = lg_stage_evaluate_rows(c, all, first->local + a, b - a); if (rc_ != LG_OK) return rc_; }
            if (head) { const int rc_ = lg_stage_hash_rows(c, all, first->local + a, b - a, first->pos + a, col_rows); if (rc_ != LG_OK) return rc_; }
        }
        head_done = head;
        const uint32_t rest0 = first->local + n0;
        if (local_rows > rest0) { const int rc_ = lg_stage_evaluate_rows(c, all, rest0, local_rows - rest0); if (rc_ != LG_OK) return rc_; }
    } else {
        c->held.touch_staged();
    }
    if (ev && !round_robin) LG_HIP(c, hipEventRecord(ev[1], c->st.main));
    if (ev && !round_robin) LG_HIP(c, hipEventRecord(ev[2], c->st.main));
    // Plane groups: every hop is cut into P runs of planes and rank g works on group c while rank g + 1 works on group c - 1.  With
    // n / P <= 32 768 columns per launch the four-lanes-per-column kernel takes the hash (a shorter chain per block), which is what
    // makes the (G + P - 1) steps cheaper than G steps over all columns -- measured per rank at S22: 2.26 ms for all 65 536
    // columns in one launch, 1.53 / 1.13 ms per group of 32 768 / 16 384 (tools/chain_probe.py).  0 = choose by the size of the
    // group.  The block layout's chain wraps around (rank G - 1 hands back to rank 0), where a rank would have to send and receive
    // in the same step: one group there.
    uint32_t P = plane_groups;
    const bool auto_groups = P == 0;
    if (auto_groups) P = world <= 2 ? 1 : (world <= 4 ? 2 : 4);
    if (layout != LG_RELAY_CONTIGUOUS) P = 1;
    while (P & (P - 1)) P &= P - 1;                       // a power of two (the planes are)
    while (P > 1 && (P > c->nplanes || (auto_groups && (c->n / P) < 8192))) P >>= 1;
    if (P < 1) P = 1;
    const uint32_t per = c->nplanes / P;
    const size_t group_bytes = (size_t)per * c->ki * LG_HSTATE_BYTES;
    for (size_t i = 0; i < chain.size() && !round_robin; i++) {
        const Link& l = chain[i];
        if (l.owner != rank) continue;
        for (uint32_t g = 0; g < P; g++) {
            uint8_t* gstate = reinterpret_cast<uint8_t*>(c->d_hstate) + (size_t)g * group_bytes;
            const uint32_t gmask = (per >= 32 ? 0xffffffffu : ((1u << per) - 1u)) << (g * per);
            if (exchange && i > 0 && chain[i - 1].owner != rank) {
                { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
                const int rc_ = comm->recv(comm->user, gstate, group_bytes, chain[i - 1].owner, static_cast<void*>(c->st.main));
                if (rc_ != 0) return comm_fail(c, "receive of the column states", rc_);
            }
            if (!(head_done && i == 0)) { const int rc_ = lg_stage_hash_rows(c, gmask, l.local, (uint32_t)l.n, l.pos, col_rows); if (rc_ != LG_OK) return rc_; }
            if (exchange && i + 1 < chain.size() && chain[i + 1].owner != rank) {
                { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
                const int rc_ = comm->send(comm->user, gstate, group_bytes, chain[i + 1].owner, static_cast<void*>(c->st.main));
                if (rc_ != 0) return comm_fail(c, "send of the column states", rc_);
            }
        }
    }
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    if (ev) LG_HIP(c, hipEventRecord(ev[3], c->st.main));
    if (exchange || ((comm->flags & LG_COMM_EXCHANGE_AT_WORLD_1) && comm->broadcast)) {
        { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
        const int rc_ = comm->broadcast(comm->user, c->d_leaves, (uint64_t)c->n * 32, exchange ? chain.back().owner : 0, static_cast<void*>(c->st.main));
        if (rc_ != 0) return comm_fail(c, "broadcast of the leaf digests", rc_);
    }
    if (ev) LG_HIP(c, hipEventRecord(ev[4], c->st.main));
    { const int rc_ = lg_stage_merkle(c); if (rc_ != LG_OK) return rc_; }
    c->held.planes = all;    // every plane of this rank's rows is here (a rank without rows holds the tree only)
    if (ev) { LG_HIP(c, hipEventRecord(ev[5], c->st.main)); c->shard.wait_pairs[c->shard.commits % lg_ctx::kShardProfRing] = 0; c->shard.commits++; }
    return LG_OK;
}

int lg_commit_row_relay(lg_ctx* c, const lg_comm* comm, uint64_t col_rows, int layout, uint32_t plane_groups, const uint64_t* preenc_rows) {
    bool began = false;
    const int rc = commit_row_relay_body(c, comm, col_rows, layout, plane_groups, preenc_rows, &began);
    return (rc != LG_OK && began) ? abandon_staged(c, rc) : rc;
}

// mean milliseconds per stage of the sharded commits since lg_profile_enable(ctx, 1) (at most the last 16): coset mode
// {interpolate, wait for the last piece of the coefficient all-gather, evaluate + hash, digest all-gather, tree}; row relay
// {encode (+ the head's overlapped hash), 0, the relay (waiting for the previous rank, own hash, hand-over), digest broadcast, tree}
int lg_shard_profile_read(lg_ctx* c, float ms_out[5], uint32_t* samples_out) {
    if (!c || !ms_out) return LG_ERR_BAD_ARG;
    if (!c->shard.ev_valid || !c->prof.on || c->shard.commits == 0) return LG_ERR_STATE;
    LG_HIP(c, hipSetDevice(c->device));
    const uint64_t have = std::min<uint64_t>(c->shard.commits, lg_ctx::kShardProfRing);
    double acc[lg_ctx::kShardStages] = {0, 0, 0, 0, 0};
    for (uint64_t s = 0; s < have; s++) {
        hipEvent_t* ev = c->shard.ev[(c->shard.commits - 1 - s) % lg_ctx::kShardProfRing];
        LG_HIP(c, hipEventSynchronize(ev[lg_ctx::kShardStages]));
        for (int i = 0; i < lg_ctx::kShardStages; i++) {
            float ms = 0;
            LG_HIP(c, hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            acc[i] += ms;
        }
        // coset-sharded commits: the stalls of the encode stream waiting for exchange pieces move from "evaluate + hash" to
        // "all-gather" (with one piece on the encode stream itself the collective sits between marks 1 and 2 already)
        // (commits with different numbers of pieces may share the ring: each entry knows its own)
        const uint32_t pairs = c->shard.wait_pairs[(c->shard.commits - 1 - s) % lg_ctx::kShardProfRing];
        if (pairs) {
            double stall = 0;
            for (uint32_t p = 0; p < pairs; p++) {
                float ms = 0;
                LG_HIP(c, hipEventElapsedTime(&ms, ev[lg_ctx::kShardStages + 1 + 2 * p], ev[lg_ctx::kShardStages + 2 + 2 * p]));
                stall += ms;
            }
            acc[1] += stall;
            acc[2] -= stall;
        }
    }
    for (int i = 0; i < lg_ctx::kShardStages; i++) ms_out[i] = (float)(acc[i] / (double)have);
    if (samples_out) *samples_out = (uint32_t)have;
    return LG_OK;
}

}  // extern "C"
