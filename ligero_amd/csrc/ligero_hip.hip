// C ABI implementation (include/ligero_hip.h) of the MI355X Ligero encode-and-commit path.
// Host side: context, device buffers, domain tables, kernel launches on one HIP stream.
// gfx950 only; there is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "../../include/ligero_hip.h"
#include "fr29_gfx950.h"
#include "fr_gfx950.h"
#include "hash_kernels.h"
#include "host_fr.h"
#include "ntt_kernels.h"
#include "ntt_launch.h"
#include "challenge_kernels.h"
#include "generic_path.h"
#include "subproof_kernels.h"

using lg::fr;

// ----------------------------------------------------------------------------- context
struct lg_ctx {
    int device = 0;
    uint32_t rows = 0, k = 0, n = 0, batch = 1;
    int logk = 0, logn = 0;
    // k = O * ki: transforms are done as O-way folded size-ki transforms (ntt_kernels.h); the
    // codeword lives in 8 O planes of [total_rows][ki], column j = (8 O) q + s <-> plane s, slot q
    int logki = 0, logo = 0, lognp = 3;
    uint32_t ki = 0, nplanes = 8;
    uint64_t total_rows = 0;  // batch * rows
    hipStream_t stream = nullptr;    // encode stream; every public call is ordered on it
    hipStream_t stream_h = nullptr;  // column-hash / Merkle stream of the commit pipeline
    hipStream_t stream_up = nullptr;  // host -> device copies of lg_encode_commit's row chunks
    hipStream_t stream_dn = nullptr;  // device -> host copies of the coefficient rows
    static constexpr int kMaxChunks = 8;
    hipEvent_t ev_chunk[kMaxChunks] = {};  // "rows of chunk c are encoded"
    hipEvent_t ev_up[kMaxChunks] = {};     // "rows of chunk c have arrived from the host"
    hipEvent_t ev_coef[kMaxChunks] = {};   // "rows of chunk c are interpolated"
    hipEvent_t ev_done = nullptr;          // "tree of this commit is complete"
    hipEvent_t ev_hashed = nullptr, ev_tree = nullptr;   // single-chunk commits: leaves complete / tree complete (on stream_h)
    // staged hashes (lg_stage_hash_rows) run on stream_h behind everything issued so far on the encode stream and the encode
    // stream does not wait for them until something reads their result (settle_hash): the hash of one row range runs beside
    // the evaluation of the next
    hipEvent_t ev_stage_in = nullptr, ev_stage_hash = nullptr;
    bool hash_pending = false;
    bool async_tree = true;                // LG_ASYNC_TREE=0 turns the overlap below off (A/B knob)
    bool tree_pending = false;             // the tree of the last commit is still being built on stream_h / stream_t
    // single-chunk commits: the column hash of commit i runs on stream_h BESIDE the interpolation / evaluation of commit
    // i + 1 (one wave per SIMD of a latency chain fills issue slots instead of holding the machine): U is double buffered
    bool async_hash = true;                // LG_ASYNC_HASH=0: hash on the encode stream, one U buffer (A/B knob)
    // Overlapped single-chunk commits rotate through a ring of U / leaf / node buffers: two deep, or three deep with a second hash
    // stream when U is small (a lone small proof is a chain of dependent Blake2s compressions: two of those chains in flight)
    static constexpr int kRing = 3;
    fr* d_u_pp[kRing] = {nullptr, nullptr, nullptr};    // [0] = d_u_alloc, the others allocated by the first overlapped commits
    int u_parity = 0;                                   // ring slot of the current commitment
    int ring_depth = 2;
    uint64_t async_seq = 0;                             // overlapped commits issued: picks the hash stream
    hipStream_t stream_h2 = nullptr;                    // second hash stream (ring_depth == 3)
    hipEvent_t ev_hash_free[kRing] = {nullptr, nullptr, nullptr};   // "the hash that read U[p] is done" (on its hash stream)
    uint4* d_hstate = nullptr;             // [batch][np][ki][lg::kColStateVec] Blake2s state between row chunks / ranks (LG_BUF_HSTATE)
    gf_state* gf = nullptr;                // set for contexts over a generic field (lg_ctx_create_field): every supported
                                           // entry point forwards to generic_path.hip, the others return LG_ERR_UNSUPPORTED
    lg_ctx* aux2k = nullptr;               // tables of the size-2k domain (intermediate_domain, mod.rs:212), created on demand
    fr* d_sub_partial = nullptr; size_t sub_partial_elems = 0;  // row-sum partials of the sub-proof polynomials
    fr* d_sub_q = nullptr;                 // [2k] evaluations / coefficients
    fr* d_sub_r = nullptr; size_t sub_r_elems = 0;              // challenge vector
    fr r3;                                 // 2^768 mod p
    // constraint matrix A in CSC form (lg_upload_constraint_matrix) and the device-side challenge generator
    uint32_t* d_a_colptr = nullptr; uint32_t* d_a_row = nullptr; fr* d_a_val = nullptr;
    uint32_t* d_a_heavy = nullptr; uint32_t a_nheavy = 0;   // columns with more than lg::kHeavyColumn entries
    uint32_t* d_a_seg = nullptr; uint32_t a_nseg = 0;       // their segments: [seg_begin | seg_end | heavy_seg_ptr] (challenge_kernels.h)
    fr* d_a_seg_partial = nullptr;                          // [batch][a_nseg]
    uint64_t a_rows = 0, a_nnz = 0; bool a_loaded = false;
    // gate map of the circuit (lg_upload_gate_map): for every position of the solution vector the sources of x and y
    uint32_t* d_gate_l = nullptr; uint32_t* d_gate_r = nullptr; fr* d_gate_consts = nullptr;
    uint64_t gate_npos = 0; uint32_t gate_nconst = 0; bool gate_backward = false;
    uint32_t* d_seeds = nullptr;           // [batch][8]
    uint32_t* d_cc_counts = nullptr; size_t cc_counts_cap = 0;
    uint32_t* d_short_flag = nullptr;
    fr* d_rlin = nullptr; size_t rlin_elems = 0;   // r_linear [batch][4mk]
    uint32_t force_chunks = 0;             // LG_FORCE_CHUNKS (testing knob): pipeline depth regardless of size
    uint64_t quad_hash_max_columns = 32768; // single-chunk commits with at most this many columns use the four-lanes-per-column
                                           // Blake2s (LG_HASH_QUAD_MAX_COLUMNS overrides; 0 = never)
    // resident commitment
    fr* d_preenc = nullptr;   // [total_rows][k]  Montgomery
    fr* d_coeffs = nullptr;   // [total_rows][k]  Montgomery
    fr* d_u = nullptr;        // [8 O][total_rows][ki] canonical integers; planes 8c hold the message.  In a sharded
                              // context only planes [own_plane0, own_plane0 + own_planes) exist and this is the VIRTUAL
                              // base d_u_alloc - own_plane0 * plane, so kernels keep indexing by absolute plane id
    fr* d_u_alloc = nullptr;      // what hipMalloc returned for d_u
    fr* d_preenc_alloc = nullptr; // sharded context: allocation behind the rows [pre_row0, pre_row0 + pre_rows) of d_preenc
    // sharded (lg_ctx_create_sharded) single-proof context: one rank of a proof that is split over several GPUs
    bool sharded = false;
    uint32_t own_plane0 = 0, own_planes = 0;   // planes this context can hold (all of them unless sharded)
    uint32_t coeff_rows_alloc = 0;             // rows of LG_BUF_COEFFS (>= rows: padding for equal all-gather shards)
    uint32_t pre_row0 = 0, pre_rows = 0;       // sharded: rows of d_preenc that are allocated
    // what the resident commitment covers (a staged commit may hold only some planes / message rows)
    uint32_t have_planes = 0;                  // mask of the planes of d_u that belong to the current commitment
    uint32_t have_row0 = 0, have_row1 = 0;     // message rows [have_row0, have_row1) of d_preenc that belong to it
    uint8_t* d_leaves = nullptr;  // [batch][n][32]    (of the current commitment: one of d_leaves_pp)
    // Single-chunk commits with the hash overlap on: leaves and nodes are double-buffered like U, and the tree has a stream of its
    // own, so that the tree of commit i runs beside the column hash of commit i + 1 (a stream of small commitments is bound by
    // the longest of its three chains -- encode, hash, tree -- instead of hash + tree)
    uint8_t* d_leaves_pp[3] = {nullptr, nullptr, nullptr};
    uint8_t* d_nodes_pp[3] = {nullptr, nullptr, nullptr};
    hipStream_t stream_t = nullptr;
    hipEvent_t ev_leaves_free[3] = {nullptr, nullptr, nullptr};   // "the tree that read leaves[p] / wrote nodes[p] is done" (on stream_t)
    uint8_t* d_digest_xchg = nullptr;   // [world][ki][planes per rank][32] staging of the sharded commit's digest all-gather
    uint8_t* d_nodes = nullptr;   // [batch][n-1][32]
    // domain tables: 29-bit limbs, value * 2^261 mod p, three planes each (limbs 0-3 | 4-7 | 8)
    uint8_t* d_tw_fwd = nullptr;    // butterfly twiddles of the size-ki transform, pass order (lg::pass_tw_offset)
    uint8_t* d_tw_inv = nullptr;    // same for the inverse transform
    uint8_t* d_coset_tw = nullptr;  // [plane s < 8 O][d < k] = omega_n^(s d)
    uint8_t* d_fold_inv = nullptr;  // O = 2: omega_k^-d, d < ki (with quotients); O = 4: [h < O][d < k] = omega_k^(-h d) / k
    uint8_t* d_first2 = nullptr;    // log2 k = 1 (mod 3): coefficients of the dot-product radix-2 first pass (ntt_kernels.h)
    uint32_t n_pass_tw = 0;
    lg::f29 w8_fwd[3], w8_inv[3], w8q_fwd[3], w8q_inv[3], one29, oneq29, scale29, invk29, invkq29;
    fr r2;
    // scratch for row operators / openings (grown on demand)
    fr* d_scratch_a = nullptr; size_t scratch_a_elems = 0;  // inputs / coefficients
    fr* d_scratch_b = nullptr; size_t scratch_b_elems = 0;  // planes / outputs
    fr* d_scratch_c = nullptr; size_t scratch_c_elems = 0;  // natural-order output
    uint32_t* d_idx = nullptr; size_t idx_cap = 0;
    uint8_t* d_path_out = nullptr; size_t path_cap = 0;
    // rows whose message planes (s = 0 mod 8) the interpolation of the staged commit in progress wrote itself (sorted, disjoint):
    // the evaluation skips those planes for them
    std::vector<std::pair<uint32_t, uint32_t>> canon_ranges;
    // lg_commit_sharded: the rows this rank owns are `shard_pieces` ranges, kept compact (piece-major) in d_preenc_alloc
    uint32_t shard_pieces = 0, shard_world = 0, shard_rank = 0, compact_rows = 0;
    hipStream_t stream_x = nullptr;        // exchange stream of lg_commit_sharded: the all-gather of piece c + 1 beside the evaluation of piece c
    static constexpr int kShardStages = 5;
    static constexpr int kShardProfRing = 16;
    hipEvent_t ev_shard[kShardProfRing][kShardStages + 1 + 2 * kMaxChunks] = {};   // stage marks, then (before, after) of every wait for an exchange piece
    bool ev_shard_valid = false;
    uint64_t shard_commits = 0;
    uint32_t shard_wait_pairs[kShardProfRing] = {};   // per profiled commit: exchange pieces it waited for (0: row relay)
    bool committed = false;
    bool staging = false;                  // between lg_stage_interpolate and lg_stage_merkle
    bool profiling = false;
    static constexpr int kProfRing = 64;                 // commits remembered by the profiler
    hipEvent_t ev[kProfRing][6] = {};  // 0 start, 1 interpolate done, 2 evaluate done | hash stream: 3 first hash start, 4 last hash done, 5 tree done
    bool ev_valid = false;
    uint64_t prof_commits = 0;                            // commits recorded since lg_profile_enable(1)
    char err[256] = {0};
};

static int fail_hip(lg_ctx* c, hipError_t e, const char* what) {
    if (c) snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString(e));
    return (e == hipErrorOutOfMemory) ? LG_ERR_OOM : LG_ERR_HIP;
}
#define LG_HIP(c, call)                                   \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return fail_hip(c, e_, #call); \
    } while (0)

static uint32_t all_planes_mask(const lg_ctx* c) { return c->nplanes >= 32 ? 0xffffffffu : ((1u << c->nplanes) - 1u); }
static uint32_t own_planes_mask(const lg_ctx* c) {
    const uint32_t hi = c->own_plane0 + c->own_planes;   // <= 32
    const uint32_t upto = hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u);
    return upto & ~((1u << c->own_plane0) - 1u);
}
// A staged (coset-sharded) commit leaves only some planes of U and some message rows on this device: entry points
// that would read the others fail with LG_ERR_STATE instead of returning stale or foreign data.
static int need_planes(lg_ctx* c, uint32_t mask, const char* what) {
    if ((c->have_planes & mask) == mask) return LG_OK;
    snprintf(c->err, sizeof(c->err), "%s needs coset planes 0x%x of the commitment, this context holds 0x%x (staged / sharded commit)", what, mask,
             c->have_planes);
    return LG_ERR_STATE;
}
static int need_all_message_rows(lg_ctx* c, const char* what) {
    if (c->have_row0 == 0 && c->have_row1 == c->rows) return LG_OK;
    snprintf(c->err, sizeof(c->err), "%s needs every row of preenc_u, this context holds rows [%u, %u) of %u", what, c->have_row0, c->have_row1, c->rows);
    return LG_ERR_STATE;
}

// every kernel launch is followed by its own error check (a failed launch must not be reported against a later one)
#define LG_LAUNCH(c, ...)                  \
    do {                                   \
        hipLaunchKernelGGL(__VA_ARGS__);   \
        LG_HIP(c, hipGetLastError());      \
    } while (0)

// Montgomery-form (2^256) host element -> 29-bit limbs of value * 2^261 mod p
static lg::f29 to_f29(const lg_host::Fr& a_mont) {
    static const lg_host::Fr m32 = lg_host::to_mont(lg_host::Fr{{32, 0, 0, 0}});
    const lg_host::Fr t = lg_host::mul(a_mont, m32);  // raw limbs now read (a * 2^5) * 2^256 = a * 2^261 mod p
    lg::f29 r;
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 6, sh = bit & 63;
        uint64_t x = t.l[w] >> sh;
        if (sh > 35 && w < 3) x |= t.l[w + 1] << (64 - sh);
        r.v[i] = (i < 8) ? (uint32_t)(x & 0x1fffffffu) : (uint32_t)x;
    }
    return r;
}
// canonical value of a Montgomery-form host element -> 29-bit limbs (plain operand of shoup29)
static lg::f29 split29(const uint64_t (&t)[5]) {
    lg::f29 r;
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 6, sh = bit & 63;
        uint64_t x = t[w] >> sh;
        if (sh > 35) x |= t[w + 1] << (64 - sh);
        r.v[i] = (uint32_t)(x & 0x1fffffffu);
    }
    return r;
}
static lg::f29 to_f29_plain(const lg_host::Fr& a_mont) {
    const lg_host::Fr a = lg_host::from_mont(a_mont);
    const uint64_t t[5] = {a.l[0], a.l[1], a.l[2], a.l[3], 0};
    return split29(t);
}
// Barrett quotient floor(w * 2^261 / p) of the canonical value w < p (second operand of shoup29)
static lg::f29 to_f29_quot(const lg_host::Fr& a_mont) {
    lg_host::Fr rem = lg_host::from_mont(a_mont);
    uint64_t q[5] = {0, 0, 0, 0, 0};
    for (int bit = 260; bit >= 0; bit--) {
        // rem < p < 2^254: the doubled value fits four words
        rem = lg_host::Fr{{rem.l[0] << 1, (rem.l[1] << 1) | (rem.l[0] >> 63), (rem.l[2] << 1) | (rem.l[1] >> 63), (rem.l[3] << 1) | (rem.l[2] >> 63)}};
        if (lg_host::geq(rem, lg_host::kP)) {
            rem = lg_host::sub_raw(rem, lg_host::kP);
            q[bit >> 6] |= 1ull << (bit & 63);
        }
    }
    return split29(q);
}
// three-plane table image of `count` constants
static void fill_planes(std::vector<uint8_t>& img, size_t count, size_t e, const lg::f29& v) {
    uint32_t* lo = reinterpret_cast<uint32_t*>(img.data());
    uint32_t* mid = lo + 4 * count;
    uint32_t* hi = mid + 4 * count;
    for (int i = 0; i < 4; i++) { lo[4 * e + i] = v.v[i]; mid[4 * e + i] = v.v[4 + i]; }
    hi[e] = v.v[8];
}
static lg::Tw29 planes_of(const uint8_t* base, size_t count) {
    lg::Tw29 t;
    t.lo = reinterpret_cast<const uint4*>(base);
    t.mid = reinterpret_cast<const uint4*>(base + 16 * count);
    t.hi = reinterpret_cast<const uint32_t*>(base + 32 * count);
    return t;
}
// six-plane image (constants, then their quotients) for shoup29: 72 bytes per constant
static void fill_planes_q(std::vector<uint8_t>& img, size_t count, size_t e, const lg_host::Fr& a_mont) {
    fill_planes(img, count, e, to_f29_plain(a_mont));
    const lg::f29 q = to_f29_quot(a_mont);
    uint32_t* lo = reinterpret_cast<uint32_t*>(img.data() + 36 * count);
    uint32_t* mid = lo + 4 * count;
    uint32_t* hi = mid + 4 * count;
    for (int i = 0; i < 4; i++) { lo[4 * e + i] = q.v[i]; mid[4 * e + i] = q.v[4 + i]; }
    hi[e] = q.v[8];
}
static lg::Tw29q planes_q_of(const uint8_t* base, size_t count) {
    lg::Tw29q t;
    t.w = planes_of(base, count);
    t.q = planes_of(base + 36 * count, count);
    return t;
}

// flags of the cross-stream "chunk encoded" / "tree done" events (LG_EVENT_FLAGS overrides, for experiments)
static unsigned lg_event_flags() {
    if (const char* f = getenv("LG_EVENT_FLAGS")) return (unsigned)strtoul(f, nullptr, 0);
    return hipEventDisableTiming;
}

static fr to_dev(const lg_host::Fr& a) {
    fr r;
    for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (uint32_t)a.l[i];
        r.v[2 * i + 1] = (uint32_t)(a.l[i] >> 32);
    }
    return r;
}
static int ilog2_exact(uint32_t x) {
    if (x == 0 || (x & (x - 1))) return -1;
    int l = 0;
    while ((1u << l) < x) l++;
    return l;
}

// ----------------------------------------------------------------------------- small kernels
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
        a.cols += (uint64_t)p * ncol_elems;
        a.sib += 32 * (uint64_t)p * a.t;
        a.paths += 32 * (uint64_t)p * a.t * plen;
    }
    if (gid < ncol_elems) {
        const uint32_t c = (uint32_t)(gid / a.rows), i = (uint32_t)(gid % a.rows);
        const uint32_t j = a.idx[c];
        const uint32_t s = j & ((1u << a.lognp) - 1), q = j >> a.lognp;
        fr x = fr_load(a.u + (uint64_t)s * a.plane_stride + (a.row_base + i) * a.k + q);
        fr y, z;
        fr_mul_lazy(y, x, a.r2);
        fr_reduce(z, y);
        fr_store(a.cols + gid, z);
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

// a1 on the device (mod.rs:483-516): x, y, z are functions of w and the circuit's wiring -- x[p] / y[p] = the values of the
// operands of the Mul gate at position p (a position of w, or a constant that has no position), z[p] = w[p]; zero elsewhere
struct WitnessGatherArgs {
    fr* pre;                 // [batch][4 m][k]: blocks X, Y, Z, W; W is read, X / Y / Z are written
    const uint32_t* left;    // [m k]: kGateNone, kGateConst | index into consts, or a position of w
    const uint32_t* right;
    const fr* consts;
    uint64_t mk;             // m * k
    uint64_t pos0, pos1;     // positions [pos0, pos1) of every proof
    uint32_t batch;
};
constexpr uint32_t kGateNone = 0xffffffffu, kGateConst = 0x80000000u;
__global__ void __launch_bounds__(256) witness_gather_kernel(const WitnessGatherArgs a) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t span = a.pos1 - a.pos0;
    if (gid >= span * a.batch) return;
    const uint64_t pos = a.pos0 + gid % span;
    fr* base = a.pre + (gid / span) * 4 * a.mk;
    const fr* w = base + 3 * a.mk;
    const uint32_t l = a.left[pos], r = a.right[pos];
    fr x, y, z;
    if (l == kGateNone) {
#pragma unroll
        for (int i = 0; i < 8; i++) x.v[i] = 0;
        y = x; z = x;
    } else {
        x = (l & kGateConst) ? fr_load(a.consts + (l & ~kGateConst)) : fr_load(w + l);
        y = (r & kGateConst) ? fr_load(a.consts + (r & ~kGateConst)) : fr_load(w + r);
        z = fr_load(w + pos);
    }
    fr_store(base + pos, x);
    fr_store(base + a.mk + pos, y);
    fr_store(base + 2 * a.mk + pos, z);
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

// ----------------------------------------------------------------------------- launches
static lg::NttArgs interp_args(const lg_ctx* c, const fr* in, fr* out, fr* canon_out, uint32_t row0, uint32_t rows) {
    lg::NttArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.out = out; a.canon_out = canon_out;
    a.tw = planes_q_of(c->d_tw_inv, c->n_pass_tw ? c->n_pass_tw : 1);
    if (c->logo == 1) {
        a.coset_tw = planes_q_of(c->d_fold_inv, c->ki);  // the odd half's factors of the radix-2 fold
    } else {
        a.coset_tw.w = planes_of(c->d_fold_inv, (size_t)c->k << c->logo);
        a.coset_tw.q = a.coset_tw.w;  // unused: the fold is a Montgomery dot product
    }
    a.first2 = a.coset_tw.w;      // unused
    for (int i = 0; i < 3; i++) { a.w8[i] = c->w8_inv[i]; a.w8q[i] = c->w8q_inv[i]; }
    a.one = c->one29;
    a.oneq = c->oneq29;
    a.invk = c->invk29;
    a.invkq = c->invkq29;
    a.chunk_rows = rows ? rows : 1;  // contiguous rows unless the caller narrows it (commit pipeline)
    a.proof_stride = 0;
    a.scale = c->scale29;
    a.rows = rows; a.row0 = row0; a.ncos = 0;
    a.plane_stride = 0;
    a.canon_mask = 0xffffffffu;
    return a;
}
// with_message: also evaluate the planes that coincide with the message (needed when only
// coefficients are given: lg_reed_solomon_evaluate); the commit path copies the message instead
static lg::NttArgs eval_args(const lg_ctx* c, const fr* coeffs, fr* planes, uint64_t plane_stride, uint32_t row0, uint32_t rows,
                             bool with_message) {
    lg::NttArgs a;
    memset(&a, 0, sizeof(a));
    a.in = coeffs; a.out = planes; a.canon_out = nullptr;
    a.tw = planes_q_of(c->d_tw_fwd, c->n_pass_tw ? c->n_pass_tw : 1);
    if (c->logo == 0) {
        a.coset_tw = planes_q_of(c->d_coset_tw, (size_t)c->k * c->nplanes);
    } else {
        a.coset_tw.w = planes_of(c->d_coset_tw, (size_t)c->k * c->nplanes);
        a.coset_tw.q = a.coset_tw.w;  // unused: the fold is a Montgomery dot product
    }
    a.first2 = planes_of(c->d_first2, (size_t)c->nplanes * 2 * c->k);
    for (int i = 0; i < 3; i++) { a.w8[i] = c->w8_fwd[i]; a.w8q[i] = c->w8q_fwd[i]; }
    a.one = c->one29;
    a.oneq = c->oneq29;
    a.invk = c->invk29;
    a.invkq = c->invkq29;
    a.chunk_rows = rows ? rows : 1;  // contiguous rows unless the caller narrows it (commit pipeline)
    a.proof_stride = 0;
    a.scale = c->scale29;
    a.rows = rows; a.row0 = row0;
    a.ncos = 0;
    for (uint32_t s = 0; s < c->nplanes; s++)
        if (with_message || (s & 7) != 0) a.cosets[a.ncos++] = (uint8_t)s;
    a.plane_stride = plane_stride;
    return a;
}

static int grow(lg_ctx* c, fr** p, size_t* cap, size_t need) {
    if (*cap >= need) return LG_OK;
    if (*p) LG_HIP(c, hipFree(*p));
    *p = nullptr;
    *cap = 0;
    LG_HIP(c, hipMalloc(reinterpret_cast<void**>(p), need * sizeof(fr)));
    *cap = need;
    return LG_OK;
}

// the four-lanes-per-column kernel's view of a column-hash launch (same rows, same parked state)
static lg::ColHashQuadArgs quad_args_of(const lg::ColHashArgs& h) {
    lg::ColHashQuadArgs qa;
    memset(&qa, 0, sizeof(qa));
    qa.u = h.u; qa.leaves = h.leaves; qa.rows = h.rows; qa.k = h.k; qa.lognp = h.lognp;
    qa.proof_begin = h.proof_begin; qa.proof_count = h.proof_count; qa.plane_begin = h.plane_begin; qa.plane_count = h.plane_count;
    qa.plane_stride = h.plane_stride;
    qa.state = h.state; qa.row_begin = h.row_begin; qa.row_end = h.row_end; qa.first = h.first; qa.last = h.last;
    qa.col_pos = h.col_pos; qa.col_rows = h.col_rows;
    return qa;
}
// ... which needs an even position in the column and, unless it finalises, an even number of rows (whole 64-byte blocks)
static bool quad_can_take(const lg::ColHashArgs& h) {
    const uint32_t nrows = h.row_end - h.row_begin;
    return (h.col_pos & 1) == 0 && nrows > 0 && (h.last || (nrows & 1) == 0);
}

// The commit is pipelined over row chunks on two streams: while the encode stream evaluates
// chunk c+1, the hash stream absorbs chunk c into the per-column Blake2s states.  The column
// hash is latency bound (one sequential chain per column, only n columns), so it fills issue
// slots the NTT kernel leaves idle instead of extending the critical path.
struct Chunk {
    uint32_t proof_begin, proof_count, row_begin, row_end;
};
static int plan_chunks(const lg_ctx* c, Chunk* out, bool from_host = false) {
    int n = 0;
    // a chunk must be big enough (>= 2^25 codeword elements, a few ms of encoding) for the extra
    // launches and cross-stream waits to pay; small commits run as one chunk
    const uint64_t elems = c->total_rows * c->n;
    uint32_t want = c->force_chunks ? c->force_chunks : (uint32_t)(elems >> 25);
    if (from_host && !c->force_chunks) {
        // streamed input: the PCIe copy of chunk c+1 hides behind the encoding of chunk c, which pays
        // from ~16 MiB of input per chunk; at least 4 chunks so that the exposed first copy is short
        const uint64_t bytes = c->total_rows * c->k * sizeof(fr);
        const uint32_t up = bytes >= (64ull << 20) ? 8 : (bytes >= (16ull << 20) ? 4 : 1);
        if (up > want) want = up;
    }
    if (want > (uint32_t)lg_ctx::kMaxChunks) want = lg_ctx::kMaxChunks;
    if (want <= 1) {
        out[0] = Chunk{0, c->batch, 0, c->rows};
        return 1;
    }
    // Row ranges, each covering the same rows of EVERY proof of the batch: the hash kernel keeps all
    // batch * n column chains busy in every chunk and only their length shrinks.  (Splitting a batch
    // by proofs instead halves the number of chains per launch but not their length, which is what a
    // latency-bound kernel's duration follows: measured 1.35 ms vs 1.19 ms unsplit on the Poseidon
    // batch.)  Boundaries on even rows: a Blake2s block holds two rows.
    uint32_t parts = want;
    const uint32_t pairs = c->rows / 2;
    if (parts > pairs) parts = pairs;
    if (parts <= 1) {
        out[0] = Chunk{0, c->batch, 0, c->rows};
        return 1;
    }
    // The hash of the LAST chunk has nothing left to hide behind, so the chunks taper: weights 4, 4, ..., 4, 3, 1
    // (the exposed tail is 1/(4 parts - 4) of the hash instead of 1/parts).  LG_CHUNK_TAPER=0: equal chunks.
    static const bool taper = [] { const char* e = getenv("LG_CHUNK_TAPER"); return !e || atoi(e) != 0; }();
    std::vector<uint32_t> w(parts, 4);
    if (taper && parts >= 4 && pairs >= 8 * parts) { w[parts - 2] = 3; w[parts - 1] = 1; }   // (3,2,1 / 2,1 tails measured the same)
    uint64_t total = 0, acc = 0;
    for (uint32_t x : w) total += x;
    uint32_t r0 = 0;
    for (uint32_t i = 0; i < parts; i++) {
        acc += w[i];
        uint32_t r1 = (i + 1 == parts) ? c->rows : 2 * (uint32_t)((uint64_t)pairs * acc / total);
        if (r1 <= r0) r1 = r0 + 2;
        out[n++] = Chunk{0, c->batch, r0, r1};
        r0 = r1;
    }
    return n;
}

static int read_back(lg_ctx* c, void* dst, const void* src, size_t bytes);

// ----------------------------------------------------------------------------- ABI
extern "C" {

const char* lg_status_string(int s) {
    switch (s) {
        case LG_OK: return "ok";
        case LG_ERR_BAD_ARG: return "bad argument";
        case LG_ERR_BAD_DIMS: return "bad dimensions";
        case LG_ERR_NO_DEVICE: return "no such HIP device";
        case LG_ERR_HIP: return "HIP runtime error";
        case LG_ERR_OOM: return "out of memory";
        case LG_ERR_STATE: return "invalid call order";
        case LG_ERR_UNSUPPORTED: return "unsupported shape";
        case LG_ERR_COMM: return "communication callback failed";
        default: return "unknown status";
    }
}
const char* lg_last_error(const lg_ctx* c) { return c ? c->err : ""; }
uint32_t lg_abi_version(void) { return LG_ABI_VERSION; }

void lg_ctx_destroy(lg_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->stream_h) hipStreamSynchronize(c->stream_h);
    if (c->stream_up) hipStreamSynchronize(c->stream_up);
    if (c->stream_dn) hipStreamSynchronize(c->stream_dn);
    if (c->stream_t) hipStreamSynchronize(c->stream_t);
    if (c->stream_h2) hipStreamSynchronize(c->stream_h2);
    if (c->stream_x) hipStreamSynchronize(c->stream_x);
    if (c->aux2k) lg_ctx_destroy(c->aux2k);
    hipSetDevice(c->device);
    if (c->gf) gf_destroy(c->gf);
    for (void* b : {(void*)c->d_gate_l, (void*)c->d_gate_r, (void*)c->d_gate_consts})
        if (b) hipFree(b);
    void* bufs2[] = {c->d_digest_xchg, c->d_sub_partial, c->d_sub_q, c->d_sub_r, c->d_a_colptr, c->d_a_row, c->d_a_val, c->d_a_heavy, c->d_a_seg, c->d_a_seg_partial, c->d_seeds, c->d_cc_counts, c->d_short_flag, c->d_rlin};
    for (void* b : bufs2)
        if (b) hipFree(b);
    void* bufs[] = {c->sharded ? c->d_preenc_alloc : c->d_preenc, c->d_coeffs, c->d_u_alloc, c->d_leaves_pp[0], c->d_nodes_pp[0], c->d_leaves_pp[1], c->d_nodes_pp[1], c->d_leaves_pp[2], c->d_nodes_pp[2], c->d_tw_fwd, c->d_tw_inv, c->d_coset_tw, c->d_fold_inv, c->d_first2,
                    c->d_scratch_a, c->d_scratch_b, c->d_scratch_c, c->d_idx, c->d_path_out, c->d_hstate};
    for (void* b : bufs)
        if (b) hipFree(b);
    if (c->ev_valid)
        for (auto& set : c->ev)
            for (auto& e : set) hipEventDestroy(e);
    for (auto& e : c->ev_chunk)
        if (e) hipEventDestroy(e);
    for (auto& e : c->ev_up)
        if (e) hipEventDestroy(e);
    for (auto& e : c->ev_coef)
        if (e) hipEventDestroy(e);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->d_u_pp[1]) hipFree(c->d_u_pp[1]);
    if (c->d_u_pp[2]) hipFree(c->d_u_pp[2]);
    for (auto& e : c->ev_hash_free)
        if (e) hipEventDestroy(e);
    if (c->ev_hashed) hipEventDestroy(c->ev_hashed);
    if (c->ev_tree) hipEventDestroy(c->ev_tree);
    if (c->ev_stage_in) hipEventDestroy(c->ev_stage_in);
    if (c->ev_stage_hash) hipEventDestroy(c->ev_stage_hash);
    for (auto& e : c->ev_leaves_free)
        if (e) hipEventDestroy(e);
    if (c->ev_shard_valid)
        for (auto& set : c->ev_shard)
            for (auto& e : set) hipEventDestroy(e);
    if (c->stream_x) hipStreamDestroy(c->stream_x);
    if (c->stream_t) hipStreamDestroy(c->stream_t);
    if (c->stream_h2) hipStreamDestroy(c->stream_h2);
    if (c->stream_up) hipStreamDestroy(c->stream_up);
    if (c->stream_dn) hipStreamDestroy(c->stream_dn);
    if (c->stream_h) hipStreamDestroy(c->stream_h);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

}  // extern "C"

struct ShardSpec {
    uint32_t plane_begin, plane_count, coeff_rows_alloc;
};
static int ctx_create_impl(lg_ctx** out, int device, uint32_t rows, uint32_t k, uint32_t n, uint32_t batch, const ShardSpec* shard) {
    if (!out) return LG_ERR_BAD_ARG;
    *out = nullptr;
    const int logk = ilog2_exact(k), logn = ilog2_exact(n);
    if (rows == 0 || batch == 0 || logk < 1 || logn < 0 || n != 8 * (uint64_t)k || logn > lg_host::kTwoAdicity) return LG_ERR_BAD_DIMS;
    if (logk > 14) return LG_ERR_UNSUPPORTED;
    if ((uint64_t)rows * batch > 0xffffffffull / 8) return LG_ERR_BAD_DIMS;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LG_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return LG_ERR_NO_DEVICE;
    lg_ctx* c = new (std::nothrow) lg_ctx();
    if (!c) return LG_ERR_OOM;
    c->device = device; c->rows = rows; c->k = k; c->n = n; c->batch = batch; c->logk = logk; c->logn = logn;
    c->total_rows = (uint64_t)rows * batch;
    // whole rows stay in LDS up to k = 4096 (one workgroup per CU, 188 VGPRs: the column-hash
    // waves of the commit pipeline still fit beside it); larger k folds an outer radix 2 or 4
    c->logki = logk <= 12 ? logk : 12;
    c->logo = logk - c->logki;
    c->ki = 1u << c->logki;
    c->nplanes = 8u << c->logo;
    c->lognp = 3 + c->logo;
    if (const char* fc = getenv("LG_FORCE_CHUNKS")) c->force_chunks = (uint32_t)atoi(fc);
    if (const char* qm = getenv("LG_HASH_QUAD_MAX_COLUMNS")) c->quad_hash_max_columns = strtoull(qm, nullptr, 0);
    c->own_plane0 = 0; c->own_planes = c->nplanes; c->coeff_rows_alloc = (uint32_t)c->total_rows;
    if (shard) {
        if (batch != 1 || shard->plane_count == 0 || (uint64_t)shard->plane_begin + shard->plane_count > c->nplanes ||
            shard->coeff_rows_alloc < rows) {
            delete c;
            return LG_ERR_BAD_ARG;
        }
        c->sharded = true;
        c->own_plane0 = shard->plane_begin; c->own_planes = shard->plane_count; c->coeff_rows_alloc = shard->coeff_rows_alloc;
    }
    int rc = LG_OK;
    auto body = [&]() -> int {
        LG_HIP(c, hipSetDevice(device));
        LG_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        LG_HIP(c, hipStreamCreateWithFlags(&c->stream_h, hipStreamNonBlocking));
        LG_HIP(c, hipStreamCreateWithFlags(&c->stream_up, hipStreamNonBlocking));
        LG_HIP(c, hipStreamCreateWithFlags(&c->stream_dn, hipStreamNonBlocking));
        LG_HIP(c, hipStreamCreateWithFlags(&c->stream_t, hipStreamNonBlocking));
        LG_HIP(c, hipStreamCreateWithFlags(&c->stream_h2, hipStreamNonBlocking));
        for (auto& e : c->ev_leaves_free) LG_HIP(c, hipEventCreateWithFlags(&e, lg_event_flags()));
        for (auto& e : c->ev_chunk) LG_HIP(c, hipEventCreateWithFlags(&e, lg_event_flags()));
        for (auto& e : c->ev_up) LG_HIP(c, hipEventCreateWithFlags(&e, lg_event_flags()));
        for (auto& e : c->ev_coef) LG_HIP(c, hipEventCreateWithFlags(&e, lg_event_flags()));
        LG_HIP(c, hipEventCreateWithFlags(&c->ev_done, lg_event_flags()));
        LG_HIP(c, hipEventCreateWithFlags(&c->ev_hashed, lg_event_flags()));
        LG_HIP(c, hipEventCreateWithFlags(&c->ev_tree, lg_event_flags()));
        LG_HIP(c, hipEventCreateWithFlags(&c->ev_stage_in, lg_event_flags()));
        LG_HIP(c, hipEventCreateWithFlags(&c->ev_stage_hash, lg_event_flags()));
        if (const char* e = getenv("LG_ASYNC_TREE")) c->async_tree = atoi(e) != 0;
        if (const char* e = getenv("LG_ASYNC_HASH")) c->async_hash = atoi(e) != 0;
        for (auto& e : c->ev_hash_free) LG_HIP(c, hipEventCreateWithFlags(&e, lg_event_flags()));
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_hstate), (size_t)batch * n * sizeof(uint4) * lg::kColStateVec));
        const size_t mat = (size_t)c->total_rows * k;
        // sharded: the message rows arrive shard by shard (lg_stage_interpolate allocates what it is given), the
        // coefficient buffer is padded so that equal all-gather shards fit, and only the owned planes of U exist
        if (!c->sharded) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_preenc), mat * sizeof(fr)));
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_coeffs), (size_t)c->coeff_rows_alloc * k * sizeof(fr)));
        {
            const size_t plane = (size_t)c->total_rows * c->ki;
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_u_alloc), (size_t)c->own_planes * plane * sizeof(fr)));
            c->d_u = c->d_u_alloc - (size_t)c->own_plane0 * plane;   // never dereferenced outside the owned planes
            c->d_u_pp[0] = c->d_u_alloc;
            // a third ring slot costs one more U: only where U is small, i.e. where a commit is latency-bound (LG_RING_DEPTH overrides)
            c->ring_depth = ((size_t)c->nplanes * plane * sizeof(fr) <= (size_t{64} << 20)) ? 3 : 2;
            if (const char* rd = getenv("LG_RING_DEPTH")) { const int v = atoi(rd); if (v == 2 || v == 3) c->ring_depth = v; }
        }
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_leaves), (size_t)batch * n * 32));
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_nodes), (size_t)batch * (n - 1) * 32));
        c->d_leaves_pp[0] = c->d_leaves;
        c->d_nodes_pp[0] = c->d_nodes;
        // an unsharded context owns the whole of d_preenc from the start (a zero-copy producer may fill LG_BUF_PREENC and call
        // lg_commit_resident); only a staged commit in progress narrows the range
        if (!c->sharded) { c->have_row0 = 0; c->have_row1 = rows; }
        // domain tables: large_domain (size n) generator wn; small_domain generator wk = wn^8 (mod.rs:89, 204-211)
        using namespace lg_host;
        const Fr wn = domain_generator(logn);
        const Fr wk = domain_generator(logk);
        const Fr wk_inv = inverse(wk);
        std::vector<Fr> pn(n), pki(c->ki), pki_inv(c->ki), pk_inv(k);  // powers of wn, w_ki, w_ki^-1, wk^-1
        {
            Fr a = kOneMont;
            for (uint32_t e = 0; e < n; e++) { pn[e] = a; a = mul(a, wn); }
            const Fr wki = pow_u64(wk, 1ull << c->logo), wki_inv = pow_u64(wk_inv, 1ull << c->logo);
            a = kOneMont;
            Fr b = kOneMont;
            for (uint32_t e = 0; e < c->ki; e++) { pki[e] = a; pki_inv[e] = b; a = mul(a, wki); b = mul(b, wki_inv); }
            a = kOneMont;
            for (uint32_t e = 0; e < k; e++) { pk_inv[e] = a; a = mul(a, wk_inv); }
        }
        const Fr inv_k = inverse(to_mont(Fr{{k, 0, 0, 0}}));
        // butterfly twiddles in pass order
        c->n_pass_tw = (uint32_t)lg::pass_tw_total(c->logki);
        {
            const size_t cnt = c->n_pass_tw ? c->n_pass_tw : 1;
            std::vector<uint8_t> tf(cnt * 72), ti(cnt * 72);
            int logs = c->logki, logr = (c->logki < 3) ? c->logki : ((c->logki % 3) ? (c->logki % 3) : 3);
            while (logs > 0) {
                const int logsub = logs - logr;
                if (logsub > 0) {
                    const size_t off = (size_t)lg::pass_tw_offset(c->logki, logs);
                    // the inverse transform's 1/k rides on the first pass' twiddles (outer fold of 4: on the fold table instead)
                    const bool scaled = (logs == c->logki) && c->logo <= 1;
                    for (uint32_t m = 1; m < (1u << logr); m++)
                        for (uint32_t i0 = 0; i0 < (1u << logsub); i0++) {
                            const uint32_t e = (i0 * m) << (c->logki - logs);
                            fill_planes_q(tf, cnt, off + ((size_t)(m - 1) << logsub) + i0, pki[e]);
                            fill_planes_q(ti, cnt, off + ((size_t)(m - 1) << logsub) + i0, scaled ? mul(pki_inv[e], inv_k) : pki_inv[e]);
                        }
                }
                logs -= logr;
                logr = 3;
            }
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_tw_fwd), tf.size()));
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_tw_inv), ti.size()));
            LG_HIP(c, hipMemcpy(c->d_tw_fwd, tf.data(), tf.size(), hipMemcpyHostToDevice));
            LG_HIP(c, hipMemcpy(c->d_tw_inv, ti.data(), ti.size(), hipMemcpyHostToDevice));
        }
        // pre-scale table [plane][d] = wn^(s d mod n)
        {
            const size_t cnt = (size_t)c->nplanes * k;
            // O = 1: plain values + quotients (shoup29); O > 1: Montgomery operands of the fold's dot product
            std::vector<uint8_t> ct(cnt * (c->logo == 0 ? 72 : 36));
            for (uint32_t sp = 0; sp < c->nplanes; sp++)
                for (uint32_t d = 0; d < k; d++) {
                    // times 2^-256: the evaluation leaves the ABI's Montgomery form with its first product
                    const Fr w = from_mont(pn[((uint64_t)sp * d) & (n - 1)]);
                    if (c->logo == 0)
                        fill_planes_q(ct, cnt, (size_t)sp * k + d, w);
                    else
                        fill_planes(ct, cnt, (size_t)sp * k + d, to_f29(w));
                }
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_coset_tw), ct.size()));
            LG_HIP(c, hipMemcpy(c->d_coset_tw, ct.data(), ct.size(), hipMemcpyHostToDevice));
        }
        // dot-product coefficients of the radix-2 first pass (k = 2, 16, 128, 1024): [plane][4][i0 < k/2]
        if (c->logo == 0 && logk % 3 == 1 && logk > 1) {
            const uint32_t half = k / 2;
            const size_t cnt = (size_t)c->nplanes * 2 * k;
            std::vector<uint8_t> ft(cnt * 36);
            std::vector<Fr> pk(half);  // wk^i0
            Fr a = kOneMont;
            for (uint32_t i = 0; i < half; i++) { pk[i] = a; a = mul(a, wk); }
            for (uint32_t sp = 0; sp < c->nplanes; sp++)
                for (uint32_t i0 = 0; i0 < half; i0++) {
                    const Fr pre0 = pn[((uint64_t)sp * i0) & (n - 1)], pre1 = pn[((uint64_t)sp * (i0 + half)) & (n - 1)];
                    const Fr c10 = mul(pre0, pk[i0]), t = mul(pre1, pk[i0]);
                    const Fr zero = {{0, 0, 0, 0}};
                    const Fr c11 = (t.l[0] | t.l[1] | t.l[2] | t.l[3]) ? sub_raw(kP, t) : zero;
                    const size_t base = (size_t)sp * 4 * half + i0;
                    // all four carry 2^-256 (see the pre-scale table)
                    fill_planes(ft, cnt, base, to_f29(from_mont(pre0)));
                    fill_planes(ft, cnt, base + half, to_f29(from_mont(pre1)));
                    fill_planes(ft, cnt, base + 2 * (size_t)half, to_f29(from_mont(c10)));
                    fill_planes(ft, cnt, base + 3 * (size_t)half, to_f29(from_mont(c11)));
                }
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_first2), ft.size()));
            LG_HIP(c, hipMemcpy(c->d_first2, ft.data(), ft.size(), hipMemcpyHostToDevice));
        }
        // outer fold of the inverse transform.  Radix 2 (k = 8192) is a butterfly in the load stage: wk^-d, d < ki, for its
        // odd half.  Radix 4: dot-product factors [h][d] = wk^(-h d mod k) / k.
        if (c->logo == 1) {
            const size_t cnt = c->ki;
            std::vector<uint8_t> ft(cnt * 72);
            for (uint32_t d = 0; d < c->ki; d++) fill_planes_q(ft, cnt, d, pk_inv[d]);
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_fold_inv), ft.size()));
            LG_HIP(c, hipMemcpy(c->d_fold_inv, ft.data(), ft.size(), hipMemcpyHostToDevice));
        } else {
            const size_t cnt = (size_t)k << c->logo;
            std::vector<uint8_t> ft(cnt * 36);
            for (uint32_t h = 0; h < (1u << c->logo); h++)
                for (uint32_t d = 0; d < k; d++) fill_planes(ft, cnt, (size_t)h * k + d, to_f29(mul(pk_inv[((uint64_t)h * d) & (k - 1)], inv_k)));
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_fold_inv), ft.size()));
            LG_HIP(c, hipMemcpy(c->d_fold_inv, ft.data(), ft.size(), hipMemcpyHostToDevice));
        }
        const Fr w8 = domain_generator(3), w8i = inverse(w8);
        Fr p = w8, pi = w8i;
        for (int i = 0; i < 3; i++) {
            c->w8_fwd[i] = to_f29_plain(p);
            c->w8_inv[i] = to_f29_plain(pi);
            c->w8q_fwd[i] = to_f29_quot(p);
            c->w8q_inv[i] = to_f29_quot(pi);
            p = mul(p, w8);
            pi = mul(pi, w8i);
        }
        c->one29 = to_f29_plain(kOneMont);
        c->oneq29 = to_f29_quot(kOneMont);
        c->scale29 = to_f29(inv_k);
        c->invk29 = to_f29_plain(inv_k);
        c->invkq29 = to_f29_quot(inv_k);
        c->r2 = to_dev(kR2);
        c->r3 = to_dev(mul(kR2, kR2));  // R^2 (*) R^2 = R^4 / R = R^3
        return LG_OK;
    };
    rc = body();
    if (rc != LG_OK) {
        lg_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return LG_OK;
}

extern "C" {

int lg_ctx_create_batched(lg_ctx** out, int device, uint32_t rows, uint32_t k, uint32_t n, uint32_t batch) {
    return ctx_create_impl(out, device, rows, k, n, batch, nullptr);
}
int lg_ctx_create(lg_ctx** out, int device, uint32_t rows, uint32_t k, uint32_t n) {
    return ctx_create_impl(out, device, rows, k, n, 1, nullptr);
}
int lg_ctx_create_sharded(lg_ctx** out, int device, uint32_t rows, uint32_t k, uint32_t n, uint32_t plane_begin, uint32_t plane_count,
                          uint32_t coeff_rows_alloc) {
    const ShardSpec sp = {plane_begin, plane_count, coeff_rows_alloc ? coeff_rows_alloc : rows};
    return ctx_create_impl(out, device, rows, k, n, 1, &sp);
}
int lg_ctx_create_field(lg_ctx** out, int device, int field, uint32_t rows, uint32_t k, uint32_t n, uint32_t batch) {
    if (field == LG_FIELD_BN254_FR) return ctx_create_impl(out, device, rows, k, n, batch, nullptr);
    if (!out) return LG_ERR_BAD_ARG;
    *out = nullptr;
    if (field != LG_FIELD_BLS12_377_FQ && field != LG_FIELD_BN254_FR_GENERIC) return LG_ERR_BAD_ARG;
    const int logk = ilog2_exact(k), logn = ilog2_exact(n);
    if (rows == 0 || batch == 0 || logk < 1 || logn < 0 || n != 8 * (uint64_t)k) return LG_ERR_BAD_DIMS;
    if ((uint64_t)rows * batch > 0xffffffffull / 8) return LG_ERR_BAD_DIMS;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LG_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return LG_ERR_NO_DEVICE;
    lg_ctx* c = new (std::nothrow) lg_ctx();
    if (!c) return LG_ERR_OOM;
    c->device = device; c->rows = rows; c->k = k; c->n = n; c->batch = batch; c->logk = logk; c->logn = logn;
    c->total_rows = (uint64_t)rows * batch;
    c->ki = k; c->logki = logk; c->nplanes = 8; c->own_planes = 8; c->coeff_rows_alloc = (uint32_t)c->total_rows;
    auto body = [&]() -> int {
        LG_HIP(c, hipSetDevice(device));
        LG_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        return gf_create(&c->gf, field, rows, k, n, batch, c->stream, c->err, sizeof(c->err));
    };
    const int rc = body();
    if (rc != LG_OK) {
        lg_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return LG_OK;
}
uint32_t lg_ctx_element_words(const lg_ctx* c) { return c ? (c->gf ? gf_element_words64(c->gf) : 4u) : 0u; }
int lg_ctx_planes(const lg_ctx* c, uint32_t* nplanes, uint32_t* plane_begin, uint32_t* plane_count) {
    if (!c) return LG_ERR_BAD_ARG;
    if (nplanes) *nplanes = c->nplanes;
    if (plane_begin) *plane_begin = c->own_plane0;
    if (plane_count) *plane_count = c->own_planes;
    return LG_OK;
}

int lg_ctx_stream(lg_ctx* c, void** stream_out) {
    if (!c || !stream_out) return LG_ERR_BAD_ARG;
    *stream_out = static_cast<void*>(c->stream);
    return LG_OK;
}

int lg_ctx_dims(const lg_ctx* c, uint32_t* rows, uint32_t* k, uint32_t* n, uint32_t* batch) {
    if (!c) return LG_ERR_BAD_ARG;
    if (rows) *rows = c->rows;
    if (k) *k = c->k;
    if (n) *n = c->n;
    if (batch) *batch = c->batch;
    return LG_OK;
}

int lg_ctx_pipeline_chunks(const lg_ctx* c, uint32_t* chunks_out) {
    if (!c || !chunks_out) return LG_ERR_BAD_ARG;
    if (c->gf) { *chunks_out = 1; return LG_OK; }
    Chunk chunks[lg_ctx::kMaxChunks];
    *chunks_out = (uint32_t)plan_chunks(c, chunks);
    return LG_OK;
}

int lg_upload_preenc(lg_ctx* c, const uint64_t* preenc) {
    if (!c || !preenc) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); return gf_upload(c->gf, preenc); }
    if (c->sharded) return LG_ERR_STATE;   // a sharded context takes its row shard through lg_stage_interpolate
    LG_HIP(c, hipSetDevice(c->device));
    c->have_row0 = 0; c->have_row1 = c->rows;
    LG_HIP(c, hipMemcpyAsync(c->d_preenc, preenc, (size_t)c->total_rows * c->k * sizeof(fr), hipMemcpyHostToDevice, c->stream));
    return LG_OK;
}

int lg_profile_enable(lg_ctx* c, int on) {
    if (!c) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;   // generic-field contexts serve the hot path only
    LG_HIP(c, hipSetDevice(c->device));
    if (on && !c->ev_valid) {
        for (auto& set : c->ev)
            for (auto& e : set) LG_HIP(c, hipEventCreate(&e));
        c->ev_valid = true;
    }
    c->profiling = on != 0;
    c->prof_commits = 0;
    c->shard_commits = 0;
    return LG_OK;
}

// Single-chunk commits build their Merkle tree on the second stream and do NOT make the encode stream wait for
// it: the tree is latency bound (a few workgroups, ten dependent SHA-256 levels, 0.075 ms on the Poseidon batch)
// and the next commit's interpolation and evaluation do not touch the leaves, so in a stream of commits the tree
// hides behind them.  Everything that reads or rewrites leaves / nodes calls settle_tree() first.
static int settle_tree(lg_ctx* c) {
    if (c->tree_pending) {
        LG_HIP(c, hipStreamWaitEvent(c->stream, c->ev_tree, 0));
        c->tree_pending = false;
    }
    return LG_OK;
}

// the same for staged column hashes queued on the hash stream (lg_stage_hash_rows)
static int settle_hash(lg_ctx* c) {
    if (c->hash_pending) {
        LG_HIP(c, hipStreamWaitEvent(c->stream, c->ev_stage_hash, 0));
        c->hash_pending = false;
    }
    return LG_OK;
}

// The commit (mod.rs:521-551).  host_pre == nullptr: the matrix is resident in d_preenc.  Otherwise the
// rows are streamed from host memory chunk by chunk (same row range of every proof: one strided copy),
// so that the PCIe transfer of chunk c+1 overlaps the encoding of chunk c; host_coeffs (optional)
// receives the coefficient rows the same way in the other direction.
static int commit_core(lg_ctx* c, const uint64_t* host_pre, uint64_t* host_coeffs) {
    if (c->sharded) {
        snprintf(c->err, sizeof(c->err), "a sharded context holds planes [%u, %u) only: use the lg_stage_* calls", c->own_plane0, c->own_plane0 + c->own_planes);
        return LG_ERR_STATE;
    }
    if (!host_pre) {
        const int rc_ = need_all_message_rows(c, "lg_commit_resident");
        if (rc_ != LG_OK) return rc_;
    }
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    const uint64_t plane = c->total_rows * c->ki;
    const bool streamed = host_pre != nullptr;
    const bool prof = c->profiling && c->ev_valid && !streamed;
    hipEvent_t* ev = c->ev[c->prof_commits % lg_ctx::kProfRing];
    Chunk chunks[lg_ctx::kMaxChunks];
    const int nchunks = plan_chunks(c, chunks, streamed);
    // one chunk: the hash of THIS commit has nothing of its own to hide behind; it is put beside the NEXT commit's encoding
    // (async_hash) -- or, with that off, everything stays on the encode stream (no cross-stream waits)
    const bool async_hash = c->async_hash && c->async_tree && nchunks == 1;
    // (three deep: consecutive overlapped commits alternate between the two hash streams, so two column-hash chains run at once)
    hipStream_t hs = (nchunks > 1 || async_hash) ? ((async_hash && c->ring_depth == 3 && ((c->async_seq + 1) & 1)) ? c->stream_h2 : c->stream_h) : c->stream;
    if (async_hash) {
        // this commit encodes into the other U buffer; the hash that last read it (two commits ago) must be done.  The slot's three
        // buffers are allocated on first use, each checked on its own, and the context only moves to the slot once all three exist:
        // a failed allocation leaves the previous commitment and the ring position as they were
        const int par = (c->u_parity + 1) % c->ring_depth;
        if (!c->d_u_pp[par]) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_u_pp[par]), (size_t)c->nplanes * plane * sizeof(fr)));
        if (!c->d_leaves_pp[par]) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_leaves_pp[par]), (size_t)c->batch * c->n * 32));
        if (!c->d_nodes_pp[par]) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_nodes_pp[par]), (size_t)c->batch * (c->n - 1) * 32));
        c->u_parity = par;
        c->async_seq++;
        c->d_u = c->d_u_pp[par];
        LG_HIP(c, hipStreamWaitEvent(c->stream, c->ev_hash_free[par], 0));   // (never recorded = no-op)
        // ... and hashes into the other leaf buffer / builds the other tree (readers use c->d_leaves / c->d_nodes: this commitment's)
        // what an earlier commitment left queued against the buffers about to become "current" is covered below: the hash waits
        // for the tree that last read leaves[par]; read-backs of the previous commitment were issued on the encode stream
        c->d_leaves = c->d_leaves_pp[par];
        c->d_nodes = c->d_nodes_pp[par];
    }
    if (prof) LG_HIP(c, hipEventRecord(ev[0], c->stream));
    if (streamed) {
        // copies go on their own stream and the encode stream picks the chunks up by event.  Copy c+1 is
        // issued after the kernels of chunk c: a copy from pageable memory blocks the calling thread, and
        // this order lets the device work through chunk c meanwhile
        LG_HIP(c, hipEventRecord(c->ev_done, c->stream));            // earlier work on the encode stream may still read d_preenc
        LG_HIP(c, hipStreamWaitEvent(c->stream_up, c->ev_done, 0));
    } else {
        // rows -> coefficients (mod.rs:521-526) in one launch; also emits the canonical message = coset plane 0
        // (with an outer fold the canonical message planes 8c are written by the same kernel: every input is loaded by
        // the O workgroups of its row anyway, and a separate pass over the matrix cost 3.5 ms of S22's 94)
        lg::NttArgs a = interp_args(c, c->d_preenc, c->d_coeffs, c->d_u, 0, (uint32_t)c->total_rows);
        a.plane_stride = plane;
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->stream, a));
    }
    if (prof) LG_HIP(c, hipEventRecord(ev[1], c->stream));
    auto upload_chunk = [&](int i) -> int {
        const Chunk& ch = chunks[i];
        const size_t pitch = (size_t)c->rows * c->k * sizeof(fr);    // one proof
        const size_t off = (size_t)ch.row_begin * c->k * sizeof(fr), width = (size_t)(ch.row_end - ch.row_begin) * c->k * sizeof(fr);
        LG_HIP(c, hipMemcpy2DAsync(reinterpret_cast<uint8_t*>(c->d_preenc) + off, pitch, reinterpret_cast<const uint8_t*>(host_pre) + off, pitch,
                                   width, c->batch, hipMemcpyHostToDevice, c->stream_up));
        LG_HIP(c, hipEventRecord(c->ev_up[i], c->stream_up));
        return LG_OK;
    };
    if (streamed) {
        const int rc = upload_chunk(0);
        if (rc != LG_OK) return rc;
    }
    for (int i = 0; i < nchunks; i++) {
        const Chunk& ch = chunks[i];
        // coefficients -> cosets 1..7 of the order-n domain (mod.rs:528-533)
        const uint32_t row0 = ch.proof_begin * c->rows + ch.row_begin;
        const uint32_t nrows = ch.proof_count * (ch.row_end - ch.row_begin);
        if (streamed) {
            LG_HIP(c, hipStreamWaitEvent(c->stream, c->ev_up[i], 0));
            lg::NttArgs ia = interp_args(c, c->d_preenc, c->d_coeffs, c->d_u, row0, nrows);
            ia.chunk_rows = ch.row_end - ch.row_begin;
            ia.proof_stride = c->rows;
            ia.plane_stride = plane;
            LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->stream, ia));
            if (host_coeffs) LG_HIP(c, hipEventRecord(c->ev_coef[i], c->stream));
        }
        lg::NttArgs a = eval_args(c, c->d_coeffs, c->d_u, plane, row0, nrows, false);
        a.chunk_rows = ch.row_end - ch.row_begin;  // rows [row_begin, row_end) of each proof
        a.proof_stride = c->rows;
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, true, c->stream, a));
        if (prof && i + 1 == nchunks) LG_HIP(c, hipEventRecord(ev[2], c->stream));
        // column hashes (mod.rs:536-542) of the rows just encoded, on the hash stream
        if (nchunks > 1) {
            LG_HIP(c, hipEventRecord(c->ev_chunk[i], c->stream));
            LG_HIP(c, hipStreamWaitEvent(hs, c->ev_chunk[i], 0));
        }
        if (async_hash) {
            // the hash waits for this encoding, and for the tree (two commits ago, on the tree stream) that read the leaf buffer it is
            // about to rewrite; the previous commit's tree reads the OTHER buffer and runs beside this hash
            LG_HIP(c, hipEventRecord(c->ev_hashed, c->stream));
            LG_HIP(c, hipStreamWaitEvent(hs, c->ev_hashed, 0));
            LG_HIP(c, hipStreamWaitEvent(hs, c->ev_leaves_free[c->u_parity], 0));   // (never recorded = no-op)
            // (a tree that an earlier, differently scheduled commit left reading either buffer ran on stream_h -- stream order -- or on
            // the encode stream, before the event just waited for)
        } else if (i == 0) {   // the previous commit's tree may still be reading the leaves this hash is about to rewrite
            const int rc = settle_tree(c);
            if (rc != LG_OK) return rc;
            if (nchunks > 1) LG_HIP(c, hipStreamWaitEvent(hs, c->ev_tree, 0));   // (ev_tree: completed or never recorded = no-op)
        }
        if (prof && i == 0) LG_HIP(c, hipEventRecord(ev[3], hs));
        lg::ColHashArgs h;
        memset(&h, 0, sizeof(h));
        h.u = reinterpret_cast<const uint4*>(c->d_u);
        h.leaves = c->d_leaves;
        h.state = c->d_hstate;
        h.rows = c->rows; h.k = c->ki; h.lognp = (uint32_t)c->lognp;
        h.proof_begin = ch.proof_begin; h.proof_count = ch.proof_count;
        h.row_begin = ch.row_begin; h.row_end = ch.row_end;
        h.first = ch.row_begin == 0;
        h.last = ch.row_end == c->rows;
        h.plane_begin = 0; h.plane_count = c->nplanes;
        h.plane_stride = plane;
        h.col_pos = h.row_begin; h.col_rows = c->rows;
        const uint64_t threads = (uint64_t)ch.proof_count * c->n;
        if (h.first && h.last && threads <= c->quad_hash_max_columns) {
            // few columns (a single small proof): the one-lane-per-column kernel would be one latency chain per SIMD;
            // four lanes per column shorten the chain (hash_kernels.h)
            const lg::ColHashQuadArgs qa = quad_args_of(h);
            hipLaunchKernelGGL(lg::blake2s_columns_quad_kernel, dim3((uint32_t)((threads + 63) / 64)), dim3(256), 0, hs, qa);
        } else {
            hipLaunchKernelGGL(lg::blake2s_columns_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, hs, h);
        }
        LG_HIP(c, hipGetLastError());
        if (streamed && i + 1 < nchunks) {
            const int rc = upload_chunk(i + 1);
            if (rc != LG_OK) return rc;
        }
        if (streamed && host_coeffs) {
            // coefficient rows of this chunk go home while its cosets are being evaluated (issued after
            // the kernels for the same reason as the uploads: a copy to pageable memory blocks this thread)
            LG_HIP(c, hipStreamWaitEvent(c->stream_dn, c->ev_coef[i], 0));
            const size_t pitch = (size_t)c->rows * c->k * sizeof(fr);
            const size_t off = (size_t)ch.row_begin * c->k * sizeof(fr), width = (size_t)(ch.row_end - ch.row_begin) * c->k * sizeof(fr);
            LG_HIP(c, hipMemcpy2DAsync(reinterpret_cast<uint8_t*>(host_coeffs) + off, pitch, reinterpret_cast<const uint8_t*>(c->d_coeffs) + off, pitch,
                                       width, c->batch, hipMemcpyDeviceToHost, c->stream_dn));
        }
    }
    if (prof) LG_HIP(c, hipEventRecord(ev[4], hs));
    if (async_hash) LG_HIP(c, hipEventRecord(c->ev_hash_free[c->u_parity], hs));
    const bool async_tree = c->async_tree && nchunks == 1;
    // with the hash overlap on, the tree goes to its own stream (it follows this hash by event; the next commit's hash, on
    // stream_h, does not queue behind it)
    hipStream_t ms = async_hash ? c->stream_t : (async_tree ? c->stream_h : hs);
    if (async_hash) {
        LG_HIP(c, hipStreamWaitEvent(ms, c->ev_hash_free[c->u_parity], 0));   // recorded just above: this commit's hash is done
    } else if (async_tree) {
        LG_HIP(c, hipEventRecord(c->ev_hashed, c->stream));
        LG_HIP(c, hipStreamWaitEvent(c->stream_h, c->ev_hashed, 0));
    }
    // Merkle tree (mod.rs:544-551): nine levels per launch
    {
        lg::MerkleArgs m;
        m.leaves = c->d_leaves; m.nodes = c->d_nodes; m.n = c->n; m.logn = (uint32_t)c->logn; m.batch = c->batch;
        uint32_t depth = (uint32_t)c->logn;
        bool leaf = true;
        while (depth > 0) {
            m.in_depth = depth;
            m.chunks = depth > 9 ? (1u << (depth - 9)) : 1u;
            const dim3 grid(c->batch * m.chunks);
            if (leaf)
                LG_LAUNCH(c, lg::merkle_subtree_kernel<true>, grid, dim3(256), 0, ms, m);
            else
                LG_LAUNCH(c, lg::merkle_subtree_kernel<false>, grid, dim3(256), 0, ms, m);
            leaf = false;
            depth = depth > 9 ? depth - 9 : 0;
        }
        LG_HIP(c, hipGetLastError());
    }
    if (prof) {
        LG_HIP(c, hipEventRecord(ev[5], ms));
        c->prof_commits++;
    }
    if (async_hash) LG_HIP(c, hipEventRecord(c->ev_leaves_free[c->u_parity], ms));
    if (async_tree) {
        LG_HIP(c, hipEventRecord(c->ev_tree, ms));
        c->tree_pending = true;
    }
    // everything issued later on the encode stream (read-backs, the next commit) sees the tree
    if (nchunks > 1) {
        LG_HIP(c, hipEventRecord(c->ev_done, hs));
        LG_HIP(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));
    }
    c->committed = true;
    c->staging = false;
    c->have_planes = all_planes_mask(c);
    c->have_row0 = 0; c->have_row1 = c->rows;
    if (streamed && host_coeffs) LG_HIP(c, hipStreamSynchronize(c->stream_dn));
    return LG_OK;
}

// a commit that fails half way leaves no commitment behind (the buffers may be partly rewritten)
static int commit_checked(lg_ctx* c, const uint64_t* host_pre, uint64_t* host_coeffs) {
    const int rc = commit_core(c, host_pre, host_coeffs);
    if (rc != LG_OK && rc != LG_ERR_STATE) c->committed = false;
    return rc;
}

int lg_commit_resident(lg_ctx* c) {
    if (!c) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); return gf_commit(c->gf, nullptr, nullptr); }
    return commit_checked(c, nullptr, nullptr);
}

int lg_host_register(lg_ctx* c, void* ptr, size_t bytes) {
    if (!c || !ptr || bytes == 0) return LG_ERR_BAD_ARG;
    LG_HIP(c, hipSetDevice(c->device));
    LG_HIP(c, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return LG_OK;
}
int lg_host_unregister(lg_ctx* c, void* ptr) {
    if (!c || !ptr) return LG_ERR_BAD_ARG;
    LG_HIP(c, hipSetDevice(c->device));
    LG_HIP(c, hipHostUnregister(ptr));
    return LG_OK;
}

int lg_profile_read(lg_ctx* c, float ms_out[LG_STAGE_COUNT], uint32_t* samples_out) {
    if (!c || !ms_out) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;   // generic-field contexts serve the hot path only
    if (!c->ev_valid || !c->profiling || c->prof_commits == 0) return LG_ERR_STATE;
    LG_HIP(c, hipSetDevice(c->device));
    const uint64_t have = c->prof_commits < lg_ctx::kProfRing ? c->prof_commits : lg_ctx::kProfRing;
    double acc[LG_STAGE_COUNT] = {0, 0, 0, 0};
    static const int from[LG_STAGE_COUNT] = {0, 1, 3, 4}, to[LG_STAGE_COUNT] = {1, 2, 4, 5};
    for (uint64_t s = 0; s < have; s++) {
        hipEvent_t* ev = c->ev[(c->prof_commits - 1 - s) % lg_ctx::kProfRing];
        LG_HIP(c, hipEventSynchronize(ev[5]));
        LG_HIP(c, hipEventSynchronize(ev[2]));
        for (int i = 0; i < LG_STAGE_COUNT; i++) {
            float ms = 0;
            LG_HIP(c, hipEventElapsedTime(&ms, ev[from[i]], ev[to[i]]));
            acc[i] += ms;
        }
    }
    for (int i = 0; i < LG_STAGE_COUNT; i++) ms_out[i] = (float)(acc[i] / (double)have);
    if (samples_out) *samples_out = (uint32_t)have;
    return LG_OK;
}

int lg_sync(lg_ctx* c) {
    if (!c) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); return gf_sync(c->gf); }
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipStreamSynchronize(c->stream));
    return LG_OK;
}

static int read_back(lg_ctx* c, void* dst, const void* src, size_t bytes) {
    LG_HIP(c, hipSetDevice(c->device));
    LG_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    LG_HIP(c, hipStreamSynchronize(c->stream));
    return LG_OK;
}

int lg_read_root(lg_ctx* c, uint8_t* root_out) {
    if (!c || !root_out) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); return gf_committed(c->gf) ? gf_read_root(c->gf, root_out) : LG_ERR_STATE; }
    if (!c->committed) return LG_ERR_STATE;
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipMemcpy2DAsync(root_out, 32, c->d_nodes, (size_t)(c->n - 1) * 32, 32, c->batch, hipMemcpyDeviceToHost, c->stream));
    LG_HIP(c, hipStreamSynchronize(c->stream));
    return LG_OK;
}
int lg_read_coeffs(lg_ctx* c, uint64_t* out) {
    if (!c || !out) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); return gf_committed(c->gf) ? gf_read_coeffs(c->gf, out) : LG_ERR_STATE; }
    if (!c->committed) return LG_ERR_STATE;
    return read_back(c, out, c->d_coeffs, (size_t)c->total_rows * c->k * sizeof(fr));
}
int lg_read_leaves(lg_ctx* c, uint8_t* out) {
    if (!c || !out) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); return gf_committed(c->gf) ? gf_read_leaves(c->gf, out) : LG_ERR_STATE; }
    if (!c->committed) return LG_ERR_STATE;
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    return read_back(c, out, c->d_leaves, (size_t)c->batch * c->n * 32);
}
int lg_read_nodes(lg_ctx* c, uint8_t* out) {
    if (!c || !out) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); return gf_committed(c->gf) ? gf_read_nodes(c->gf, out) : LG_ERR_STATE; }
    if (!c->committed) return LG_ERR_STATE;
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    return read_back(c, out, c->d_nodes, (size_t)c->batch * (c->n - 1) * 32);
}

int lg_encode_commit(lg_ctx* c, const uint64_t* preenc, uint64_t* coeffs_out, uint8_t* root_out) {
    if (!c || !preenc || !root_out) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); const int rc_ = gf_commit(c->gf, preenc, coeffs_out); return rc_ != LG_OK ? rc_ : gf_read_root(c->gf, root_out); }
    // rows stream in (and coefficients out) while earlier rows are being encoded
    const int rc = commit_checked(c, preenc, coeffs_out);
    if (rc != LG_OK) return rc;
    return lg_read_root(c, root_out);
}

// ---- a1 on the device: the commit from the solution vector w alone (mod.rs:483-516 on the GPU) ---------------------------------
static int merkle_launches(lg_ctx* c, hipStream_t ms) {
    lg::MerkleArgs m;
    m.leaves = c->d_leaves; m.nodes = c->d_nodes; m.n = c->n; m.logn = (uint32_t)c->logn; m.batch = c->batch;
    uint32_t depth = (uint32_t)c->logn;
    bool leaf = true;
    while (depth > 0) {
        m.in_depth = depth;
        m.chunks = depth > 9 ? (1u << (depth - 9)) : 1u;
        const dim3 grid(c->batch * m.chunks);
        if (leaf)
            LG_LAUNCH(c, lg::merkle_subtree_kernel<true>, grid, dim3(256), 0, ms, m);
        else
            LG_LAUNCH(c, lg::merkle_subtree_kernel<false>, grid, dim3(256), 0, ms, m);
        leaf = false;
        depth = depth > 9 ? depth - 9 : 0;
    }
    return LG_OK;
}

int lg_upload_gate_map(lg_ctx* c, uint64_t npos, const uint32_t* left, const uint32_t* right, const uint64_t* constants, uint32_t nconst) {
    if (!c || (npos && (!left || !right)) || (nconst && !constants)) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    if (c->sharded || (c->rows & 3)) return LG_ERR_STATE;
    const uint64_t mk = (uint64_t)(c->rows / 4) * c->k;
    if (npos > mk || nconst >= lg::kGateConst) return LG_ERR_BAD_ARG;
    bool backward = true;
    for (uint64_t p = 0; p < npos; p++) {
        const uint32_t l = left[p], r = right[p];
        if ((l == lg::kGateNone) != (r == lg::kGateNone)) return LG_ERR_BAD_ARG;
        if (l == lg::kGateNone) continue;
        for (uint32_t s : {l, r}) {
            if (s & lg::kGateConst) { if ((s & ~lg::kGateConst) >= nconst) return LG_ERR_BAD_ARG; }
            else { if (s >= npos) return LG_ERR_BAD_ARG; if (s >= p) backward = false; }
        }
    }
    LG_HIP(c, hipSetDevice(c->device));
    LG_HIP(c, hipStreamSynchronize(c->stream));
    for (void* b : {(void*)c->d_gate_l, (void*)c->d_gate_r, (void*)c->d_gate_consts})
        if (b) LG_HIP(c, hipFree(b));
    c->d_gate_l = c->d_gate_r = nullptr; c->d_gate_consts = nullptr; c->gate_npos = 0;
    // positions past the solution vector (the zero padding up to m k, mod.rs:506-509) are no gates
    std::vector<uint32_t> pad(mk - npos, lg::kGateNone);
    LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_gate_l), mk * 4));
    LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_gate_r), mk * 4));
    LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_gate_consts), (nconst ? nconst : 1) * sizeof(fr)));
    if (npos) {
        LG_HIP(c, hipMemcpy(c->d_gate_l, left, npos * 4, hipMemcpyHostToDevice));
        LG_HIP(c, hipMemcpy(c->d_gate_r, right, npos * 4, hipMemcpyHostToDevice));
    }
    if (mk > npos) {
        LG_HIP(c, hipMemcpy(c->d_gate_l + npos, pad.data(), (mk - npos) * 4, hipMemcpyHostToDevice));
        LG_HIP(c, hipMemcpy(c->d_gate_r + npos, pad.data(), (mk - npos) * 4, hipMemcpyHostToDevice));
    }
    if (nconst) LG_HIP(c, hipMemcpy(c->d_gate_consts, constants, (size_t)nconst * sizeof(fr), hipMemcpyHostToDevice));
    c->gate_npos = npos; c->gate_nconst = nconst; c->gate_backward = backward;
    return LG_OK;
}

// x, y, z of positions [pos0, pos1) of every proof from the W block already in d_preenc
static int witness_gather(lg_ctx* c, uint64_t pos0, uint64_t pos1) {
    if (pos1 <= pos0) return LG_OK;
    lg::WitnessGatherArgs g;
    g.pre = c->d_preenc; g.left = c->d_gate_l; g.right = c->d_gate_r; g.consts = c->d_gate_consts;
    g.mk = (uint64_t)(c->rows / 4) * c->k; g.pos0 = pos0; g.pos1 = pos1; g.batch = c->batch;
    const uint64_t threads = (pos1 - pos0) * c->batch;
    LG_LAUNCH(c, lg::witness_gather_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->stream, g);
    return LG_OK;
}

// The commit (mod.rs:483-551) from w alone.  Only the W block crosses PCIe (a quarter of preenc_u); it travels in row steps,
// and as soon as step j is there the X, Y, Z rows of the same positions are gathered and the X and Y rows of the step are
// encoded (one launch for both blocks) -- with circuits whose gates refer backwards only, as compiled circuits do, a gate's
// operands have arrived with or before its own position -- so the transfer hides behind encoding.  A column's Blake2s absorbs
// the rows in order (X block first).  Large commits hash step by step on the hash stream, each launch held back until the NEXT
// step's evaluation starts: a short kernel (an interpolation) that runs beside a hash launch is stretched to the hash's length
// -- its workgroups on the CUs the hash occupies get what the older hash waves leave -- while the long evaluations absorb it
// (measured: rocprofv3 timeline).  Small commits (one chunk in plan_chunks' terms: both kernels issue bound,
// nothing to gain from running them side by side) hash once at the end on the encode stream.
static int commit_from_witness(lg_ctx* c, const uint64_t* host_w, uint64_t* host_coeffs) {
    if (c->sharded) return LG_ERR_STATE;
    if (!c->d_gate_l) {
        snprintf(c->err, sizeof(c->err), "lg_encode_commit_from_witness needs the circuit's gate map (lg_upload_gate_map)");
        return LG_ERR_STATE;
    }
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    const uint32_t m = c->rows / 4;
    const uint64_t mk = (uint64_t)m * c->k, plane = c->total_rows * c->ki;
    const size_t wbytes = (size_t)c->batch * mk * sizeof(fr);
    Chunk resident_plan[lg_ctx::kMaxChunks];
    const bool small = plan_chunks(c, resident_plan) == 1;
    // Steps of the W upload (row ranges of the W block = position ranges of all four blocks).  The first step's transfer has nothing
    // to hide behind and the last rows' hash has nothing left to hide it: both ends taper (weights 1, 2, 3, 3, 3 of the upload;
    // 4, 3, 2, 1 of the W block's rows).  LG_WITNESS_STEPS / LG_WITNESS_TAIL override the counts (experiments).
    static const int env_steps = [] { const char* e = getenv("LG_WITNESS_STEPS"); return e ? atoi(e) : 0; }();
    static const int env_tail = [] { const char* e = getenv("LG_WITNESS_TAIL"); return e ? atoi(e) : 0; }();
    uint32_t J = wbytes >= (size_t{64} << 20) ? 5 : (wbytes >= (size_t{8} << 20) ? 3 : 1);
    if (env_steps > 0) J = (uint32_t)std::min(env_steps, 5);
    if (!c->gate_backward) J = 1;                      // forward references: the whole of w first
    if (J > m) J = m;
    uint32_t CW = small ? 1 : 4, CZ = small ? 1 : 2;   // rows of the Z and W blocks: chunks as large commits are cut anyway
    if (env_tail > 0) CW = (uint32_t)std::min(env_tail, 4);
    if (CW > m) CW = m;
    if (CZ > m) CZ = m;
    auto cuts = [&](uint32_t parts, const uint32_t* weight) {   // row boundaries 0 = b[0] < ... < b[parts] = m by cumulative weight
        std::vector<uint32_t> bnd(parts + 1, 0);
        uint64_t total = 0, acc = 0;
        for (uint32_t i = 0; i < parts; i++) total += weight[i];
        for (uint32_t i = 0; i < parts; i++) {
            acc += weight[i];
            bnd[i + 1] = (i + 1 == parts) ? m : std::max<uint32_t>(bnd[i] + 1, (uint32_t)((uint64_t)m * acc / total));
            if (bnd[i + 1] > m) bnd[i + 1] = m;
        }
        return bnd;
    };
    static const uint32_t w_up[5][5] = {{1}, {1, 2}, {1, 2, 3}, {1, 2, 3, 3}, {1, 2, 3, 3, 3}};
    static const uint32_t w_tail[4][4] = {{1}, {2, 1}, {3, 2, 1}, {4, 3, 2, 1}};
    static const uint32_t w_even[2] = {1, 1};
    const std::vector<uint32_t> ub = cuts(J, w_up[J - 1]), zb = cuts(CZ, w_even), wb = cuts(CW, w_tail[CW - 1]);
    // encode steps: rows [r0, r1) of `blocks` consecutive blocks of every proof (blocks = 2: the X and the Y block) in one launch
    struct Step { uint32_t r0, r1, blocks; int upload; };
    std::vector<Step> enc;
    // (a small commit hashes at the end anyway, so nothing is gained by finishing the X block early: every step encodes its rows of
    // all four blocks -- the Z rows are gathered with the step, the W rows are the upload itself -- and the whole encoding overlaps
    // the transfer)
    for (uint32_t j = 0; j < J; j++) enc.push_back(Step{ub[j], ub[j + 1], small ? 4u : 2u, (int)j});
    if (!small) {
        for (uint32_t j = 0; j < CZ; j++) enc.push_back(Step{2 * m + zb[j], 2 * m + zb[j + 1], 1, -1});
        for (uint32_t j = 0; j < CW; j++) enc.push_back(Step{3 * m + wb[j], 3 * m + wb[j + 1], 1, -1});
    }
    // hash launches in row order: (rows, index of the encode step that completes them)
    struct HashStep { uint32_t r0, r1; size_t after; };
    std::vector<HashStep> hashes;
    if (small) {
        hashes.push_back(HashStep{0, c->rows, enc.size() - 1});
    } else {
        for (uint32_t j = 0; j < J; j++) hashes.push_back(HashStep{ub[j], ub[j + 1], j});
        hashes.push_back(HashStep{m, 2 * m, (size_t)J - 1});
        for (size_t i = J; i < enc.size(); i++) hashes.push_back(HashStep{enc[i].r0, enc[i].r1, i});
    }
    hipStream_t hs = small ? c->stream : c->stream_h;
    // earlier work on the encode stream may still read d_preenc; the previous commit's tree may still read the leaves
    LG_HIP(c, hipEventRecord(c->ev_done, c->stream));
    LG_HIP(c, hipStreamWaitEvent(c->stream_up, c->ev_done, 0));
    if (!small) LG_HIP(c, hipStreamWaitEvent(hs, c->ev_done, 0));
    auto upload = [&](uint32_t j) -> int {
        const uint32_t a = ub[j], b = ub[j + 1];
        const size_t width = (size_t)(b - a) * c->k * sizeof(fr);
        LG_HIP(c, hipMemcpy2DAsync(reinterpret_cast<uint8_t*>(c->d_preenc) + ((size_t)3 * m + a) * c->k * sizeof(fr), (size_t)c->rows * c->k * sizeof(fr),
                                   reinterpret_cast<const uint8_t*>(host_w) + (size_t)a * c->k * sizeof(fr), (size_t)mk * sizeof(fr), width, c->batch,
                                   hipMemcpyHostToDevice, c->stream_up));
        LG_HIP(c, hipEventRecord(c->ev_up[j], c->stream_up));
        return LG_OK;
    };
    auto hash_launch = [&](const HashStep& hr) -> int {
        lg::ColHashArgs h;
        memset(&h, 0, sizeof(h));
        h.u = reinterpret_cast<const uint4*>(c->d_u);
        h.leaves = c->d_leaves;
        h.state = c->d_hstate;
        h.rows = c->rows; h.k = c->ki; h.lognp = (uint32_t)c->lognp;
        h.proof_begin = 0; h.proof_count = c->batch;
        h.row_begin = hr.r0; h.row_end = hr.r1;
        h.first = hr.r0 == 0;
        h.last = hr.r1 == c->rows;
        h.plane_begin = 0; h.plane_count = c->nplanes;
        h.plane_stride = plane;
        h.col_pos = hr.r0; h.col_rows = c->rows;
        const uint64_t threads = (uint64_t)c->batch * c->n;
        if (h.first && h.last && threads <= c->quad_hash_max_columns) {   // few columns: four lanes per column (hash_kernels.h)
            const lg::ColHashQuadArgs qa = quad_args_of(h);
            LG_LAUNCH(c, lg::blake2s_columns_quad_kernel, dim3((uint32_t)((threads + 63) / 64)), dim3(256), 0, hs, qa);
        } else {
            LG_LAUNCH(c, lg::blake2s_columns_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, hs, h);
        }
        return LG_OK;
    };
    { const int rc_ = upload(0); if (rc_ != LG_OK) return rc_; }
    size_t next_hash = 0;
    for (size_t i = 0; i < enc.size(); i++) {
        const Step& st = enc[i];
        if (st.upload >= 0) {
            LG_HIP(c, hipStreamWaitEvent(c->stream, c->ev_up[st.upload], 0));
            const int rc_ = witness_gather(c, (uint64_t)st.r0 * c->k, (uint64_t)st.r1 * c->k);
            if (rc_ != LG_OK) return rc_;
        }
        const uint32_t span = st.r1 - st.r0, nrows = c->batch * st.blocks * span;
        lg::NttArgs ia = interp_args(c, c->d_preenc, c->d_coeffs, c->d_u, st.r0, nrows);
        ia.chunk_rows = span;
        ia.proof_stride = c->rows;
        ia.blk_count = st.blocks; ia.blk_stride = m;
        ia.plane_stride = plane;
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->stream, ia));
        if (host_coeffs) LG_HIP(c, hipEventRecord(c->ev_coef[i % lg_ctx::kMaxChunks], c->stream));
        if (!small && next_hash < hashes.size() && hashes[next_hash].after < i) {
            // the hashes of the rows complete by now start together with the evaluation below
            LG_HIP(c, hipEventRecord(c->ev_stage_in, c->stream));
            LG_HIP(c, hipStreamWaitEvent(hs, c->ev_stage_in, 0));
            while (next_hash < hashes.size() && hashes[next_hash].after < i) {
                const int rc_ = hash_launch(hashes[next_hash++]);
                if (rc_ != LG_OK) return rc_;
            }
        }
        lg::NttArgs a = eval_args(c, c->d_coeffs, c->d_u, plane, st.r0, nrows, false);
        a.chunk_rows = span;
        a.proof_stride = c->rows;
        a.blk_count = st.blocks; a.blk_stride = m;
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, true, c->stream, a));
        // the next step's rows start travelling (issued after this step's kernels: a copy from pageable memory blocks this thread)
        if (st.upload >= 0 && (uint32_t)st.upload + 1 < J) { const int rc_ = upload((uint32_t)st.upload + 1); if (rc_ != LG_OK) return rc_; }
        if (host_coeffs) {
            LG_HIP(c, hipStreamWaitEvent(c->stream_dn, c->ev_coef[i % lg_ctx::kMaxChunks], 0));
            const size_t pitch = (size_t)c->rows * c->k * sizeof(fr), width = (size_t)span * c->k * sizeof(fr);
            for (uint32_t blk = 0; blk < st.blocks; blk++) {
                const size_t off = ((size_t)st.r0 + (size_t)blk * m) * c->k * sizeof(fr);
                LG_HIP(c, hipMemcpy2DAsync(reinterpret_cast<uint8_t*>(host_coeffs) + off, pitch, reinterpret_cast<const uint8_t*>(c->d_coeffs) + off, pitch, width,
                                           c->batch, hipMemcpyDeviceToHost, c->stream_dn));
            }
        }
    }
    if (!small) {
        LG_HIP(c, hipEventRecord(c->ev_stage_in, c->stream));
        LG_HIP(c, hipStreamWaitEvent(hs, c->ev_stage_in, 0));
    }
    while (next_hash < hashes.size()) {
        const int rc_ = hash_launch(hashes[next_hash++]);
        if (rc_ != LG_OK) return rc_;
    }
    { const int rc_ = merkle_launches(c, hs); if (rc_ != LG_OK) return rc_; }
    if (!small) {
        LG_HIP(c, hipEventRecord(c->ev_done, hs));
        LG_HIP(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));
    }
    c->committed = true;
    c->staging = false;
    c->have_planes = all_planes_mask(c);
    c->have_row0 = 0; c->have_row1 = c->rows;
    if (host_coeffs) LG_HIP(c, hipStreamSynchronize(c->stream_dn));
    return LG_OK;
}

int lg_encode_commit_from_witness(lg_ctx* c, const uint64_t* w, uint64_t* coeffs_out, uint8_t* root_out) {
    if (!c || !w || !root_out) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    const int rc = commit_from_witness(c, w, coeffs_out);
    if (rc != LG_OK) {
        if (rc != LG_ERR_STATE) c->committed = false;
        return rc;
    }
    return lg_read_root(c, root_out);
}

int lg_read_codeword_rows(lg_ctx* c, uint32_t proof, uint32_t row0, uint32_t nrows, uint64_t* out) {
    if (!c || !out) return LG_ERR_BAD_ARG;
    if (c->gf) {
        if (!gf_committed(c->gf)) return LG_ERR_STATE;
        if (proof >= c->batch || (uint64_t)row0 + nrows > c->rows) return LG_ERR_BAD_ARG;
        LG_HIP(c, hipSetDevice(c->device));
        return gf_read_codeword_rows(c->gf, proof, row0, nrows, out);
    }
    if (!c->committed) return LG_ERR_STATE;
    if (proof >= c->batch || (uint64_t)row0 + nrows > c->rows) return LG_ERR_BAD_ARG;
    if (nrows == 0) return LG_OK;
    { const int rc_ = need_planes(c, all_planes_mask(c), "lg_read_codeword_rows"); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    const size_t elems = (size_t)nrows * c->n;
    int rc = grow(c, &c->d_scratch_c, &c->scratch_c_elems, elems);
    if (rc != LG_OK) return rc;
    const uint64_t threads = elems;
    hipLaunchKernelGGL(lg::planes_to_rows_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->stream, c->d_u,
                       c->total_rows * c->ki, (uint64_t)proof * c->rows + row0, nrows, c->ki, (uint32_t)c->lognp, c->r2, c->d_scratch_c);
    LG_HIP(c, hipGetLastError());
    return read_back(c, out, c->d_scratch_c, elems * sizeof(fr));
}

// opens t columns of each of `nproofs` consecutive proofs starting at `proof0` (one launch)
static int open_columns_impl(lg_ctx* c, uint32_t proof0, uint32_t nproofs, const uint32_t* idx, uint32_t t, uint64_t* cols_out, uint8_t* sib_out,
                             uint8_t* paths_out) {
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
    if (!c->committed) return LG_ERR_STATE;
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
    const uint32_t plen = (uint32_t)c->logn - 1;
    if (c->idx_cap < nidx) {
        if (c->d_idx) LG_HIP(c, hipFree(c->d_idx));
        c->d_idx = nullptr; c->idx_cap = 0;
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_idx), nidx * sizeof(uint32_t)));
        c->idx_cap = nidx;
    }
    const size_t path_bytes = nidx * (plen + 1) * 32;
    if (c->path_cap < path_bytes) {
        if (c->d_path_out) LG_HIP(c, hipFree(c->d_path_out));
        c->d_path_out = nullptr; c->path_cap = 0;
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_path_out), path_bytes));
        c->path_cap = path_bytes;
    }
    int rc = grow(c, &c->d_scratch_c, &c->scratch_c_elems, nidx * c->rows);
    if (rc != LG_OK) return rc;
    LG_HIP(c, hipMemcpyAsync(c->d_idx, idx, nidx * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    lg::GatherArgs g;
    memset(&g, 0, sizeof(g));
    g.u = c->d_u;
    g.leaves = c->d_leaves;
    g.nodes = c->d_nodes;
    g.idx = c->d_idx;
    g.cols = c->d_scratch_c;
    g.sib = c->d_path_out;
    g.paths = c->d_path_out + nidx * 32;
    g.r2 = c->r2;
    g.plane_stride = c->total_rows * c->ki;
    g.lognp = (uint32_t)c->lognp;
    g.proof0 = proof0;
    g.rows = c->rows; g.k = c->ki; g.n = c->n; g.logn = (uint32_t)c->logn; g.t = t;
    const uint64_t threads = (uint64_t)t * c->rows + (uint64_t)t * (plen + 1);
    hipLaunchKernelGGL(lg::gather_columns_kernel, dim3((uint32_t)((threads + 255) / 256), nproofs), dim3(256), 0, c->stream, g);
    LG_HIP(c, hipGetLastError());
    LG_HIP(c, hipMemcpyAsync(cols_out, c->d_scratch_c, nidx * c->rows * sizeof(fr), hipMemcpyDeviceToHost, c->stream));
    LG_HIP(c, hipMemcpyAsync(sib_out, g.sib, nidx * 32, hipMemcpyDeviceToHost, c->stream));
    if (plen) LG_HIP(c, hipMemcpyAsync(paths_out, g.paths, nidx * plen * 32, hipMemcpyDeviceToHost, c->stream));
    LG_HIP(c, hipStreamSynchronize(c->stream));
    return LG_OK;
}

int lg_open_columns(lg_ctx* c, uint32_t proof, const uint32_t* idx, uint32_t t, uint64_t* cols_out, uint8_t* sib_out, uint8_t* paths_out) {
    return open_columns_impl(c, proof, 1, idx, t, cols_out, sib_out, paths_out);
}

int lg_open_columns_batch(lg_ctx* c, const uint32_t* idx, uint32_t t, uint64_t* cols_out, uint8_t* sib_out, uint8_t* paths_out) {
    if (!c) return LG_ERR_BAD_ARG;
    return open_columns_impl(c, 0, c->batch, idx, t, cols_out, sib_out, paths_out);
}

// ---- sub-proof polynomials on the resident commitment -------------------------
// Every call serves all proofs of the batch in one set of launches (proof index = blockIdx.z).
static int sub_buffers(lg_ctx* c, size_t partial_elems, size_t r_elems) {
    int rc = grow(c, &c->d_sub_partial, &c->sub_partial_elems, partial_elems);
    if (rc != LG_OK) return rc;
    rc = grow(c, &c->d_sub_r, &c->sub_r_elems, r_elems);
    if (rc != LG_OK) return rc;
    if (!c->d_sub_q) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_sub_q), (size_t)c->batch * 2 * c->k * sizeof(fr)));
    return LG_OK;
}
static uint32_t sub_chunks(uint32_t rows, uint32_t* per_chunk) {
    uint32_t per = rows / 256;  // at most ~256 partial rows
    if (per < 32) per = 32;
    *per_chunk = per;
    return (rows + per - 1) / per;
}
static int sub_finish(lg_ctx* c, uint32_t nchunks, uint32_t cols, const fr& post, fr* out, uint32_t stride, uint32_t off, uint64_t out_proof) {
    hipLaunchKernelGGL(lg::rowsum_finish_kernel, dim3((cols + 255) / 256, 1, c->batch), dim3(256), 0, c->stream, c->d_sub_partial, nchunks, cols, post,
                       out, stride, off, out_proof);
    LG_HIP(c, hipGetLastError());
    return LG_OK;
}
// size-2k inverse NTT of the batch rows in d_sub_q (intermediate_domain of mod.rs:212), then copy out
static int sub_interpolate_2k(lg_ctx* c, uint64_t* coeffs_out) {
    if (c->logk + 1 > 14) return LG_ERR_UNSUPPORTED;
    if (!c->aux2k) {
        int rc = lg_ctx_create_batched(&c->aux2k, c->device, 1, 2 * c->k, 16 * c->k, c->batch);
        if (rc != LG_OK) return rc;
        LG_HIP(c, hipSetDevice(c->device));
    }
    lg_ctx* x = c->aux2k;
    lg::NttArgs a = interp_args(x, c->d_sub_q, x->d_coeffs, nullptr, 0, c->batch);
    LG_HIP(c, lg::launch_ntt(x->logki, x->logo, false, c->stream, a));
    return read_back(c, coeffs_out, x->d_coeffs, (size_t)c->batch * 2 * c->k * sizeof(fr));
}

int lg_interleaved_row_mul(lg_ctx* c, const uint64_t* r, uint64_t* out) {
    if (!c || !r || !out) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); return gf_interleaved_row_mul(c->gf, r, out); }
    { const int rc_ = need_all_message_rows(c, "lg_interleaved_row_mul"); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    uint32_t per;
    const uint32_t nch = sub_chunks(c->rows, &per);
    int rc = sub_buffers(c, (size_t)c->batch * nch * 2 * c->k, c->total_rows);
    if (rc != LG_OK) return rc;
    LG_HIP(c, hipMemcpyAsync(c->d_sub_r, r, (size_t)c->total_rows * sizeof(fr), hipMemcpyHostToDevice, c->stream));
    lg::RowSumArgs a;
    memset(&a, 0, sizeof(a));
    a.a = c->d_preenc; a.a_proof = (uint64_t)c->rows * c->k; a.a_row = c->k; a.a_col = 1;
    a.b = nullptr; a.r = c->d_sub_r;
    a.partial = c->d_sub_partial;
    a.rows = c->rows; a.cols = c->k; a.rows_per_chunk = per; a.nchunks = nch;
    hipLaunchKernelGGL(lg::rowsum_mul_kernel, dim3((c->k + 255) / 256, nch, c->batch), dim3(256), 0, c->stream, a);
    LG_HIP(c, hipGetLastError());
    // Montgomery x Montgomery -> Montgomery already: multiply by one (R) only to normalise
    rc = sub_finish(c, nch, c->k, to_dev(lg_host::kOneMont), c->d_sub_q, 1, 0, c->k);
    if (rc != LG_OK) return rc;
    return read_back(c, out, c->d_sub_q, (size_t)c->batch * c->k * sizeof(fr));
}

// buffers of the linear test: d_scratch_a = r_a rows | their coefficients, d_scratch_b = planes
static int linear_buffers(lg_ctx* c, uint32_t* per_out, uint32_t* nch_out) {
    const uint64_t R = c->total_rows;
    const size_t mat = (size_t)R * c->k;
    *nch_out = sub_chunks(c->rows, per_out);
    int rc = sub_buffers(c, (size_t)c->batch * *nch_out * 2 * c->k, 1);
    if (rc != LG_OK) return rc;
    rc = grow(c, &c->d_scratch_a, &c->scratch_a_elems, 2 * mat);
    if (rc != LG_OK) return rc;
    // the planes s = 4 (mod 8) of the r_a rows' encodings, slot s >> 3 (an eighth of a codeword matrix, not a whole one)
    return grow(c, &c->d_scratch_b, &c->scratch_b_elems, (size_t)((c->nplanes + 7) / 8) * R * c->ki);
}
// plane_mask: the planes s = 0 (mod 4) to serve (all of them, or the owned ones of a sharded context); points_out set =
// stop before the interpolation and hand back the 2k point values (slots of planes outside the mask are zero)
static int linear_core(lg_ctx* c, uint32_t per, uint32_t nch, uint32_t plane_mask, uint64_t* coeffs_out, uint64_t* points_out);
// the planes of the size-2k evaluation domain (s = 0 mod 4) a sub-proof call on this context serves
static uint32_t sub_plane_mask(const lg_ctx* c) { return (c->sharded ? own_planes_mask(c) : all_planes_mask(c)) & 0x11111111u; }
static int sub_points_begin(lg_ctx* c) {   // unserved slots of the point array read as zero
    if (!c->d_sub_q) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_sub_q), (size_t)c->batch * 2 * c->k * sizeof(fr)));
    LG_HIP(c, hipMemsetAsync(c->d_sub_q, 0, (size_t)c->batch * 2 * c->k * sizeof(fr), c->stream));
    return LG_OK;
}

int lg_linear_constraint_poly(lg_ctx* c, const uint64_t* r_a, uint64_t* coeffs_out) {
    if (!c || !r_a || !coeffs_out) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); return gf_linear_constraint_poly(c->gf, r_a, coeffs_out); }
    if (!c->committed) return LG_ERR_STATE;
    if (c->logk + 1 > 14) return LG_ERR_UNSUPPORTED;
    { const int rc_ = need_planes(c, all_planes_mask(c) & 0x11111111u, "lg_linear_constraint_poly"); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    uint32_t per, nch;
    int rc = linear_buffers(c, &per, &nch);
    if (rc != LG_OK) return rc;
    LG_HIP(c, hipMemcpyAsync(c->d_scratch_a, r_a, (size_t)c->total_rows * c->k * sizeof(fr), hipMemcpyHostToDevice, c->stream));
    return linear_core(c, per, nch, all_planes_mask(c) & 0x11111111u, coeffs_out, nullptr);
}

int lg_upload_constraint_matrix(lg_ctx* c, uint64_t num_rows, uint64_t nnz, const uint64_t* row_idx, const uint64_t* col_idx, const uint64_t* values) {
    if (!c || (nnz && (!row_idx || !col_idx || !values))) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;   // generic-field contexts serve the hot path only
    const uint64_t cols = (uint64_t)c->rows * c->k;   // 4 m k
    if (num_rows > 0xffffffffull || nnz > 0xffffffffull) return LG_ERR_UNSUPPORTED;
    for (uint64_t e = 0; e < nnz; e++)
        if (row_idx[e] >= num_rows || col_idx[e] >= cols) return LG_ERR_BAD_ARG;
    LG_HIP(c, hipSetDevice(c->device));
    // COO -> CSC (counting sort by column; duplicates stay duplicates, the product adds them up like row_mul does)
    std::vector<uint32_t> colptr(cols + 1, 0), erow(nnz);
    std::vector<fr> eval(nnz);
    for (uint64_t e = 0; e < nnz; e++) colptr[col_idx[e] + 1]++;
    for (uint64_t cc = 0; cc < cols; cc++) colptr[cc + 1] += colptr[cc];
    std::vector<uint32_t> fill(colptr.begin(), colptr.end() - 1);
    for (uint64_t e = 0; e < nnz; e++) {
        const uint32_t pos = fill[col_idx[e]]++;
        erow[pos] = (uint32_t)row_idx[e];
        memcpy(eval[pos].v, values + 4 * e, sizeof(fr));
    }
    std::vector<uint32_t> heavy;
    for (uint64_t cc = 0; cc < cols; cc++)
        if (colptr[cc + 1] - colptr[cc] > lg::kHeavyColumn) heavy.push_back((uint32_t)cc);
    // segments of the heavy columns: [seg_begin (nseg) | seg_end (nseg) | heavy_seg_ptr (nheavy + 1)]
    std::vector<uint32_t> seg_begin, seg_end, heavy_seg_ptr{0};
    for (uint32_t cc : heavy) {
        for (uint32_t e = colptr[cc]; e < colptr[cc + 1]; e += lg::kHeavySegment) {
            seg_begin.push_back(e);
            seg_end.push_back(std::min(colptr[cc + 1], e + lg::kHeavySegment));
        }
        heavy_seg_ptr.push_back((uint32_t)seg_begin.size());
    }
    for (void* b : {(void*)c->d_a_colptr, (void*)c->d_a_row, (void*)c->d_a_val, (void*)c->d_a_heavy, (void*)c->d_a_seg, (void*)c->d_a_seg_partial})
        if (b) LG_HIP(c, hipFree(b));
    c->d_a_colptr = nullptr; c->d_a_row = nullptr; c->d_a_val = nullptr; c->d_a_heavy = nullptr; c->d_a_seg = nullptr; c->d_a_seg_partial = nullptr;
    c->a_loaded = false;
    LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_a_heavy), (heavy.size() ? heavy.size() : 1) * 4));
    if (!heavy.empty()) LG_HIP(c, hipMemcpy(c->d_a_heavy, heavy.data(), heavy.size() * 4, hipMemcpyHostToDevice));
    c->a_nheavy = (uint32_t)heavy.size();
    c->a_nseg = (uint32_t)seg_begin.size();
    if (c->a_nseg) {
        std::vector<uint32_t> seg(seg_begin);
        seg.insert(seg.end(), seg_end.begin(), seg_end.end());
        seg.insert(seg.end(), heavy_seg_ptr.begin(), heavy_seg_ptr.end());
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_a_seg), seg.size() * 4));
        LG_HIP(c, hipMemcpy(c->d_a_seg, seg.data(), seg.size() * 4, hipMemcpyHostToDevice));
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_a_seg_partial), (size_t)c->batch * c->a_nseg * sizeof(fr)));
    }
    LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_a_colptr), colptr.size() * 4));
    LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_a_row), (nnz ? nnz : 1) * 4));
    LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_a_val), (nnz ? nnz : 1) * sizeof(fr)));
    LG_HIP(c, hipMemcpy(c->d_a_colptr, colptr.data(), colptr.size() * 4, hipMemcpyHostToDevice));
    if (nnz) {
        LG_HIP(c, hipMemcpy(c->d_a_row, erow.data(), nnz * 4, hipMemcpyHostToDevice));
        LG_HIP(c, hipMemcpy(c->d_a_val, eval.data(), nnz * sizeof(fr), hipMemcpyHostToDevice));
    }
    c->a_rows = num_rows; c->a_nnz = nnz; c->a_loaded = true;
    return LG_OK;
}

// r_linear (ChaCha20 + F::rand from the seeds) and r_a = A.row_mul(r_linear) into d_scratch_a (launches only; the caller
// checks the candidate-stream flag with linear_seed_flag once the stream has been synchronised)
static int linear_ra_from_seeds(lg_ctx* c, const uint8_t* seeds, uint32_t* per_out, uint32_t* nch_out) {
    uint32_t per, nch;
    int rc = linear_buffers(c, &per, &nch);
    if (rc != LG_OK) return rc;
    *per_out = per; *nch_out = nch;
    const uint64_t n = (uint64_t)c->rows * c->k;      // entries of r_a per proof = columns of A this context holds
    // entries of r_linear = rows of A.  The reference's A is square (4mk x 4mk); a context that holds only a row shard of the proof's
    // matrix (row relay, blocks layout) holds the matching COLUMNS of A and still needs every challenge
    const uint64_t rlen = c->a_rows;
    if (n > 0x7fffffffull || rlen > 0x7fffffffull) return LG_ERR_UNSUPPORTED;
    // 75.6 % of the 32-byte chunks are accepted; 1.5 chunks per element leaves > 50 standard deviations of margin
    const uint32_t blocks = (uint32_t)((rlen * 3 + 3) / 4 + 64), wgs = (blocks + 255) / 256;
    if (!c->d_seeds) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_seeds), (size_t)c->batch * 32));
    if (!c->d_short_flag) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_short_flag), 4));
    if (c->cc_counts_cap < (size_t)c->batch * wgs) {
        if (c->d_cc_counts) LG_HIP(c, hipFree(c->d_cc_counts));
        c->d_cc_counts = nullptr;
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_cc_counts), (size_t)c->batch * wgs * 4));
        c->cc_counts_cap = (size_t)c->batch * wgs;
    }
    rc = grow(c, &c->d_rlin, &c->rlin_elems, (size_t)c->batch * rlen);
    if (rc != LG_OK) return rc;
    LG_HIP(c, hipMemcpyAsync(c->d_seeds, seeds, (size_t)c->batch * 32, hipMemcpyHostToDevice, c->stream));
    LG_HIP(c, hipMemsetAsync(c->d_short_flag, 0, 4, c->stream));
    lg::ChaChaArgs a;
    a.seeds = c->d_seeds; a.out = c->d_rlin; a.counts = c->d_cc_counts; a.short_flag = c->d_short_flag;
    a.n = (uint32_t)rlen; a.blocks = blocks; a.wgs = wgs;
    LG_LAUNCH(c, lg::chacha_count_kernel, dim3(wgs, c->batch), dim3(256), 0, c->stream, a);
    LG_LAUNCH(c, lg::chacha_scatter_kernel, dim3(wgs, c->batch), dim3(256), 0, c->stream, a);
    lg::SparseRowMulArgs m;
    m.col_ptr = c->d_a_colptr; m.ent_row = c->d_a_row; m.ent_val = c->d_a_val;
    m.r = c->d_rlin; m.out = c->d_scratch_a; m.heavy = c->d_a_heavy; m.cols = (uint32_t)n; m.rows_in = (uint32_t)rlen;
    LG_LAUNCH(c, lg::sparse_row_mul_kernel, dim3((uint32_t)((n + 255) / 256), c->batch), dim3(256), 0, c->stream, m);
    if (c->a_nheavy) {
        lg::HeavySegArgs h;
        h.m = m;
        h.seg_begin = c->d_a_seg; h.seg_end = c->d_a_seg + c->a_nseg; h.heavy_seg_ptr = c->d_a_seg + 2 * (size_t)c->a_nseg;
        h.seg_partial = c->d_a_seg_partial; h.nseg = c->a_nseg;
        LG_LAUNCH(c, lg::sparse_row_mul_heavy_segments_kernel, dim3(c->a_nseg, c->batch), dim3(256), 0, c->stream, h);
        LG_LAUNCH(c, lg::sparse_row_mul_heavy_finish_kernel, dim3(c->a_nheavy, c->batch), dim3(256), 0, c->stream, h);
    }
    return LG_OK;
}
static int linear_seed_flag(lg_ctx* c) {
    uint32_t flag = 0;
    LG_HIP(c, hipMemcpy(&flag, c->d_short_flag, 4, hipMemcpyDeviceToHost));
    if (flag) {
        snprintf(c->err, sizeof(c->err), "ChaCha candidate stream too short for %llu elements", (unsigned long long)c->rows * c->k);
        return LG_ERR_STATE;
    }
    return LG_OK;
}
static int linear_from_seeds(lg_ctx* c, const uint8_t* seeds, uint32_t plane_mask, uint64_t* coeffs_out, uint64_t* points_out) {
    uint32_t per, nch;
    int rc = linear_ra_from_seeds(c, seeds, &per, &nch);
    if (rc != LG_OK) return rc;
    rc = linear_core(c, per, nch, plane_mask, coeffs_out, points_out);   // synchronises on the stream when it reads the result back
    if (rc != LG_OK) return rc;
    return linear_seed_flag(c);
}

// The VERIFIER's side of the linear test (mod.rs:748-830) on the device: r_linear from the seed, r_a = A.row_mul(r_linear), every
// r_a row interpolated and encoded on the large domain (mod.rs:773-781, 815-818), and for each opened column j the sum
// sum_i r_i(eta_j) * U[i][j] with the column the proof carries.  The encodings go where a commitment's codeword matrix lives,
// so a commitment this context held is void afterwards.
int lg_verifier_linear_sums_from_seed(lg_ctx* c, const uint8_t* seed, const uint32_t* idx, uint32_t t, const uint64_t* cols, uint64_t* sums_out) {
    if (!c || !seed || (t && (!idx || !cols || !sums_out))) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    if (c->batch != 1) return LG_ERR_UNSUPPORTED;
    if (c->sharded) {
        snprintf(c->err, sizeof(c->err), "lg_verifier_linear_sums_from_seed needs room for every coset plane; a sharded context holds [%u, %u)", c->own_plane0,
                 c->own_plane0 + c->own_planes);
        return LG_ERR_STATE;
    }
    if (!c->a_loaded) return LG_ERR_STATE;
    for (uint32_t i = 0; i < t; i++)
        if (idx[i] >= c->n) return LG_ERR_BAD_ARG;
    if (t == 0) return LG_OK;
    LG_HIP(c, hipSetDevice(c->device));
    // nothing of an earlier commit may still be reading or writing U, the leaves or the tree
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    for (hipStream_t st : {c->stream_h, c->stream_h2, c->stream_t})
        if (st) LG_HIP(c, hipStreamSynchronize(st));
    LG_HIP(c, hipStreamSynchronize(c->stream));
    c->committed = false; c->staging = false; c->have_planes = 0;
    uint32_t per, nch;
    int rc = linear_ra_from_seeds(c, seed, &per, &nch);
    if (rc != LG_OK) return rc;
    const uint64_t R = c->total_rows;
    const size_t mat = (size_t)R * c->k;
    const uint64_t plane = R * c->ki;
    fr* d_ra = c->d_scratch_a;
    fr* d_rc = c->d_scratch_a + mat;
    {
        lg::NttArgs a = interp_args(c, d_ra, d_rc, nullptr, 0, (uint32_t)R);
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->stream, a));
        lg::NttArgs e = eval_args(c, d_rc, c->d_u, plane, 0, (uint32_t)R, true);
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, true, c->stream, e));
    }
    // gathered[c][i] = r_i(eta_j) for the opened j (Montgomery), next to the proof's columns
    rc = grow(c, &c->d_scratch_c, &c->scratch_c_elems, 2 * (size_t)t * c->rows);
    if (rc != LG_OK) return rc;
    if (c->idx_cap < t) {
        if (c->d_idx) LG_HIP(c, hipFree(c->d_idx));
        c->d_idx = nullptr; c->idx_cap = 0;
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_idx), (size_t)t * sizeof(uint32_t)));
        c->idx_cap = t;
    }
    fr* d_gath = c->d_scratch_c;
    fr* d_cols = c->d_scratch_c + (size_t)t * c->rows;
    LG_HIP(c, hipMemcpyAsync(c->d_idx, idx, (size_t)t * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    LG_HIP(c, hipMemcpyAsync(d_cols, cols, (size_t)t * c->rows * sizeof(fr), hipMemcpyHostToDevice, c->stream));
    lg::GatherArgs g;
    memset(&g, 0, sizeof(g));
    g.u = c->d_u; g.leaves = c->d_leaves; g.nodes = c->d_nodes; g.idx = c->d_idx; g.cols = d_gath;
    // only the column threads are wanted, but the tail of the last workgroup falls into the kernel's path section: give it
    // real memory to write (what it writes -- pieces of a stale tree -- is never read)
    const uint32_t plen = (uint32_t)c->logn - 1;
    const size_t path_bytes = (size_t)t * (plen + 1) * 32;
    if (c->path_cap < path_bytes) {
        if (c->d_path_out) LG_HIP(c, hipFree(c->d_path_out));
        c->d_path_out = nullptr; c->path_cap = 0;
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_path_out), path_bytes));
        c->path_cap = path_bytes;
    }
    g.sib = c->d_path_out; g.paths = c->d_path_out + (size_t)t * 32;
    g.r2 = c->r2; g.plane_stride = plane; g.lognp = (uint32_t)c->lognp; g.proof0 = 0;
    g.rows = c->rows; g.k = c->ki; g.n = c->n; g.logn = (uint32_t)c->logn; g.t = t;
    const uint64_t threads = (uint64_t)t * c->rows;
    LG_LAUNCH(c, lg::gather_columns_kernel, dim3((uint32_t)((threads + 255) / 256), 1), dim3(256), 0, c->stream, g);
    // sums[c] = sum_i gathered[c][i] (*) cols[c][i]: "columns" of the row-sum kernel = the t openings, its rows = the 4m entries
    const uint32_t nchs = sub_chunks(c->rows, &per);
    rc = sub_buffers(c, std::max<size_t>((size_t)nchs * t, (size_t)nch * 2 * c->k), 1);
    if (rc != LG_OK) return rc;
    rc = grow(c, &c->d_sub_r, &c->sub_r_elems, t);       // the t sums
    if (rc != LG_OK) return rc;
    lg::RowSumArgs a;
    memset(&a, 0, sizeof(a));
    a.a = d_gath; a.a_row = 1; a.a_col = c->rows;
    a.b = d_cols; a.b_row = 1; a.b_col = c->rows;
    a.partial = c->d_sub_partial;
    a.rows = c->rows; a.cols = t; a.rows_per_chunk = per; a.nchunks = nchs;
    LG_LAUNCH(c, lg::rowsum_mul_kernel, dim3((t + 255) / 256, nchs, 1), dim3(256), 0, c->stream, a);
    // Montgomery x Montgomery -> Montgomery already: multiply by one (R) only to normalise
    LG_LAUNCH(c, lg::rowsum_finish_kernel, dim3((t + 255) / 256, 1, 1), dim3(256), 0, c->stream, c->d_sub_partial, nchs, t, to_dev(lg_host::kOneMont), c->d_sub_r,
              1u, 0u, (uint64_t)t);
    rc = read_back(c, sums_out, c->d_sub_r, (size_t)t * sizeof(fr));
    if (rc != LG_OK) return rc;
    return linear_seed_flag(c);
}

int lg_linear_constraint_poly_from_seeds(lg_ctx* c, const uint8_t* seeds, uint64_t* coeffs_out) {
    if (!c || !seeds || !coeffs_out) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;   // generic-field contexts serve the hot path only
    if (!c->committed || !c->a_loaded) return LG_ERR_STATE;
    if (c->logk + 1 > 14) return LG_ERR_UNSUPPORTED;
    { const int rc_ = need_planes(c, all_planes_mask(c) & 0x11111111u, "lg_linear_constraint_poly_from_seeds"); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    return linear_from_seeds(c, seeds, all_planes_mask(c) & 0x11111111u, coeffs_out, nullptr);
}

static int linear_core(lg_ctx* c, uint32_t per, uint32_t nch, uint32_t plane_mask, uint64_t* coeffs_out, uint64_t* points_out) {
    const uint32_t rows = c->rows, O = 1u << c->logo;
    const uint64_t R = c->total_rows;
    const size_t mat = (size_t)R * c->k;
    int rc = LG_OK;
    fr* d_ra = c->d_scratch_a;
    fr* d_rc = c->d_scratch_a + mat;
    // r_polys = small_domain.ifft(row) (mod.rs:726-729), then their values on the odd points of the
    // size-2k domain = planes s = 4 (mod 8) of their encoding
    const uint64_t plane = R * c->ki;
    {
        lg::NttArgs a = interp_args(c, d_ra, d_rc, nullptr, 0, (uint32_t)R);
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->stream, a));
        // one launch per computed plane, each into its own slot: the kernel addresses plane s at out + s * plane_stride
        for (uint32_t s = 4; s < c->nplanes; s += 8) {
            if (!(plane_mask & (1u << s))) continue;
            lg::NttArgs e = eval_args(c, d_rc, c->d_scratch_b + (uint64_t)(s >> 3) * plane - (uint64_t)s * plane, plane, 0, (uint32_t)R, true);
            e.ncos = 1;
            e.cosets[0] = (uint8_t)s;
            LG_HIP(c, lg::launch_ntt(c->logki, c->logo, true, c->stream, e));
        }
    }
    if (points_out) { rc = sub_points_begin(c); if (rc != LG_OK) return rc; }
    for (uint32_t s = 0; s < c->nplanes; s += 4) {
        if (!(plane_mask & (1u << s))) continue;
        lg::RowSumArgs a;
        memset(&a, 0, sizeof(a));
        a.a = c->d_u + (uint64_t)s * plane; a.a_proof = (uint64_t)rows * c->ki; a.a_row = c->ki; a.a_col = 1;   // u_i on this plane (canonical)
        if ((s & 7) == 0) {  // message plane 8c': r_i there = r_a[i][O j + c'] (Montgomery)
            a.b = d_ra + (s >> 3); a.b_proof = (uint64_t)rows * c->k; a.b_row = c->k; a.b_col = O;
        } else {             // computed plane (canonical)
            a.b = c->d_scratch_b + (uint64_t)(s >> 3) * plane; a.b_proof = (uint64_t)rows * c->ki; a.b_row = c->ki; a.b_col = 1;
        }
        a.partial = c->d_sub_partial;
        a.rows = rows; a.cols = c->ki; a.rows_per_chunk = per; a.nchunks = nch;
        hipLaunchKernelGGL(lg::rowsum_mul_kernel, dim3((c->ki + 255) / 256, nch, c->batch), dim3(256), 0, c->stream, a);
        LG_HIP(c, hipGetLastError());
        // canonical x Montgomery = plain -> x R^2; canonical x canonical = plain / R -> x R^3; point index j = (np/4) q + s/4
        rc = sub_finish(c, nch, c->ki, (s & 7) == 0 ? c->r2 : c->r3, c->d_sub_q, c->nplanes / 4, s / 4, 2 * (uint64_t)c->k);
        if (rc != LG_OK) return rc;
    }
    if (points_out) return read_back(c, points_out, c->d_sub_q, (size_t)c->batch * 2 * c->k * sizeof(fr));
    return sub_interpolate_2k(c, coeffs_out);
}

static int quadratic_core(lg_ctx* c, const uint64_t* r, uint32_t plane_mask, uint64_t* coeffs_out, uint64_t* points_out);
int lg_quadratic_constraint_poly(lg_ctx* c, const uint64_t* r, uint64_t* coeffs_out) {
    if (!c || !r || !coeffs_out) return LG_ERR_BAD_ARG;
    if (c->gf) { LG_HIP(c, hipSetDevice(c->device)); return gf_quadratic_constraint_poly(c->gf, r, coeffs_out); }
    if (!c->committed) return LG_ERR_STATE;
    if ((c->rows & 3) != 0) return LG_ERR_BAD_ARG;
    if (c->logk + 1 > 14) return LG_ERR_UNSUPPORTED;
    { const int rc_ = need_planes(c, all_planes_mask(c) & 0x11111111u, "lg_quadratic_constraint_poly"); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    return quadratic_core(c, r, all_planes_mask(c) & 0x11111111u, coeffs_out, nullptr);
}
static int quadratic_core(lg_ctx* c, const uint64_t* r, uint32_t plane_mask, uint64_t* coeffs_out, uint64_t* points_out) {
    const uint32_t m = c->rows / 4;
    uint32_t per;
    const uint32_t nch = sub_chunks(m, &per);
    int rc = sub_buffers(c, (size_t)c->batch * nch * 2 * c->k, (size_t)c->batch * m);
    if (rc != LG_OK) return rc;
    LG_HIP(c, hipMemcpyAsync(c->d_sub_r, r, (size_t)c->batch * m * sizeof(fr), hipMemcpyHostToDevice, c->stream));
    const uint64_t plane = c->total_rows * c->ki;
    if (points_out) { rc = sub_points_begin(c); if (rc != LG_OK) return rc; }
    for (uint32_t s = 0; s < c->nplanes; s += 4) {
        if (!(plane_mask & (1u << s))) continue;
        lg::QuadSumArgs a;
        memset(&a, 0, sizeof(a));
        a.u = c->d_u + (uint64_t)s * plane;
        a.r = c->d_sub_r;
        a.partial = c->d_sub_partial;
        a.r2 = c->r2;
        a.m = m; a.ki = c->ki; a.rows_per_chunk = per; a.nchunks = nch;
        hipLaunchKernelGGL(lg::quadsum_kernel, dim3((c->ki + 255) / 256, nch, c->batch), dim3(256), 0, c->stream, a);
        LG_HIP(c, hipGetLastError());
        rc = sub_finish(c, nch, c->ki, c->r2, c->d_sub_q, c->nplanes / 4, s / 4, 2 * (uint64_t)c->k);   // plain -> Montgomery
        if (rc != LG_OK) return rc;
    }
    if (points_out) return read_back(c, points_out, c->d_sub_q, (size_t)c->batch * 2 * c->k * sizeof(fr));
    return sub_interpolate_2k(c, coeffs_out);
}

// ---- the same three sums as POINT VALUES on the planes this context holds (coset-sharded commitments) --------
// The polynomials above are interpolated from their values at the size-2k domain, codeword indices 4 j, j = (np/4) q + s/4 for
// plane s = 0 (mod 4).  Every such value is a sum over ALL rows of data of ONE plane, so the rank owning the plane computes
// it alone; the host layer all-gathers the 2k-slot arrays (slot j belongs to plane 4 (j mod np/4)) and any rank interpolates.
// preenc_u.row_mul(r) is the same thing on the planes s = 0 (mod 8): message position p is codeword index 8 p, slot 2 p.
int lg_subproof_points(lg_ctx* c, int which, const void* challenge, uint64_t* points_out, uint32_t* plane_mask_out) {
    if (!c || !challenge || !points_out) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;   // generic-field contexts serve the hot path only
    if (c->batch != 1) return LG_ERR_UNSUPPORTED;
    if (!c->committed) return LG_ERR_STATE;
    if (c->logk + 1 > 14) return LG_ERR_UNSUPPORTED;
    uint32_t mask = sub_plane_mask(c);
    if (which == LG_SUB_INTERLEAVED) mask &= 0x01010101u;
    if (plane_mask_out) *plane_mask_out = mask;
    { const int rc_ = need_planes(c, mask, "lg_subproof_points"); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    int rc = LG_OK;
    switch (which) {
        case LG_SUB_INTERLEAVED: {
            uint32_t per;
            const uint32_t nch = sub_chunks(c->rows, &per);
            rc = sub_buffers(c, (size_t)nch * 2 * c->k, c->total_rows);
            if (rc != LG_OK) return rc;
            LG_HIP(c, hipMemcpyAsync(c->d_sub_r, challenge, (size_t)c->total_rows * sizeof(fr), hipMemcpyHostToDevice, c->stream));
            rc = sub_points_begin(c);
            if (rc != LG_OK) return rc;
            const uint64_t plane = c->total_rows * c->ki;
            for (uint32_t s = 0; s < c->nplanes; s += 8) {
                if (!(mask & (1u << s))) continue;
                lg::RowSumArgs a;
                memset(&a, 0, sizeof(a));
                a.a = c->d_u + (uint64_t)s * plane; a.a_proof = (uint64_t)c->rows * c->ki; a.a_row = c->ki; a.a_col = 1;   // canonical
                a.b = nullptr; a.r = c->d_sub_r;                                                                        // Montgomery
                a.partial = c->d_sub_partial;
                a.rows = c->rows; a.cols = c->ki; a.rows_per_chunk = per; a.nchunks = nch;
                LG_LAUNCH(c, lg::rowsum_mul_kernel, dim3((c->ki + 255) / 256, nch, 1), dim3(256), 0, c->stream, a);
                rc = sub_finish(c, nch, c->ki, c->r2, c->d_sub_q, c->nplanes / 4, s / 4, 2 * (uint64_t)c->k);   // plain -> Montgomery
                if (rc != LG_OK) return rc;
            }
            return read_back(c, points_out, c->d_sub_q, (size_t)2 * c->k * sizeof(fr));
        }
        case LG_SUB_LINEAR: {
            if (mask == 0) { memset(points_out, 0, (size_t)2 * c->k * sizeof(fr)); return LG_OK; }
            uint32_t per, nch;
            rc = linear_buffers(c, &per, &nch);
            if (rc != LG_OK) return rc;
            LG_HIP(c, hipMemcpyAsync(c->d_scratch_a, challenge, (size_t)c->total_rows * c->k * sizeof(fr), hipMemcpyHostToDevice, c->stream));
            return linear_core(c, per, nch, mask, nullptr, points_out);
        }
        case LG_SUB_LINEAR_FROM_SEED:
            if (mask == 0) { memset(points_out, 0, (size_t)2 * c->k * sizeof(fr)); return LG_OK; }   // no plane, no work (and no matrix needed)
            if (!c->a_loaded) return LG_ERR_STATE;
            return linear_from_seeds(c, static_cast<const uint8_t*>(challenge), mask, nullptr, points_out);
        case LG_SUB_QUADRATIC:
            if ((c->rows & 3) != 0) return LG_ERR_BAD_ARG;
            if (mask == 0) { memset(points_out, 0, (size_t)2 * c->k * sizeof(fr)); return LG_OK; }
            return quadratic_core(c, static_cast<const uint64_t*>(challenge), mask, nullptr, points_out);
        default: return LG_ERR_BAD_ARG;
    }
}

int lg_subproof_finish(lg_ctx* c, int which, const uint64_t* points, uint64_t* out) {
    if (!c || !points || !out) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    if (c->batch != 1) return LG_ERR_UNSUPPORTED;
    if (which == LG_SUB_INTERLEAVED) {   // the values at the message positions are the result
        for (uint32_t p = 0; p < c->k; p++) memcpy(out + 4 * (size_t)p, points + 4 * (size_t)(2 * p), sizeof(fr));
        return LG_OK;
    }
    if (which != LG_SUB_LINEAR && which != LG_SUB_LINEAR_FROM_SEED && which != LG_SUB_QUADRATIC) return LG_ERR_BAD_ARG;
    if (c->logk + 1 > 14) return LG_ERR_UNSUPPORTED;
    LG_HIP(c, hipSetDevice(c->device));
    if (!c->d_sub_q) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_sub_q), (size_t)2 * c->k * sizeof(fr)));
    LG_HIP(c, hipMemcpyAsync(c->d_sub_q, points, (size_t)2 * c->k * sizeof(fr), hipMemcpyHostToDevice, c->stream));
    return sub_interpolate_2k(c, out);
}

// ---- staged commit for one proof sharded over several GPUs ----------------
// Coset-sharded: rank g interpolates its row shard, the host layer all-gathers the coefficient rows (RCCL), then rank g
// evaluates and hashes the planes it owns for ALL rows, the host layer all-gathers the leaf digests, and every rank builds
// the (replicated) tree.  Row-relay: rank g keeps its rows end to end (all planes) and the columns' Blake2s states travel
// from rank to rank (lg_stage_hash_rows).
static uint32_t message_planes_mask(const lg_ctx* c) {   // planes s = 0 (mod 8): they hold the canonical message itself
    return all_planes_mask(c) & 0x01010101u;
}
static void add_canon_range(lg_ctx* c, uint32_t r0, uint32_t r1) {
    auto& v = c->canon_ranges;
    v.emplace_back(r0, r1);
    std::sort(v.begin(), v.end());
    size_t w = 0;
    for (size_t i = 1; i < v.size(); i++) {
        if (v[i].first <= v[w].second) v[w].second = std::max(v[w].second, v[i].second);
        else v[++w] = v[i];
    }
    v.resize(w + 1);
}

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
    if (c->sharded) {
        // the message rows of a sharded proof exist only shard by shard: hold exactly the range handed over
        const bool inside = c->d_preenc_alloc && row0 >= c->pre_row0 && (uint64_t)row0 + nrows <= (uint64_t)c->pre_row0 + c->pre_rows;
        if (!inside) {
            if (!preenc_rows) return LG_ERR_STATE;   // "already resident" rows that were never uploaded
            LG_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_preenc_alloc) LG_HIP(c, hipFree(c->d_preenc_alloc));
            c->d_preenc_alloc = nullptr; c->pre_rows = 0;
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_preenc_alloc), (size_t)nrows * c->k * sizeof(fr)));
            c->pre_row0 = row0; c->pre_rows = nrows;
            c->d_preenc = c->d_preenc_alloc - (size_t)row0 * c->k;   // virtual base: indexed by absolute row
        }
    }
    if (preenc_rows)
        LG_HIP(c, hipMemcpyAsync(c->d_preenc + (size_t)row0 * c->k, preenc_rows, (size_t)nrows * c->k * sizeof(fr), hipMemcpyHostToDevice, c->stream));
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
    LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->stream, a));
    if (!c->staging) c->canon_ranges.clear();
    if (msg_held) add_canon_range(c, row0, row0 + nrows);
    // a staged commit starts (or grows by an adjacent row range); what an earlier commitment left in U is void
    if (c->staging && row0 == c->have_row1) c->have_row1 = row0 + nrows;
    else if (c->staging && row0 + nrows == c->have_row0) c->have_row0 = row0;
    else if (!(c->staging && row0 >= c->have_row0 && row0 + nrows <= c->have_row1)) { c->have_row0 = row0; c->have_row1 = row0 + nrows; }
    c->staging = true;
    c->have_planes = 0;
    c->committed = false;
    return LG_OK;
}

static int stage_plane_args(lg_ctx* c, uint32_t plane_mask) {
    if (c->gf) return LG_ERR_UNSUPPORTED;   // generic-field contexts serve the hot path only
    if (c->batch != 1) return LG_ERR_STATE;
    if (c->nplanes < 32 && (plane_mask >> c->nplanes) != 0) return LG_ERR_BAD_ARG;
    if (plane_mask & ~own_planes_mask(c)) {
        snprintf(c->err, sizeof(c->err), "plane mask 0x%x reaches outside the planes [%u, %u) this sharded context holds", plane_mask, c->own_plane0,
                 c->own_plane0 + c->own_planes);
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
    LG_HIP(c, lg::launch_ntt(c->logki, c->logo, true, c->stream, a));
    return LG_OK;
}
static int stage_evaluate_range(lg_ctx* c, uint32_t plane_mask, uint32_t r0, uint32_t r1) {
    if (r1 <= r0 || plane_mask == 0) return LG_OK;
    const uint32_t nomsg = plane_mask & ~message_planes_mask(c);
    uint32_t at = r0;
    if (c->staging && nomsg != plane_mask)
        for (const auto& cr : c->canon_ranges) {
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
        const uint64_t threads = (uint64_t)h.plane_count * c->ki;
        // few columns (a rank's planes of a coset-sharded proof: n / G of them): the one-lane kernel would be a latency chain on a
        // fraction of the SIMDs; four lanes per column shorten it, now also across row ranges (the parked state is the same)
        if (threads <= c->quad_hash_max_columns && quad_can_take(h)) {
            const lg::ColHashQuadArgs qa = quad_args_of(h);
            LG_LAUNCH(c, lg::blake2s_columns_quad_kernel, dim3((uint32_t)((threads + 63) / 64)), dim3(256), 0, hs, qa);
        } else {
            LG_LAUNCH(c, lg::blake2s_columns_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, hs, h);
        }
        s = e2 + 1;
    }
    return LG_OK;
}

int lg_stage_evaluate_hash(lg_ctx* c, uint32_t plane_mask) {
    if (!c) return LG_ERR_BAD_ARG;
    { const int rc_ = stage_plane_args(c, plane_mask); if (rc_ != LG_OK) return rc_; }
    if (c->committed) { c->committed = false; c->have_planes = 0; }   // re-evaluating over a finished commitment voids it
    if (plane_mask == 0) return LG_OK;
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    // Row chunks, as in commit_core: while the encode stream evaluates chunk i + 1 of the owned planes, the hash stream absorbs
    // chunk i into the column states (S22 on one rank: 104 -> 92 ms; on 8 ranks each rank's share of it)
    Chunk chunks[lg_ctx::kMaxChunks];
    const int nchunks = plan_chunks(c, chunks);
    hipStream_t hs = nchunks > 1 ? c->stream_h : c->stream;
    if (nchunks > 1) {
        LG_HIP(c, hipEventRecord(c->ev_done, c->stream));     // earlier work on the encode stream (the previous tree) may read the leaves
        LG_HIP(c, hipStreamWaitEvent(hs, c->ev_done, 0));
    }
    for (int i = 0; i < nchunks; i++) {
        const Chunk& ch = chunks[i];
        { const int rc_ = stage_evaluate_range(c, plane_mask, ch.row_begin, ch.row_end); if (rc_ != LG_OK) return rc_; }
        if (nchunks > 1) {
            LG_HIP(c, hipEventRecord(c->ev_chunk[i], c->stream));
            LG_HIP(c, hipStreamWaitEvent(hs, c->ev_chunk[i], 0));
        }
        { const int rc_ = stage_hash_launch(c, hs, plane_mask, ch.row_begin, ch.row_end - ch.row_begin, ch.row_begin, c->rows); if (rc_ != LG_OK) return rc_; }
    }
    if (nchunks > 1) {   // later work on the encode stream (lg_stage_merkle, the caller's all-gather after lg_sync) sees the leaves
        LG_HIP(c, hipEventRecord(c->ev_done, hs));
        LG_HIP(c, hipStreamWaitEvent(c->stream, c->ev_done, 0));
    }
    c->have_planes |= plane_mask;
    return LG_OK;
}

// Split form of lg_stage_evaluate_hash for a caller that receives the coefficient rows piece by piece (an all-gather cut into pieces
// that arrive while earlier pieces are being evaluated): evaluate ANY rows that are there, in any order, then hash once all are done.
int lg_stage_evaluate_rows(lg_ctx* c, uint32_t plane_mask, uint32_t row0, uint32_t nrows) {
    if (!c) return LG_ERR_BAD_ARG;
    { const int rc_ = stage_plane_args(c, plane_mask); if (rc_ != LG_OK) return rc_; }
    if ((uint64_t)row0 + nrows > c->rows) return LG_ERR_BAD_ARG;
    if (c->committed) { c->committed = false; c->have_planes = 0; }   // re-evaluating over a finished commitment voids it
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
    if (c->committed) { c->committed = false; c->have_planes = 0; }
    if (plane_mask == 0) return LG_OK;
    LG_HIP(c, hipSetDevice(c->device));
    { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }   // the previous tree may still read the leaves
    LG_HIP(c, hipEventRecord(c->ev_stage_in, c->stream));               // the rows just evaluated, a state just received
    LG_HIP(c, hipStreamWaitEvent(c->stream_h, c->ev_stage_in, 0));
    { const int rc_ = stage_hash_launch(c, c->stream_h, plane_mask, row0, nrows, col_pos, col_rows); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipEventRecord(c->ev_stage_hash, c->stream_h));
    c->hash_pending = true;
    c->have_planes |= plane_mask;
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
            LG_LAUNCH(c, lg::merkle_subtree_kernel<true>, grid, dim3(256), 0, c->stream, m);
        else
            LG_LAUNCH(c, lg::merkle_subtree_kernel<false>, grid, dim3(256), 0, c->stream, m);
        leaf = false;
        depth = depth > 9 ? depth - 9 : 0;
    }
    LG_HIP(c, hipGetLastError());
    c->committed = true;
    c->staging = false;
    return LG_OK;
}

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
    if (!c->d_digest_xchg) LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_digest_xchg), (size_t)c->n * 32));
    LG_HIP(c, hipMemcpy2DAsync(c->d_digest_xchg + (size_t)rank * block, (size_t)per * 32, c->d_leaves + (size_t)rank * per * 32, (size_t)c->nplanes * 32,
                               (size_t)per * 32, c->ki, hipMemcpyDeviceToDevice, c->stream));
    *dptr_out = c->d_digest_xchg;
    *bytes_per_rank_out = block;
    return LG_OK;
}
int lg_stage_digests_unpack(lg_ctx* c, uint32_t world) {
    if (!c) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    if (c->batch != 1 || !c->d_digest_xchg) return LG_ERR_STATE;
    uint32_t per;
    { const int rc_ = digest_run(c, world, &per); if (rc_ != LG_OK) return rc_; }
    LG_HIP(c, hipSetDevice(c->device));
    const size_t block = (size_t)c->ki * per * 32;
    for (uint32_t o = 0; o < world; o++)
        LG_HIP(c, hipMemcpy2DAsync(c->d_leaves + (size_t)o * per * 32, (size_t)c->nplanes * 32, c->d_digest_xchg + (size_t)o * block, (size_t)per * 32,
                                   (size_t)per * 32, c->ki, hipMemcpyDeviceToDevice, c->stream));
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
            // unsharded: the caller takes the whole matrix over (a zero-copy producer), so every row counts as present again
            if (!c->sharded && !c->staging) { c->have_row0 = 0; c->have_row1 = c->rows; }
            *dptr_out = c->sharded ? c->d_preenc_alloc : c->d_preenc;
            *bytes_out = (size_t)(c->sharded ? c->pre_rows : c->total_rows) * c->k * sizeof(fr);
            break;
        case LG_BUF_COEFFS: *dptr_out = c->d_coeffs; *bytes_out = (size_t)c->coeff_rows_alloc * c->k * sizeof(fr); break;
        case LG_BUF_LEAVES: *dptr_out = c->d_leaves; *bytes_out = (size_t)c->batch * c->n * 32; break;
        case LG_BUF_NODES: *dptr_out = c->d_nodes; *bytes_out = (size_t)c->batch * (c->n - 1) * 32; break;
        case LG_BUF_HSTATE: *dptr_out = c->d_hstate; *bytes_out = (size_t)c->batch * c->n * LG_HSTATE_BYTES; break;
        default: return LG_ERR_BAD_ARG;
    }
    return LG_OK;
}

// ---- one call per commit for a proof sharded over several GPUs: the stages above in one stream-ordered sequence, the exchanges
// through the caller's callbacks (include/ligero_hip.h: lg_comm).  Nothing here waits on the host.
static int comm_fail(lg_ctx* c, const char* what, int rc) {
    snprintf(c->err, sizeof(c->err), "%s: the communication callback returned %d", what, rc);
    return LG_ERR_COMM;
}
static int shard_events(lg_ctx* c, hipEvent_t** ev_out) {
    *ev_out = nullptr;
    if (!c->profiling) return LG_OK;
    if (!c->ev_shard_valid) {
        for (auto& set : c->ev_shard)
            for (auto& e : set) LG_HIP(c, hipEventCreate(&e));
        c->ev_shard_valid = true;
    }
    *ev_out = c->ev_shard[c->shard_commits % lg_ctx::kShardProfRing];
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

int lg_commit_sharded(lg_ctx* c, const lg_comm* comm, const uint64_t* preenc_rows, uint32_t pieces) {
    if (!c || !comm) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    if (c->batch != 1) return LG_ERR_STATE;
    const uint32_t world = comm->world, rank = comm->rank;
    if (world == 0 || rank >= world) return LG_ERR_BAD_ARG;
    const bool exchange = world > 1 || (comm->flags & LG_COMM_EXCHANGE_AT_WORLD_1);
    if (exchange && !comm->all_gather) return LG_ERR_BAD_ARG;
    if (c->nplanes % world != 0 || c->own_planes != c->nplanes / world || c->own_plane0 != rank * (c->nplanes / world)) {
        snprintf(c->err, sizeof(c->err), "lg_commit_sharded: rank %u of %u must hold planes [%u, %u) of %u (lg_ctx_create_sharded); this context holds [%u, %u)", rank, world,
                 rank * (c->nplanes / world), (rank + 1) * (c->nplanes / world), c->nplanes, c->own_plane0, c->own_plane0 + c->own_planes);
        return LG_ERR_STATE;
    }
    LG_HIP(c, hipSetDevice(c->device));
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
    if ((uint64_t)np * piece_rows > c->coeff_rows_alloc) {
        LG_HIP(c, hipStreamSynchronize(c->stream));
        if (c->stream_x) LG_HIP(c, hipStreamSynchronize(c->stream_x));
        LG_HIP(c, hipFree(c->d_coeffs));
        c->d_coeffs = nullptr; c->coeff_rows_alloc = 0;
        LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_coeffs), (size_t)np * piece_rows * c->k * sizeof(fr)));
        c->coeff_rows_alloc = np * piece_rows;
    }
    // this rank's message rows, compact and piece-major
    const bool same_layout = c->shard_pieces == np && c->shard_world == world && c->shard_rank == rank && c->compact_rows == own;
    if (!preenc_rows && own && !(same_layout && (c->sharded ? c->d_preenc_alloc != nullptr : true))) {
        snprintf(c->err, sizeof(c->err), "lg_commit_sharded: no resident rows of this layout (pass this rank's %u rows)", own);
        return LG_ERR_STATE;
    }
    fr* compact = nullptr;
    if (c->sharded) {
        if (own && (!c->d_preenc_alloc || !same_layout)) {
            LG_HIP(c, hipStreamSynchronize(c->stream));
            if (c->d_preenc_alloc) LG_HIP(c, hipFree(c->d_preenc_alloc));
            c->d_preenc_alloc = nullptr; c->pre_rows = 0;
            LG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_preenc_alloc), (size_t)own * c->k * sizeof(fr)));
        }
        compact = c->d_preenc_alloc;
        // (ranges that follow one another -- a single range, or one rank owning every sub-block -- keep lg_stage_interpolate's "rows
        // [pre_row0, pre_row0 + pre_rows) are resident" view of the same buffer: the compact order is then the matrix order)
        bool one_run = nranges >= 1;
        for (uint32_t i = 1; i < nranges; i++) one_run = one_run && ranges[2 * i] == ranges[2 * (i - 1)] + ranges[2 * (i - 1) + 1];
        c->pre_row0 = one_run ? ranges[0] : 0;
        c->pre_rows = one_run ? own : 0;
        c->d_preenc = one_run ? c->d_preenc_alloc - (size_t)ranges[0] * c->k : nullptr;
    } else {
        // an unsharded context (world 1): the matrix has its own full-size buffer; the rows sit at their own positions
        compact = c->d_preenc;
    }
    c->shard_pieces = np; c->shard_world = world; c->shard_rank = rank; c->compact_rows = own;
    if (preenc_rows && own) LG_HIP(c, hipMemcpyAsync(compact, preenc_rows, (size_t)own * c->k * sizeof(fr), hipMemcpyHostToDevice, c->stream));
    hipEvent_t* ev = nullptr;
    { const int rc_ = shard_events(c, &ev); if (rc_ != LG_OK) return rc_; }
    if (ev) LG_HIP(c, hipEventRecord(ev[0], c->stream));
    if (exchange && np > 1 && !c->stream_x) LG_HIP(c, hipStreamCreateWithFlags(&c->stream_x, hipStreamNonBlocking));
    // a new staged commit: what an earlier commitment left in U is void
    c->staging = true; c->committed = false; c->have_planes = 0;
    c->canon_ranges.clear();
    // the message rows this context holds as ONE run of the matrix: all of its ranges if they follow one another (one rank: the
    // whole matrix), else none -- the entry points that read preenc_u rows (lg_interleaved_row_mul, lg_commit_resident) index them
    // by matrix row, which a compact buffer of scattered ranges does not support
    bool rows_one_run = nranges >= 1;
    for (uint32_t i = 1; i < nranges; i++) rows_one_run = rows_one_run && ranges[2 * i] == ranges[2 * (i - 1)] + ranges[2 * (i - 1) + 1];
    c->have_row0 = rows_one_run ? ranges[0] : 0; c->have_row1 = rows_one_run ? ranges[0] + own : 0;
    const uint32_t msg_held = message_planes_mask(c) & own_planes_mask(c);
    const uint64_t plane = c->total_rows * c->ki;
    const uint32_t mask = own_planes_mask(c);
    // 1. interpolate this rank's rows piece by piece; piece p of the all-gather follows on the exchange stream
    uint32_t compact_off = 0;
    for (uint32_t i = 0; i < nranges; i++) {
        const uint32_t r0 = ranges[2 * i], nr = ranges[2 * i + 1], p = r0 / piece_rows;
        // `in` is indexed by the absolute row like `out`: bias the compact buffer's base accordingly
        const fr* in = c->sharded ? compact + ((int64_t)compact_off - (int64_t)r0) * (int64_t)c->k : compact;
        lg::NttArgs a = interp_args(c, in, c->d_coeffs, msg_held ? c->d_u : nullptr, r0, nr);
        if (msg_held) {
            a.plane_stride = plane;
            a.canon_mask = 0;
            for (uint32_t cc = 0; cc < (1u << c->logo); cc++)
                if (msg_held & (1u << (8 * cc))) a.canon_mask |= 1u << cc;
            add_canon_range(c, r0, r0 + nr);
        }
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->stream, a));
        compact_off += nr;
        (void)p;
    }
    if (ev) LG_HIP(c, hipEventRecord(ev[1], c->stream));
    // 2. + 3. the pieces: all-gather (in place on whole pieces of LG_BUF_COEFFS), evaluate, hash -- piece p + 1 is on the wire while
    // piece p is evaluated, the hash of piece p runs on the hash stream beside the evaluation of piece p + 1
    hipStream_t xs = (exchange && np > 1) ? c->stream_x : c->stream;
    if (exchange) {
        if (xs != c->stream) {
            LG_HIP(c, hipEventRecord(c->ev_done, c->stream));            // every own row is interpolated
            LG_HIP(c, hipStreamWaitEvent(xs, c->ev_done, 0));
        }
        for (uint32_t p = 0; p < np; p++) {
            uint8_t* base = reinterpret_cast<uint8_t*>(c->d_coeffs) + (size_t)p * piece_rows * c->k * sizeof(fr);
            const int rc_ = comm->all_gather(comm->user, base, (uint64_t)sub * c->k * sizeof(fr), static_cast<void*>(xs));
            if (rc_ != 0) return comm_fail(c, "all-gather of the coefficient rows", rc_);
            if (xs != c->stream) LG_HIP(c, hipEventRecord(c->ev_up[p], xs));
        }
    }
    if (ev) LG_HIP(c, hipEventRecord(ev[2], c->stream));
    for (uint32_t p = 0; p < np; p++) {
        // (the time this stream stands still waiting for piece p is measured by an event on either side of the wait)
        if (ev) LG_HIP(c, hipEventRecord(ev[lg_ctx::kShardStages + 1 + 2 * p], c->stream));
        if (exchange && xs != c->stream) LG_HIP(c, hipStreamWaitEvent(c->stream, c->ev_up[p], 0));
        if (ev) LG_HIP(c, hipEventRecord(ev[lg_ctx::kShardStages + 2 + 2 * p], c->stream));
        const uint32_t r0 = p * piece_rows, r1 = std::min(c->rows, (p + 1) * piece_rows);
        // (a single piece of a large commit is still cut into row chunks, as lg_stage_evaluate_hash does)
        Chunk chunks[lg_ctx::kMaxChunks];
        int nchunks = np > 1 ? 1 : plan_chunks(c, chunks);
        if (np > 1) chunks[0] = Chunk{0, 1, r0, r1};
        for (int i = 0; i < nchunks; i++) {
            { const int rc_ = stage_evaluate_range(c, mask, chunks[i].row_begin, chunks[i].row_end); if (rc_ != LG_OK) return rc_; }
            LG_HIP(c, hipEventRecord(c->ev_stage_in, c->stream));
            LG_HIP(c, hipStreamWaitEvent(c->stream_h, c->ev_stage_in, 0));
            { const int rc_ = stage_hash_launch(c, c->stream_h, mask, chunks[i].row_begin, chunks[i].row_end - chunks[i].row_begin, chunks[i].row_begin, c->rows); if (rc_ != LG_OK) return rc_; }
            LG_HIP(c, hipEventRecord(c->ev_stage_hash, c->stream_h));
            c->hash_pending = true;
        }
    }
    c->have_planes |= mask;
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    if (ev) LG_HIP(c, hipEventRecord(ev[3], c->stream));
    // 4. the digests
    if (exchange) {
        void* d = nullptr; size_t bytes = 0;
        { const int rc_ = lg_stage_digests_pack(c, world, rank, &d, &bytes); if (rc_ != LG_OK) return rc_; }
        const int rc_ = comm->all_gather(comm->user, d, (uint64_t)bytes, static_cast<void*>(c->stream));
        if (rc_ != 0) return comm_fail(c, "all-gather of the leaf digests", rc_);
        { const int rc2 = lg_stage_digests_unpack(c, world); if (rc2 != LG_OK) return rc2; }
    }
    if (ev) LG_HIP(c, hipEventRecord(ev[4], c->stream));
    // 5. the tree
    { const int rc_ = lg_stage_merkle(c); if (rc_ != LG_OK) return rc_; }
    if (ev) { LG_HIP(c, hipEventRecord(ev[5], c->stream)); c->shard_wait_pairs[c->shard_commits % lg_ctx::kShardProfRing] = np; c->shard_commits++; }
    return LG_OK;
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
    } else {
        return LG_ERR_BAD_ARG;
    }
    *nranges_out = n;
    return LG_OK;
}

int lg_commit_row_relay(lg_ctx* c, const lg_comm* comm, uint64_t col_rows, int layout, uint32_t plane_groups, const uint64_t* preenc_rows) {
    if (!c || !comm) return LG_ERR_BAD_ARG;
    if (c->gf) return LG_ERR_UNSUPPORTED;
    if (c->batch != 1 || c->sharded) return LG_ERR_STATE;
    const uint32_t world = comm->world, rank = comm->rank;
    if (world == 0 || rank >= world) return LG_ERR_BAD_ARG;
    const bool exchange = world > 1;
    if (exchange && (!comm->send || !comm->recv || !comm->broadcast)) return LG_ERR_BAD_ARG;
    // the chain: every range of every rank, in column order
    struct Link { uint64_t pos, n; uint32_t owner, local; };
    std::vector<Link> chain;
    uint32_t local_rows = 0;
    for (uint32_t r = 0; r < world; r++) {
        uint64_t rg[8]; uint32_t nr = 0;
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
    hipEvent_t* ev = nullptr;
    { const int rc_ = shard_events(c, &ev); if (rc_ != LG_OK) return rc_; }
    if (ev) LG_HIP(c, hipEventRecord(ev[0], c->stream));
    const uint32_t all = all_planes_mask(c);
    bool head_done = false;
    if (local_rows) {
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
        if (c->committed) { c->committed = false; c->have_planes = 0; }
    }
    if (ev) LG_HIP(c, hipEventRecord(ev[1], c->stream));
    if (ev) LG_HIP(c, hipEventRecord(ev[2], c->stream));
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
    for (size_t i = 0; i < chain.size(); i++) {
        const Link& l = chain[i];
        if (l.owner != rank) continue;
        for (uint32_t g = 0; g < P; g++) {
            uint8_t* gstate = reinterpret_cast<uint8_t*>(c->d_hstate) + (size_t)g * group_bytes;
            const uint32_t gmask = (per >= 32 ? 0xffffffffu : ((1u << per) - 1u)) << (g * per);
            if (exchange && i > 0 && chain[i - 1].owner != rank) {
                { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
                const int rc_ = comm->recv(comm->user, gstate, group_bytes, chain[i - 1].owner, static_cast<void*>(c->stream));
                if (rc_ != 0) return comm_fail(c, "receive of the column states", rc_);
            }
            if (!(head_done && i == 0)) { const int rc_ = lg_stage_hash_rows(c, gmask, l.local, (uint32_t)l.n, l.pos, col_rows); if (rc_ != LG_OK) return rc_; }
            if (exchange && i + 1 < chain.size() && chain[i + 1].owner != rank) {
                { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
                const int rc_ = comm->send(comm->user, gstate, group_bytes, chain[i + 1].owner, static_cast<void*>(c->stream));
                if (rc_ != 0) return comm_fail(c, "send of the column states", rc_);
            }
        }
    }
    { const int rc_ = settle_hash(c); if (rc_ != LG_OK) return rc_; }
    if (ev) LG_HIP(c, hipEventRecord(ev[3], c->stream));
    if (exchange || ((comm->flags & LG_COMM_EXCHANGE_AT_WORLD_1) && comm->broadcast)) {
        { const int rc_ = settle_tree(c); if (rc_ != LG_OK) return rc_; }
        const int rc_ = comm->broadcast(comm->user, c->d_leaves, (uint64_t)c->n * 32, exchange ? chain.back().owner : 0, static_cast<void*>(c->stream));
        if (rc_ != 0) return comm_fail(c, "broadcast of the leaf digests", rc_);
    }
    if (ev) LG_HIP(c, hipEventRecord(ev[4], c->stream));
    { const int rc_ = lg_stage_merkle(c); if (rc_ != LG_OK) return rc_; }
    c->have_planes = all;    // every plane of this rank's rows is here (a rank without rows holds the tree only)
    if (ev) { LG_HIP(c, hipEventRecord(ev[5], c->stream)); c->shard_wait_pairs[c->shard_commits % lg_ctx::kShardProfRing] = 0; c->shard_commits++; }
    return LG_OK;
}

// mean milliseconds per stage of the sharded commits since lg_profile_enable(ctx, 1) (at most the last 16): coset mode
// {interpolate, wait for the last piece of the coefficient all-gather, evaluate + hash, digest all-gather, tree}; row relay
// {encode (+ the head's overlapped hash), 0, the relay (waiting for the previous rank, own hash, hand-over), digest broadcast, tree}
int lg_shard_profile_read(lg_ctx* c, float ms_out[5], uint32_t* samples_out) {
    if (!c || !ms_out) return LG_ERR_BAD_ARG;
    if (!c->ev_shard_valid || !c->profiling || c->shard_commits == 0) return LG_ERR_STATE;
    LG_HIP(c, hipSetDevice(c->device));
    const uint64_t have = std::min<uint64_t>(c->shard_commits, lg_ctx::kShardProfRing);
    double acc[lg_ctx::kShardStages] = {0, 0, 0, 0, 0};
    for (uint64_t s = 0; s < have; s++) {
        hipEvent_t* ev = c->ev_shard[(c->shard_commits - 1 - s) % lg_ctx::kShardProfRing];
        LG_HIP(c, hipEventSynchronize(ev[lg_ctx::kShardStages]));
        for (int i = 0; i < lg_ctx::kShardStages; i++) {
            float ms = 0;
            LG_HIP(c, hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            acc[i] += ms;
        }
        // coset-sharded commits: the stalls of the encode stream waiting for exchange pieces move from "evaluate + hash" to
        // "all-gather" (with one piece on the encode stream itself the collective sits between marks 1 and 2 already)
        // (commits with different numbers of pieces may share the ring: each entry knows its own)
        const uint32_t pairs = c->shard_wait_pairs[(c->shard_commits - 1 - s) % lg_ctx::kShardProfRing];
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
    int rc = grow(c, &c->d_scratch_a, &c->scratch_a_elems, 2 * mat);
    if (rc != LG_OK) return rc;
    fr* d_in = c->d_scratch_a;
    fr* d_co = c->d_scratch_a + mat;
    LG_HIP(c, hipMemcpyAsync(d_in, in, mat * sizeof(fr), hipMemcpyHostToDevice, c->stream));
    const fr* coeffs = d_in;
    if (do_interp) {
        lg::NttArgs a = interp_args(c, d_in, d_co, nullptr, 0, nrows);
        LG_HIP(c, lg::launch_ntt(c->logki, c->logo, false, c->stream, a));
        coeffs = d_co;
    }
    if (!do_eval) return read_back(c, out, coeffs, mat * sizeof(fr));
    rc = grow(c, &c->d_scratch_b, &c->scratch_b_elems, 8 * mat);
    if (rc != LG_OK) return rc;
    rc = grow(c, &c->d_scratch_c, &c->scratch_c_elems, 8 * mat);
    if (rc != LG_OK) return rc;
    const uint64_t sstride = (uint64_t)nrows * c->ki;
    lg::NttArgs a = eval_args(c, coeffs, c->d_scratch_b, sstride, 0, nrows, true);
    LG_HIP(c, lg::launch_ntt(c->logki, c->logo, true, c->stream, a));
    const uint64_t threads = 8 * (uint64_t)mat;
    hipLaunchKernelGGL(lg::planes_to_rows_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->stream, c->d_scratch_b,
                       sstride, (uint64_t)0, nrows, c->ki, (uint32_t)c->lognp, c->r2, c->d_scratch_c);
    LG_HIP(c, hipGetLastError());
    return read_back(c, out, c->d_scratch_c, 8 * mat * sizeof(fr));
}
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
