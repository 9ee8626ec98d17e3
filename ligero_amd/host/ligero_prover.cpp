// C ABI over prover.hpp (include/ligero_prover.h): links the device library.
#include <cstring>
#include <exception>
#include <new>
#include <string>

#include "../../include/ligero_prover.h"
#include "host_handles.hpp"
#include "prover.hpp"

using namespace ligero;

struct lgp_prover {
    HipLigero hip;
    lgp_prover(const LigeroInstance& inst, int device) : hip(inst, device) {}
};
struct lgp_proof {
    LigeroProof own;                 // storage of a proof this handle owns (lgp_prove, lgp_prove_batch)
    const LigeroProof* view = &own;  // what the handle shows: its own proof, or one inside a batch prover (lgp_batch_proof)
    lgp_proof() = default;
    lgp_proof(const lgp_proof& o) : own(o.own), view(o.view == &o.own ? &own : o.view) {}
    lgp_proof& operator=(const lgp_proof& o) {
        own = o.own;
        view = (o.view == &o.own) ? &own : o.view;
        return *this;
    }
};
struct lgp_batch_prover {
    HipLigeroBatch hip;
    std::vector<lgp_proof> views;   // borrowed views of the proofs of the last lgp_prove_batch
    lgp_batch_prover(const LigeroInstance& inst, uint32_t batch, int device, unsigned threads) : hip(inst, batch, device, threads) {}
};

static thread_local std::string g_err;
template <class F>
static int guarded(F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        g_err = "out of memory";
        return LGP_ERR_OOM;
    } catch (const DeviceError& e) {
        g_err = e.what();
        return LGP_ERR_DEVICE;
    } catch (const std::exception& e) {
        g_err = e.what();
        return LGP_ERR_PANIC;
    }
}

extern "C" {

const char* lgp_last_error(void) { return g_err.c_str(); }

int lgp_prover_create(lgp_prover** out, const lgh_instance* inst, int device) {
    if (!out || !inst) return LGP_ERR_BAD_ARG;
    *out = nullptr;
    return guarded([&] { *out = new lgp_prover(inst->inst, device); return LGP_OK; });
}
void lgp_prover_destroy(lgp_prover* p) { delete p; }

int lgp_prove(lgp_prover* p, const uint64_t* node_idx, const uint64_t* values, uint64_t count, lgp_proof** proof_out) {
    if (!p || !proof_out || (count && (!node_idx || !values))) return LGP_ERR_BAD_ARG;
    *proof_out = nullptr;
    return guarded([&] {
        std::vector<std::pair<size_t, Fr>> va;
        for (uint64_t i = 0; i < count; i++) {
            Fr v;
            std::memcpy(v.l, values + 4 * i, 32);
            va.emplace_back((size_t)node_idx[i], v);
        }
        PoseidonSponge sponge = PoseidonSponge::test_sponge();
        auto* pr = new lgp_proof();
        try {
            pr->own = p->hip.prove(va, sponge);
        } catch (...) {
            delete pr;
            throw;
        }
        *proof_out = pr;
        return LGP_OK;
    });
}

int lgp_verify(lgp_prover* p, const lgp_proof* proof, int* accepted_out) {
    if (!p || !proof || !accepted_out) return LGP_ERR_BAD_ARG;
    return guarded([&] {
        PoseidonSponge sponge = PoseidonSponge::test_sponge();
        *accepted_out = p->hip.verify(*proof->view, sponge) ? 1 : 0;
        return LGP_OK;
    });
}
void lgp_proof_destroy(lgp_proof* proof) { delete proof; }

int lgp_batch_prover_create(lgp_batch_prover** out, const lgh_instance* inst, uint32_t batch, int device, uint32_t threads) {
    if (!out || !inst || batch == 0) return LGP_ERR_BAD_ARG;
    *out = nullptr;
    return guarded([&] { *out = new lgp_batch_prover(inst->inst, batch, device, threads); return LGP_OK; });
}
void lgp_batch_prover_destroy(lgp_batch_prover* p) { delete p; }
uint32_t lgp_batch_prover_threads(const lgp_batch_prover* p) { return p ? p->hip.threads() : 0; }

int lgp_prove_batch(lgp_batch_prover* p, const uint64_t* node_idx, const uint64_t* values, uint64_t count, lgp_proof** proofs_out) {
    if (!p || !node_idx || !values || count == 0) return LGP_ERR_BAD_ARG;
    const uint32_t B = p->hip.batch();
    if (proofs_out)
        for (uint32_t b = 0; b < B; b++) proofs_out[b] = nullptr;
    return guarded([&] {
        std::vector<std::vector<std::pair<size_t, Fr>>> va(B);
        for (uint32_t b = 0; b < B; b++)
            for (uint64_t i = 0; i < count; i++) {
                Fr v;
                std::memcpy(v.l, values + 4 * ((uint64_t)b * count + i), 32);
                va[b].emplace_back((size_t)node_idx[i], v);
            }
        const std::vector<LigeroProof>& proofs = p->hip.prove(va);
        if (proofs_out)
            for (uint32_t b = 0; b < B; b++) {
                proofs_out[b] = new lgp_proof();
                proofs_out[b]->own = proofs[b];   // a copy the caller owns
            }
        p->views.assign(B, lgp_proof());
        for (uint32_t b = 0; b < B; b++) p->views[b].view = &proofs[b];
        return LGP_OK;
    });
}
const lgp_proof* lgp_batch_proof(const lgp_batch_prover* p, uint32_t b) {
    return (p && b < p->views.size()) ? &p->views[b] : nullptr;
}

int lgp_proof_info(const lgp_proof* proof, uint64_t info_out[6], uint8_t root_out[32]) {
    if (!proof || !info_out || !root_out) return LGP_ERR_BAD_ARG;
    const LigeroProof& p = *proof->view;
    info_out[0] = p.interleaved_proof.preenc_u_lc.size();
    info_out[1] = p.linear_constraints_proof.polynomial.size();
    info_out[2] = p.quadratic_constraints_proof.polynomial.size();
    info_out[3] = p.interleaved_proof.open.columns.size();
    info_out[4] = p.interleaved_proof.open.columns.empty() ? 0 : p.interleaved_proof.open.columns[0].size();
    info_out[5] = p.interleaved_proof.open.paths.empty() ? 0 : p.interleaved_proof.open.paths[0].auth_path.size();
    std::memcpy(root_out, p.u_root.data(), 32);
    return LGP_OK;
}

int lgp_proof_tamper(lgp_proof* proof, int what, uint64_t index) {
    if (!proof || proof->view != &proof->own) return LGP_ERR_BAD_ARG;   // borrowed views are read-only
    LigeroProof& p = proof->own;
    auto bump = [](Fr& x) { x = fr_add(x, fr_one()); };
    auto col_elem = [&](OpenedColumns& o) -> int {
        if (o.columns.empty()) return LGP_ERR_BAD_ARG;
        auto& c = o.columns[index % o.columns.size()];
        bump(c[(index / o.columns.size()) % c.size()]);
        return LGP_OK;
    };
    switch (what) {
        case 0: p.u_root[index % 32] ^= 1; return LGP_OK;
        case 1: if (p.interleaved_proof.preenc_u_lc.empty()) return LGP_ERR_BAD_ARG; bump(p.interleaved_proof.preenc_u_lc[index % p.interleaved_proof.preenc_u_lc.size()]); return LGP_OK;
        case 2: if (p.linear_constraints_proof.polynomial.empty()) return LGP_ERR_BAD_ARG; bump(p.linear_constraints_proof.polynomial[index % p.linear_constraints_proof.polynomial.size()]); return LGP_OK;
        case 3: if (p.quadratic_constraints_proof.polynomial.empty()) return LGP_ERR_BAD_ARG; bump(p.quadratic_constraints_proof.polynomial[index % p.quadratic_constraints_proof.polynomial.size()]); return LGP_OK;
        case 4: return col_elem(p.interleaved_proof.open);
        case 5: return col_elem(p.linear_constraints_proof.open);
        case 6: return col_elem(p.quadratic_constraints_proof.open);
        case 7: {
            auto& paths = p.interleaved_proof.open.paths;
            if (paths.empty() || paths[0].auth_path.empty()) return LGP_ERR_BAD_ARG;
            auto& ph = paths[index % paths.size()];
            ph.auth_path[(index / paths.size()) % ph.auth_path.size()][0] ^= 1;
            return LGP_OK;
        }
        case 8: {
            auto& paths = p.linear_constraints_proof.open.paths;
            if (paths.empty()) return LGP_ERR_BAD_ARG;
            paths[index % paths.size()].leaf_index ^= 1;
            return LGP_OK;
        }
        default: return LGP_ERR_BAD_ARG;
    }
}

}  // extern "C"
