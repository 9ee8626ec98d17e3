// C++ host-side mirror, above the C ABI (include/ligero_hip.h), of the pieces of
// NP-Eng/ligero's `LigeroCircuit` that sit on the encode-and-commit path.  The reference is
// Rust; no Rust toolchain exists in the build image, so the host layer a Rust caller would
// write with `extern "C"` is written here in C++ with the reference's
// names, argument meaning and error behaviour:
//
//   DenseMatrix<F>                      src/matrices/mod.rs:128-172   (rows: Vec<Vec<F>>)
//   LigeroCircuit::reed_solomon_*       src/ligero/mod.rs:998-1012
//   LigeroCircuit::as_matrix            src/ligero/mod.rs:1014-1017
//   commit  (prove_inner lines 521-551) src/ligero/mod.rs:521-551
//   open_columns                        src/ligero/mod.rs:935-955
//   Path { leaf_sibling_hash, auth_path, leaf_index }   (ark-crypto-primitives merkle_tree::Path)
//
// Where the reference panics (unwrap at mod.rs:539, 549; index out of bounds in
// DenseMatrix::column) this layer throws ligero::Error carrying the lg_status.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ligero_hip.h"

namespace ligero {

// ark_bn254::Fr in memory: 4 x u64 little-endian limbs, Montgomery form
struct Fr {
    uint64_t limbs[4];
    bool operator==(const Fr& o) const { return limbs[0] == o.limbs[0] && limbs[1] == o.limbs[1] && limbs[2] == o.limbs[2] && limbs[3] == o.limbs[3]; }
};
using Digest = std::array<uint8_t, 32>;

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& what) : std::runtime_error(what + ": " + lg_status_string(st)), status(st) {}
};

// src/matrices/mod.rs:128-136
struct DenseMatrix {
    std::vector<std::vector<Fr>> rows;
    explicit DenseMatrix(std::vector<std::vector<Fr>> r) : rows(std::move(r)) {}
    size_t num_columns() const { return rows.at(0).size(); }
    // flatten for the C ABI (rows are separate heap allocations in the reference)
    std::vector<Fr> flatten() const {
        std::vector<Fr> out;
        out.reserve(rows.size() * num_columns());
        for (const auto& r : rows) {
            if (r.size() != num_columns()) throw Error(LG_ERR_BAD_ARG, "DenseMatrix: ragged rows");
            out.insert(out.end(), r.begin(), r.end());
        }
        return out;
    }
};

// ark-crypto-primitives merkle_tree::Path<C> for C = TestMerkleTreeParams
struct Path {
    Digest leaf_sibling_hash;
    std::vector<Digest> auth_path;  // root side first, log2(n) - 1 entries
    size_t leaf_index;
};

// What prove_inner holds after line 551: u_polynomial_coeffs, u (device resident), u_tree
// (device resident), u_root.
struct Commitment {
    std::vector<std::vector<Fr>> u_polynomial_coeffs;
    Digest u_root;
};

class LigeroCircuit {
public:
    // m, k as computed by compute_dimensions (mod.rs:275-279); n = 8k (mod.rs:284)
    LigeroCircuit(size_t m, size_t k, int device = 0) : m_(m), k_(k), n_(8 * k) {
        int st = lg_ctx_create(&ctx_, device, static_cast<uint32_t>(4 * m), static_cast<uint32_t>(k), static_cast<uint32_t>(n_));
        if (st != LG_OK) throw Error(st, "LigeroCircuit::new");
    }
    ~LigeroCircuit() { lg_ctx_destroy(ctx_); }
    LigeroCircuit(const LigeroCircuit&) = delete;
    LigeroCircuit& operator=(const LigeroCircuit&) = delete;

    size_t m() const { return m_; }
    size_t k() const { return k_; }
    size_t n() const { return n_; }

    // mod.rs:1014-1017
    std::vector<std::vector<Fr>> as_matrix(const std::vector<Fr>& vec) const {
        std::vector<std::vector<Fr>> out;
        for (size_t i = 0; i + k_ <= vec.size(); i += k_) out.emplace_back(vec.begin() + i, vec.begin() + i + k_);
        return out;
    }

    // mod.rs:998-1002 (msg is resized to k with zeros)
    std::vector<Fr> reed_solomon_interpolate(std::vector<Fr> msg) const {
        if (msg.size() > k_) throw Error(LG_ERR_BAD_ARG, "reed_solomon_interpolate: message longer than k");
        msg.resize(k_, Fr{{0, 0, 0, 0}});
        std::vector<Fr> out(k_);
        check(lg_reed_solomon_interpolate(ctx_, msg[0].limbs, 1, out[0].limbs), "reed_solomon_interpolate");
        return out;
    }
    // mod.rs:1004-1008 (coefficients beyond k are not representable in this code: deg < k)
    std::vector<Fr> reed_solomon_evaluate(std::vector<Fr> coeffs) const {
        if (coeffs.size() > k_) throw Error(LG_ERR_BAD_ARG, "reed_solomon_evaluate: more than k coefficients");
        coeffs.resize(k_, Fr{{0, 0, 0, 0}});
        std::vector<Fr> out(n_);
        check(lg_reed_solomon_evaluate(ctx_, coeffs[0].limbs, 1, out[0].limbs), "reed_solomon_evaluate");
        return out;
    }
    // mod.rs:1010-1012
    std::vector<Fr> reed_solomon(std::vector<Fr> msg) const {
        if (msg.size() > k_) throw Error(LG_ERR_BAD_ARG, "reed_solomon: message longer than k");
        msg.resize(k_, Fr{{0, 0, 0, 0}});
        std::vector<Fr> out(n_);
        check(lg_reed_solomon(ctx_, msg[0].limbs, 1, out[0].limbs), "reed_solomon");
        return out;
    }

    // prove_inner, mod.rs:521-551: encode every row, hash every column, Merkle-commit
    Commitment commit(const DenseMatrix& preenc_u) {
        if (preenc_u.rows.size() != 4 * m_ || preenc_u.num_columns() != k_) throw Error(LG_ERR_BAD_ARG, "commit: preenc_u is not 4m x k");
        std::vector<Fr> flat = preenc_u.flatten(), coeffs(flat.size());
        Commitment c;
        check(lg_encode_commit(ctx_, flat[0].limbs, coeffs[0].limbs, c.u_root.data()), "commit");
        c.u_polynomial_coeffs = as_matrix(coeffs);
        return c;
    }

    // mod.rs:944-952; `indices` from get_distinct_indices_from_prng (src/utils.rs:31-55)
    std::pair<std::vector<std::vector<Fr>>, std::vector<Path>> open_columns(const std::vector<size_t>& indices) {
        const size_t t = indices.size(), rows = 4 * m_;
        size_t plen = 0;
        while ((size_t{2} << plen) < n_) plen++;  // log2(n) - 1
        std::vector<uint32_t> idx(indices.begin(), indices.end());
        std::vector<Fr> cols(t * rows);
        std::vector<uint8_t> sib(t * 32), paths(t * plen * 32);
        check(lg_open_columns(ctx_, 0, idx.data(), static_cast<uint32_t>(t), cols.empty() ? nullptr : cols[0].limbs, sib.data(), paths.data()), "open_columns");
        std::pair<std::vector<std::vector<Fr>>, std::vector<Path>> out;
        for (size_t c = 0; c < t; c++) {
            out.first.emplace_back(cols.begin() + c * rows, cols.begin() + (c + 1) * rows);
            Path p;
            p.leaf_index = indices[c];
            std::copy(sib.begin() + 32 * c, sib.begin() + 32 * (c + 1), p.leaf_sibling_hash.begin());
            p.auth_path.resize(plen);
            for (size_t l = 0; l < plen; l++) std::copy(paths.begin() + 32 * (c * plen + l), paths.begin() + 32 * (c * plen + l + 1), p.auth_path[l].begin());
            out.second.push_back(std::move(p));
        }
        return out;
    }

    // prove_interleaved's preenc_u.row_mul(&r_interleaved) (mod.rs:658, src/matrices/mod.rs:138-149)
    std::vector<Fr> interleaved_row_mul(const std::vector<Fr>& r) {
        if (r.size() != 4 * m_) throw Error(LG_ERR_BAD_ARG, "interleaved_row_mul: r must have 4m entries");
        std::vector<Fr> out(k_);
        check(lg_interleaved_row_mul(ctx_, r[0].limbs, out[0].limbs), "interleaved_row_mul");
        return out;
    }
    // prove_linear_constraints (mod.rs:723-736): r_a = self.a.row_mul(&r_linear), 4m*k entries ->
    // coefficients of sum_i u_polys[i] * ifft(r_a_i), 2k of them (trailing zeros are NOT trimmed)
    std::vector<Fr> linear_constraint_poly(const std::vector<Fr>& r_a) {
        if (r_a.size() != 4 * m_ * k_) throw Error(LG_ERR_BAD_ARG, "linear_constraint_poly: r_a must have 4mk entries");
        std::vector<Fr> out(2 * k_);
        check(lg_linear_constraint_poly(ctx_, r_a[0].limbs, out[0].limbs), "linear_constraint_poly");
        return out;
    }
    // prove_quadratic_constraints (mod.rs:842-848): r has m entries -> 2k coefficients
    std::vector<Fr> quadratic_constraint_poly(const std::vector<Fr>& r) {
        if (r.size() != m_) throw Error(LG_ERR_BAD_ARG, "quadratic_constraint_poly: r must have m entries");
        std::vector<Fr> out(2 * k_);
        check(lg_quadratic_constraint_poly(ctx_, r[0].limbs, out[0].limbs), "quadratic_constraint_poly");
        return out;
    }

    // leaf digests, for a host that wants to own an ark MerkleTree (mod.rs:544-549)
    std::vector<Digest> leaves() {
        std::vector<Digest> out(n_);
        check(lg_read_leaves(ctx_, out[0].data()), "leaves");
        return out;
    }

    lg_ctx* raw() { return ctx_; }

private:
    void check(int st, const char* what) const {
        if (st != LG_OK) throw Error(st, std::string(what) + " (" + lg_last_error(ctx_) + ")");
    }
    size_t m_, k_, n_;
    lg_ctx* ctx_ = nullptr;
};

}  // namespace ligero
