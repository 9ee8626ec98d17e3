// C ABI over the host-side input pipeline (circuit.hpp): lets the pytest suite, and any other
// FFI caller, drive the C++ mirror of the reference's front end of the path.  No GPU code.
#include <algorithm>
#include <cstring>
#include <exception>
#include <new>
#include <string>

#include "../../include/ligero_host.h"
#include "host_handles.hpp"
#include "expression.hpp"
#include "transcript.hpp"

using namespace ligero;

static thread_local std::string g_err;
template <class F>
static int guarded(F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        g_err = "out of memory";
        return LGH_ERR_OOM;
    } catch (const std::exception& e) {
        g_err = e.what();
        return LGH_ERR_PANIC;
    }
}
static Fr load_fr(const uint64_t* p) { Fr v; std::memcpy(v.l, p, 32); return v; }
static void store_fr(uint64_t* p, const Fr& v) { std::memcpy(p, v.l, 32); }

struct lgh_expr {
    Expression e;
};
template <class F>
static lgh_expr* make_expr(F&& f) {
    lgh_expr* out = nullptr;
    guarded([&] { out = new lgh_expr{f()}; return LGH_OK; });
    return out;
}
static bool have(const lgh_expr* a) {
    if (!a) g_err = "null expression";
    return a != nullptr;
}

extern "C" {

const char* lgh_last_error(void) { return g_err.c_str(); }

lgh_circuit* lgh_circuit_new(void) { return new (std::nothrow) lgh_circuit(); }
void lgh_circuit_destroy(lgh_circuit* c) { delete c; }
int64_t lgh_circuit_num_nodes(const lgh_circuit* c) { return c ? (int64_t)c->c.num_nodes() : (int64_t)LGH_ERR_BAD_ARG; }
int64_t lgh_constant(lgh_circuit* c, const uint64_t v[4]) {
    if (!c || !v) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.constant(load_fr(v)); return 0; });
    return rc ? rc : r;
}
int64_t lgh_new_variable(lgh_circuit* c) {
    if (!c) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.new_variable(); return 0; });
    return rc ? rc : r;
}
int64_t lgh_new_variable_with_label(lgh_circuit* c, const char* label) {
    if (!c || !label) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.new_variable_with_label(label); return 0; });
    return rc ? rc : r;
}
int64_t lgh_get_variable(const lgh_circuit* c, const char* label) {
    if (!c || !label) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.get_variable(label); return 0; });
    return rc ? rc : r;
}
int64_t lgh_circuit_num_gates(const lgh_circuit* c) { return c ? (int64_t)c->c.num_gates() : (int64_t)LGH_ERR_BAD_ARG; }
int64_t lgh_pow_bigint(lgh_circuit* c, uint64_t node, const uint64_t* limbs, uint64_t nlimbs) {
    if (!c || (!limbs && nlimbs)) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.pow_bigint(node, limbs, nlimbs); return 0; });
    return rc ? rc : r;
}
int64_t lgh_indicator(lgh_circuit* c, uint64_t node) {
    if (!c) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.indicator(node); return 0; });
    return rc ? rc : r;
}
int64_t lgh_scalar_product(lgh_circuit* c, const uint64_t* left, const uint64_t* right, uint64_t count) {
    if (!c || (count && (!left || !right))) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] {
        r = (int64_t)c->c.scalar_product(std::vector<size_t>(left, left + count), std::vector<size_t>(right, right + count));
        return 0;
    });
    return rc ? rc : r;
}
int64_t lgh_mul_nodes(lgh_circuit* c, const uint64_t* nodes, uint64_t count) {
    if (!c || (count && !nodes)) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.mul_nodes(std::vector<size_t>(nodes, nodes + count)); return 0; });
    return rc ? rc : r;
}
int lgh_evaluate_multioutput(const lgh_circuit* c, const uint64_t* node_idx, const uint64_t* values, uint64_t count,
                             const uint64_t* outputs, uint64_t n_outputs, uint64_t* values_out, uint64_t* count_out) {
    if (!c || (count && (!node_idx || !values)) || (n_outputs && (!outputs || !values_out)) || !count_out) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        std::vector<std::pair<size_t, Fr>> vars;
        for (uint64_t j = 0; j < count; j++) vars.emplace_back((size_t)node_idx[j], load_fr(values + 4 * j));
        const auto v = c->c.evaluate_multioutput(vars, std::vector<size_t>(outputs, outputs + n_outputs));
        for (size_t j = 0; j < v.size(); j++) store_fr(values_out + 4 * j, v[j]);
        *count_out = v.size();
        return LGH_OK;
    });
}
int64_t lgh_add(lgh_circuit* c, uint64_t l, uint64_t r_) {
    if (!c) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.add(l, r_); return 0; });
    return rc ? rc : r;
}
int64_t lgh_mul(lgh_circuit* c, uint64_t l, uint64_t r_) {
    if (!c) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.mul(l, r_); return 0; });
    return rc ? rc : r;
}
int64_t lgh_pow(lgh_circuit* c, uint64_t node, uint64_t exponent) {
    if (!c) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.pow(node, exponent); return 0; });
    return rc ? rc : r;
}
int64_t lgh_minus(lgh_circuit* c, uint64_t node) {
    if (!c) return LGH_ERR_BAD_ARG;
    int64_t r = -1;
    int rc = guarded([&] { r = (int64_t)c->c.minus(node); return 0; });
    return rc ? rc : r;
}

int lgh_circuit_node(const lgh_circuit* c, uint64_t index, uint32_t* kind, uint64_t* left, uint64_t* right, uint64_t value[4], char* label, uint64_t label_capacity) {
    if (!c || index >= c->c.nodes.size()) return LGH_ERR_BAD_ARG;
    const Node& nd = c->c.nodes[index];
    if (kind) *kind = (uint32_t)nd.kind;
    if (left) *left = nd.l;
    if (right) *right = nd.r;
    if (value) store_fr(value, nd.value);
    if (label && label_capacity) {
        const size_t n = std::min<size_t>(nd.label.size(), label_capacity - 1);
        std::memcpy(label, nd.label.data(), n);
        label[n] = 0;
    }
    return LGH_OK;
}

lgh_expr* lgh_expr_variable(const char* label) {
    if (!label) { g_err = "null label"; return nullptr; }
    return make_expr([&] { return Expression::variable(label); });
}
lgh_expr* lgh_expr_constant(const uint64_t v[4]) {
    if (!v) { g_err = "null value"; return nullptr; }
    return make_expr([&] { return Expression::constant(load_fr(v)); });
}
lgh_expr* lgh_expr_add(const lgh_expr* a, const lgh_expr* b) { return have(a) && have(b) ? make_expr([&] { return a->e + b->e; }) : nullptr; }
lgh_expr* lgh_expr_mul(const lgh_expr* a, const lgh_expr* b) { return have(a) && have(b) ? make_expr([&] { return a->e * b->e; }) : nullptr; }
lgh_expr* lgh_expr_sub(const lgh_expr* a, const lgh_expr* b) { return have(a) && have(b) ? make_expr([&] { return a->e - b->e; }) : nullptr; }
lgh_expr* lgh_expr_neg(const lgh_expr* a) { return have(a) ? make_expr([&] { return -a->e; }) : nullptr; }
lgh_expr* lgh_expr_pow(const lgh_expr* a, uint64_t exponent) { return have(a) ? make_expr([&] { return a->e.pow(exponent); }) : nullptr; }
void lgh_expr_destroy(lgh_expr* e) { delete e; }
int lgh_expr_to_circuit(const lgh_expr* e, lgh_circuit** out) {
    if (!e || !out) return LGH_ERR_BAD_ARG;
    *out = nullptr;
    return guarded([&] {
        lgh_circuit* c = new lgh_circuit();
        try {
            c->c = e->e.to_arithmetic_circuit();
        } catch (...) {
            delete c;
            throw;
        }
        *out = c;
        return LGH_OK;
    });
}

int lgh_circuit_from_r1cs(lgh_circuit** out, const char* r1cs_path) {
    if (!out || !r1cs_path) return LGH_ERR_BAD_ARG;
    *out = nullptr;
    return guarded([&] {
        const R1cs cs = read_r1cs(r1cs_path);
        auto co = ArithmeticCircuit::from_constraint_system(cs);
        lgh_circuit* c = new lgh_circuit();
        c->c = std::move(co.first);
        c->outputs = std::move(co.second);
        c->n_wires = cs.n_wires;
        *out = c;
        return LGH_OK;
    });
}
int64_t lgh_circuit_num_outputs(const lgh_circuit* c) { return c ? (int64_t)c->outputs.size() : (int64_t)LGH_ERR_BAD_ARG; }
int lgh_circuit_outputs(const lgh_circuit* c, uint64_t* out) {
    if (!c || !out) return LGH_ERR_BAD_ARG;
    for (size_t i = 0; i < c->outputs.size(); i++) out[i] = c->outputs[i];
    return LGH_OK;
}

int lgh_instance_new(lgh_instance** out, const lgh_circuit* c, const uint64_t* outputs, uint64_t n_outputs, uint32_t lambda) {
    if (!out || !c || (!outputs && n_outputs)) return LGH_ERR_BAD_ARG;
    *out = nullptr;
    return guarded([&] {
        std::vector<size_t> outs(outputs, outputs + n_outputs);
        *out = new lgh_instance(c->c, std::move(outs), lambda, c->n_wires);
        return LGH_OK;
    });
}
void lgh_instance_destroy(lgh_instance* i) { delete i; }

int lgh_instance_info(const lgh_instance* i, uint64_t out[8]) {
    if (!i || !out) return LGH_ERR_BAD_ARG;
    out[0] = i->inst.m; out[1] = i->inst.k; out[2] = i->inst.n; out[3] = i->inst.t;
    out[4] = i->inst.circuit.num_nodes(); out[5] = i->inst.circuit.num_constants();
    out[6] = i->inst.outputs.size(); out[7] = i->inst.a.nnz();
    return LGH_OK;
}

int lgh_build_preenc(const lgh_instance* i, const uint64_t* node_idx, const uint64_t* values, uint64_t count, uint64_t* preenc_out, int* all_outputs_one) {
    if (!i || (count && (!node_idx || !values)) || !preenc_out) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        std::vector<std::pair<size_t, Fr>> vars;
        vars.reserve(count);
        for (uint64_t j = 0; j < count; j++) vars.emplace_back((size_t)node_idx[j], load_fr(values + 4 * j));
        bool ok = false;
        static_assert(sizeof(Fr) == 32, "Fr is four u64 limbs");
        i->inst.build_preenc_into(vars, reinterpret_cast<Fr*>(preenc_out), &ok);
        if (all_outputs_one) *all_outputs_one = ok ? 1 : 0;
        return LGH_OK;
    });
}

int lgh_build_preenc_with_labels(const lgh_instance* i, const char* const* labels, const uint64_t* values, uint64_t count, uint64_t* preenc_out, int* all_outputs_one) {
    if (!i || (count && (!labels || !values)) || !preenc_out) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        std::vector<std::pair<std::string, Fr>> vars;
        vars.reserve(count);
        for (uint64_t j = 0; j < count; j++) {
            if (!labels[j]) throw std::runtime_error("null label");
            vars.emplace_back(labels[j], load_fr(values + 4 * j));
        }
        bool ok = false;
        i->inst.build_preenc_with_labels_into(vars, reinterpret_cast<Fr*>(preenc_out), &ok);
        if (all_outputs_one) *all_outputs_one = ok ? 1 : 0;
        return LGH_OK;
    });
}

int lgh_gate_map(const lgh_instance* i, uint64_t* npos_out, uint64_t* nconst_out, uint32_t* left, uint32_t* right, uint64_t* constants) {
    if (!i || !npos_out || !nconst_out) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        const auto g = i->inst.gate_map();
        *npos_out = g.left.size();
        *nconst_out = g.constants.size();
        if (left) std::memcpy(left, g.left.data(), g.left.size() * sizeof(uint32_t));
        if (right) std::memcpy(right, g.right.data(), g.right.size() * sizeof(uint32_t));
        if (constants && !g.constants.empty()) std::memcpy(constants, g.constants[0].l, g.constants.size() * sizeof(Fr));
        return LGH_OK;
    });
}

int lgh_trace_program(const lgh_instance* i, uint64_t sizes[7], uint8_t* op, uint32_t* left, uint32_t* right, uint64_t* constants, uint32_t* order,
                      uint64_t* level_off, uint32_t* outputs, uint32_t* pos_of_node) {
    if (!i || !sizes) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        const auto t = i->inst.trace_program();
        sizes[0] = t.op.size(); sizes[1] = t.constants.size(); sizes[2] = t.order.size(); sizes[3] = t.level_off.size() - 1;
        sizes[4] = t.outputs.size(); sizes[5] = t.num_inputs; sizes[6] = t.pos_of_node.size();
        if (op) std::memcpy(op, t.op.data(), t.op.size());
        if (left) std::memcpy(left, t.left.data(), t.left.size() * sizeof(uint32_t));
        if (right) std::memcpy(right, t.right.data(), t.right.size() * sizeof(uint32_t));
        if (constants && !t.constants.empty()) std::memcpy(constants, t.constants[0].l, t.constants.size() * sizeof(Fr));
        if (order && !t.order.empty()) std::memcpy(order, t.order.data(), t.order.size() * sizeof(uint32_t));
        if (level_off) std::memcpy(level_off, t.level_off.data(), t.level_off.size() * sizeof(uint64_t));
        if (outputs && !t.outputs.empty()) std::memcpy(outputs, t.outputs.data(), t.outputs.size() * sizeof(uint32_t));
        if (pos_of_node) std::memcpy(pos_of_node, t.pos_of_node.data(), t.pos_of_node.size() * sizeof(uint32_t));
        return LGH_OK;
    });
}

int lgh_input_positions(const lgh_instance* i, const uint64_t* node_idx, uint64_t count, uint32_t* positions_out) {
    if (!i || (count && (!node_idx || !positions_out))) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        const auto& nodes = i->inst.circuit.nodes;
        // position = formatted index minus the constants in front of it (index 0, the one, keeps its position)
        std::vector<uint32_t> consts_before(nodes.size() + 1, 0);
        for (size_t j = 0; j < nodes.size(); j++) consts_before[j + 1] = consts_before[j] + ((nodes[j].kind == Node::Constant && j != 0) ? 1u : 0u);
        for (uint64_t j = 0; j < count; j++) {
            const size_t f = i->inst.bump_index((size_t)node_idx[j]);
            if (f >= nodes.size()) throw std::runtime_error("index out of bounds: assigned node not in the circuit");
            if (nodes[f].kind != Node::Variable) throw std::runtime_error("Value supplied for non-variable node");
            positions_out[j] = (uint32_t)(f - consts_before[f]);
        }
        return LGH_OK;
    });
}

int lgh_build_w(const lgh_instance* i, const uint64_t* node_idx, const uint64_t* values, uint64_t count, uint64_t* w_out, int* all_outputs_one) {
    if (!i || (count && (!node_idx || !values)) || !w_out) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        std::vector<std::pair<size_t, Fr>> vars;
        vars.reserve(count);
        for (uint64_t j = 0; j < count; j++) vars.emplace_back(i->inst.bump_index((size_t)node_idx[j]), load_fr(values + 4 * j));
        bool ok = false;
        i->inst.build_w_from_formatted(vars, reinterpret_cast<Fr*>(w_out), &ok);
        if (all_outputs_one) *all_outputs_one = ok ? 1 : 0;
        return LGH_OK;
    });
}

int lgh_a_row_mul(const lgh_instance* i, const uint64_t* r, uint64_t* out) {
    if (!i || !r || !out) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        const size_t len = 4 * i->inst.m * i->inst.k;
        std::vector<Fr> row(len);
        for (size_t j = 0; j < len; j++) row[j] = load_fr(r + 4 * j);
        const auto res = i->inst.a.row_mul(row);
        for (size_t j = 0; j < res.size(); j++) store_fr(out + 4 * j, res[j]);
        return LGH_OK;
    });
}

int lgh_a_entries(const lgh_instance* i, uint64_t* row_idx, uint64_t* col_idx, uint64_t* values) {
    if (!i || !row_idx || !col_idx || !values) return LGH_ERR_BAD_ARG;
    size_t o = 0;
    for (size_t r = 0; r < i->inst.a.num_rows(); r++)
        for (const auto& e : i->inst.a.row(r)) {
            row_idx[o] = r; col_idx[o] = e.second;
            store_fr(values + 4 * o, e.first);
            o++;
        }
    return LGH_OK;
}

int lgh_read_witness(const char* path, uint64_t* values_out, uint64_t capacity, uint64_t* count_out) {
    if (!path || !count_out || (!values_out && capacity)) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        const std::vector<Fr> w = read_witness(path);
        *count_out = w.size();
        for (size_t i = 0; i < w.size() && i < capacity; i++) store_fr(values_out + 4 * i, w[i]);
        return LGH_OK;
    });
}

// ---- Fiat-Shamir pieces (transcript.hpp; PARITY UNPINNED, see there) exposed for tests and FFI callers
struct lgh_sponge {
    PoseidonSponge s = PoseidonSponge::test_sponge();
};

void lgh_chacha_block(uint32_t rounds, const uint32_t key[8], const uint32_t words12_15[4], uint32_t out[16]) {
    if (rounds == 20) ChaChaRng<20>::block(key, words12_15, out);
    else if (rounds == 12) ChaChaRng<12>::block(key, words12_15, out);
    else ChaChaRng<8>::block(key, words12_15, out);
}
int lgh_field_elements_from_seed(const uint8_t seed[32], uint64_t n, uint64_t* out) {
    if (!seed || (!out && n)) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        std::array<uint8_t, 32> s;
        std::memcpy(s.data(), seed, 32);
        const auto v = get_field_elements_from_prng(n, s);
        for (size_t i = 0; i < v.size(); i++) store_fr(out + 4 * i, v[i]);
        return LGH_OK;
    });
}
int lgh_distinct_indices_from_seed(const uint8_t seed[32], uint64_t n, uint64_t t, uint64_t* out, uint64_t* count_out) {
    if (!seed || !out || !count_out || n == 0 || t > n) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        std::array<uint8_t, 32> s;
        std::memcpy(s.data(), seed, 32);
        const auto v = get_distinct_indices_from_prng(n, t, s);
        for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
        *count_out = v.size();
        return LGH_OK;
    });
}
lgh_sponge* lgh_sponge_new(void) {
    lgh_sponge* p = nullptr;
    guarded([&] { p = new lgh_sponge(); return LGH_OK; });
    return p;
}
void lgh_sponge_destroy(lgh_sponge* s) { delete s; }
int lgh_sponge_absorb_bytes(lgh_sponge* s, const uint8_t* data, uint64_t len) {
    if (!s || (!data && len)) return LGH_ERR_BAD_ARG;
    return guarded([&] { s->s.absorb_bytes(data, len); return LGH_OK; });
}
int lgh_sponge_absorb_elements(lgh_sponge* s, const uint64_t* elems, uint64_t count) {
    if (!s || (!elems && count)) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        std::vector<Fr> v(count);
        for (size_t i = 0; i < count; i++) v[i] = load_fr(elems + 4 * i);
        s->s.absorb_elements(v);
        return LGH_OK;
    });
}
int lgh_sponge_absorb_elements_x8(lgh_sponge* const sponges[8], const uint64_t* elems, uint64_t count) {
    if (!sponges || (!elems && count)) return LGH_ERR_BAD_ARG;
    for (int j = 0; j < 8; j++)
        if (!sponges[j]) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        std::vector<std::vector<Fr>> v(8, std::vector<Fr>(count));
        PoseidonSponge* sp[8];
        const std::vector<Fr>* el[8];
        for (int j = 0; j < 8; j++) {
            for (size_t i = 0; i < count; i++) v[j][i] = load_fr(elems + 4 * ((size_t)j * count + i));
            sp[j] = &sponges[j]->s;
            el[j] = &v[j];
        }
        PoseidonSponge::absorb_elements_x8(sp, el);
        return LGH_OK;
    });
}
int lgh_ifma_available(void) { return ifma::available() ? 1 : 0; }
int lgh_sponge_squeeze_bytes(lgh_sponge* s, uint64_t n, uint8_t* out) {
    if (!s || (!out && n)) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        const auto b = s->s.squeeze_bytes(n);
        std::memcpy(out, b.data(), n);
        return LGH_OK;
    });
}
int lgh_sponge_squeeze_elements(lgh_sponge* s, uint64_t n, uint64_t* out) {
    if (!s || (!out && n)) return LGH_ERR_BAD_ARG;
    return guarded([&] {
        const auto v = s->s.squeeze_native_field_elements(n);
        for (size_t i = 0; i < v.size(); i++) store_fr(out + 4 * i, v[i]);
        return LGH_OK;
    });
}

}  // extern "C"
