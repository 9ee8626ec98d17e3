// Host side of the path, input end (SURVEY.md section 8f #3 and the host part of #1): everything
// between the reference's fixtures (.r1cs + witness) and the matrix `preenc_u` the device
// commits to, plus the sparse constraint matrix A whose product with the linear-test challenge
// feeds lg_linear_constraint_poly.  Mirrors, with the reference's names and behaviour:
//
//   read_r1cs                                   circom .r1cs v1 (what src/reader.rs gets via ark-circom)
//   ArithmeticCircuit::{constant, new_variable, add, mul, add_nodes, pow, minus,
//       compile_sparse_scalar_product, from_constraint_system,
//       evaluation_trace_multioutput}           src/arithmetic_circuit/mod.rs:65-239, 247-271, 325-358, 455-520
//   SparseMatrix::{row_mul, h_stack, v_stack, identity, zero, neg}   src/matrices/mod.rs:6-126
//   LigeroCircuit::{new, insert_one, bump_index, compute_dimensions,
//       reed_solomon_parameters, generate_matrices}                  src/ligero/mod.rs:147-433
//   prove_inner's x/y/z/w assembly + as_matrix -> preenc_u           src/ligero/mod.rs:476-516, 1014-1017
//
// The classes are templates over the element type E (field.hpp: ark_bn254::Fr, ark_bls12_377::Fq -- the two fields the
// reference's tests instantiate); the plain names (ArithmeticCircuit, SparseMatrix, LigeroInstance) are the BN254 instances
// every circom fixture and BASELINE config uses.  Elements are in Montgomery form, same limbs as the C ABI.  Where the reference panics
// this layer throws std::runtime_error with the reference's message.  Product code: independent
// of oracle/.
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../csrc/host_fr.h"
#include "field.hpp"

namespace ligero {

// ---------------------------------------------------------------- .r1cs v1
struct R1cs {
    uint32_t n_wires = 0, n_pub_out = 0, n_pub_in = 0, n_prv_in = 0;
    std::vector<uint8_t> prime;                                   // little endian
    using Lc = std::vector<std::pair<Fr, uint32_t>>;              // (coefficient [Montgomery], wire)
    std::vector<Lc> a, b, c;
};

inline R1cs read_r1cs(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + got);
    std::fclose(f);
    auto need = [&](size_t o, size_t len) { if (o > d.size() || len > d.size() - o) throw std::runtime_error("r1cs: truncated file"); };
    auto u32 = [&](size_t o) { need(o, 4); uint32_t v; std::memcpy(&v, d.data() + o, 4); return v; };
    auto u64 = [&](size_t o) { need(o, 8); uint64_t v; std::memcpy(&v, d.data() + o, 8); return v; };
    if (d.size() < 12 || std::memcmp(d.data(), "r1cs", 4) != 0 || u32(4) != 1) throw std::runtime_error("not a circom r1cs v1 file");
    const uint32_t nsec = u32(8);
    size_t off = 12, hdr = 0, cons = 0;
    for (uint32_t s = 0; s < nsec; s++) {
        const uint32_t typ = u32(off);
        const uint64_t len = u64(off + 4);
        off += 12;
        if (typ == 1) hdr = off;
        if (typ == 2) cons = off;
        need(off, len);   // also keeps off from wrapping around
        off += len;
    }
    if (!hdr || !cons) throw std::runtime_error("r1cs: missing header or constraint section");
    R1cs r;
    const uint32_t fs = u32(hdr);
    if (fs != 32) throw std::runtime_error("r1cs: only 32-byte fields are supported");
    need(hdr + 4, fs + 28);
    r.prime.assign(d.begin() + hdr + 4, d.begin() + hdr + 4 + fs);
    uint8_t pbytes[32];
    for (int i = 0; i < 4; i++) std::memcpy(pbytes + 8 * i, &lg_host::kP.l[i], 8);
    if (std::memcmp(pbytes, r.prime.data(), 32) != 0) throw std::runtime_error("r1cs: prime is not the BN254 scalar field");
    size_t o = hdr + 4 + fs;
    r.n_wires = u32(o); r.n_pub_out = u32(o + 4); r.n_pub_in = u32(o + 8); r.n_prv_in = u32(o + 12);
    const uint32_t ncons = u32(o + 24);
    // every wire has an 8-byte label in the wire map and every constraint at least three 4-byte counts: a header that
    // claims more than the file can hold is corrupt (and would otherwise make the compiler allocate gigabytes)
    if ((uint64_t)r.n_wires * 8 > d.size() || (uint64_t)ncons * 12 > d.size()) throw std::runtime_error("r1cs: header counts exceed the file size");
    o = cons;
    auto read_lc = [&]() {
        R1cs::Lc lc;
        const uint32_t nnz = u32(o);
        o += 4;
        for (uint32_t i = 0; i < nnz; i++) {
            const uint32_t wire = u32(o);
            if (wire >= r.n_wires) throw std::runtime_error("r1cs: wire index out of range");
            Fr v;
            need(o + 4, 32);
            std::memcpy(v.l, d.data() + o + 4, 32);
            o += 4 + 32;
            lc.emplace_back(lg_host::to_mont(v), wire);
        }
        return lc;
    };
    for (uint32_t i = 0; i < ncons; i++) {
        r.a.push_back(read_lc());
        r.b.push_back(read_lc());
        r.c.push_back(read_lc());
    }
    return r;
}

// ---------------------------------------------------------------- arithmetic circuit
struct NodeBase {
    enum Kind : uint8_t { Variable, Constant, Add, Mul };
};
template <class E>
struct NodeT : NodeBase {
    Kind kind;
    size_t l = 0, r = 0;   // Add / Mul operands
    E value{};             // Constant
    std::string label;     // Variable
};
using Node = NodeT<Fr>;

template <class E>
class ArithmeticCircuitT {
public:
    using F = Field<E>;
    using Node = NodeT<E>;
    std::vector<Node> nodes;
    std::map<E, size_t, ElemLess<E>> constants;           // value -> index (mod.rs:31)
    std::unordered_map<std::string, size_t> variables;    // label -> index (mod.rs:33)

    size_t num_nodes() const { return nodes.size(); }
    size_t num_constants() const { return constants.size(); }
    size_t num_variables() const { return variables.size(); }
    size_t last() const { return nodes.size() - 1; }

    size_t constant(const E& v) {                         // mod.rs:76-84
        auto it = constants.find(v);
        if (it != constants.end()) return it->second;
        nodes.push_back(Node{{}, Node::Constant, 0, 0, v, {}});
        constants[v] = nodes.size() - 1;
        return nodes.size() - 1;
    }
    size_t new_variable_with_label(const std::string& label) {   // mod.rs:92-100
        if (variables.count(label)) throw std::runtime_error("Variable label already in use: " + label);
        nodes.push_back(Node{{}, Node::Variable, 0, 0, {}, label});
        variables[label] = nodes.size() - 1;
        return nodes.size() - 1;
    }
    size_t new_variable() { return new_variable_with_label("var_" + std::to_string(num_variables())); }   // mod.rs:107-109
    size_t add(size_t l, size_t r) {                       // mod.rs:125-131
        if (l >= nodes.size()) throw std::runtime_error("Left operand to Add not in circuit:");
        if (r >= nodes.size()) throw std::runtime_error("Right operand to Add not in circuit:");
        nodes.push_back(Node{{}, Node::Add, l, r, {}, {}});
        return nodes.size() - 1;
    }
    size_t mul(size_t l, size_t r) {                       // mod.rs:139-145
        if (l >= nodes.size()) throw std::runtime_error("Left operand to Mul not in circuit:");
        if (r >= nodes.size()) throw std::runtime_error("Right operand to Mul not in circuit:");
        nodes.push_back(Node{{}, Node::Mul, l, r, {}, {}});
        return nodes.size() - 1;
    }
    size_t add_nodes(const std::vector<size_t>& idx) {     // mod.rs:148-153
        if (idx.empty()) throw std::runtime_error("add_nodes: empty list");   // reduce().unwrap() panics upstream
        size_t acc = idx[0];
        for (size_t i = 1; i < idx.size(); i++) acc = add(acc, idx[i]);
        return acc;
    }
    size_t mul_nodes(const std::vector<size_t>& idx) {     // mod.rs:156-161
        if (idx.empty()) throw std::runtime_error("mul_nodes: empty list");
        size_t acc = idx[0];
        for (size_t i = 1; i < idx.size(); i++) acc = mul(acc, idx[i]);
        return acc;
    }
    size_t mul_unchecked(size_t l, size_t r) {             // mod.rs:134-136
        nodes.push_back(Node{{}, Node::Mul, l, r, {}, {}});
        return nodes.size() - 1;
    }
    size_t num_gates() const {                             // mod.rs:55-63
        size_t g = 0;
        for (const auto& nd : nodes) g += (nd.kind == Node::Add || nd.kind == Node::Mul);
        return g;
    }
    std::vector<size_t> new_variables(size_t num) {        // mod.rs:111-113
        std::vector<size_t> v;
        for (size_t i = 0; i < num; i++) v.push_back(new_variable());
        return v;
    }
    size_t get_variable(const std::string& label) const { // mod.rs:115-117
        const auto it = variables.find(label);
        if (it == variables.end()) throw std::runtime_error("Variable not in circuit");
        return it->second;
    }
    // pow_bigint (mod.rs:164-179): exponent as little-endian u64 limbs; pow_binary (mod.rs:188-200): square-and-multiply,
    // most significant bit first, accumulator starts at the node itself
    size_t pow_bigint(size_t node, const uint64_t* limbs, size_t nlimbs) {
        if (node >= nodes.size())
            throw std::runtime_error("Base node (" + std::to_string(node) + ") not in the circuit (which contains " + std::to_string(nodes.size()) + " nodes)");
        long top = (long)nlimbs * 64 - 1;
        while (top >= 0 && !((limbs[top / 64] >> (top % 64)) & 1)) top--;
        // exponent 0: the reference's bit list is empty and pow_binary returns the node itself (x^0 is NOT 1 there)
        size_t cur = node;
        for (long b = top - 1; b >= 0; b--) {
            cur = mul_unchecked(cur, cur);
            if ((limbs[b / 64] >> (b % 64)) & 1) cur = mul_unchecked(cur, node);
        }
        return cur;
    }
    size_t pow(size_t node, uint64_t exponent) { return pow_bigint(node, &exponent, 1); }   // mod.rs:182-184
    size_t indicator(size_t node) {                        // mod.rs:203-217: x^(p-1) -- 0 at 0, 1 elsewhere
        const E m1 = F::from_mont(F::neg(F::one()));
        return pow_bigint(node, m1.l, F::kLimbs);
    }
    size_t scalar_product(const std::vector<size_t>& l, const std::vector<size_t>& r) {      // mod.rs:228-239 (zip: shorter side)
        std::vector<size_t> products;
        for (size_t i = 0; i < l.size() && i < r.size(); i++) products.push_back(mul_unchecked(l[i], r[i]));
        return add_nodes(products);
    }
    size_t minus(size_t node) {                            // mod.rs:220-223
        const size_t m1 = constant(F::neg(F::one()));
        return mul(m1, node);
    }
    size_t compile_sparse_scalar_product(const R1cs::Lc& row) {   // mod.rs:501-520 (circom fixtures: BN254 only)
        static_assert(std::is_same<E, Fr>::value, "R1CS files carry BN254 Fr coefficients");
        std::vector<std::pair<size_t, size_t>> consts;
        for (const auto& t : row) consts.emplace_back(constant(t.first), (size_t)t.second);
        std::vector<size_t> products;
        for (const auto& cv : consts)
            products.push_back((cv.first == 0 || cv.second == 0) ? cv.first + cv.second : mul(cv.first, cv.second));
        return add_nodes(products);
    }
    // mod.rs:455-495.  Zero coefficients are dropped, as ark-relations' to_matrices does.
    static std::pair<ArithmeticCircuitT, std::vector<size_t>> from_constraint_system(const R1cs& cs) {
        static_assert(std::is_same<E, Fr>::value, "R1CS files carry BN254 Fr coefficients");
        ArithmeticCircuitT c;
        const size_t one = c.constant(F::one());
        for (uint32_t i = 1; i < cs.n_wires; i++) c.new_variable();
        auto rows = [&](const std::vector<R1cs::Lc>& mat) {
            std::vector<size_t> out;
            for (const auto& lc : mat) {
                R1cs::Lc nz;
                for (const auto& t : lc)
                    if (!fr_is_zero(t.first)) nz.push_back(t);
                out.push_back(c.compile_sparse_scalar_product(nz));
            }
            return out;
        };
        const auto a = rows(cs.a), b = rows(cs.b), cc = rows(cs.c);
        std::vector<size_t> ab, minus_c, outputs;
        for (size_t i = 0; i < a.size(); i++) ab.push_back(c.mul(a[i], b[i]));
        const size_t m1 = c.constant(F::neg(F::one()));
        for (size_t i = 0; i < cc.size(); i++) minus_c.push_back(c.mul(cc[i], m1));
        for (size_t i = 0; i < ab.size(); i++) outputs.push_back(c.add_nodes({ab[i], minus_c[i], one}));
        return {std::move(c), std::move(outputs)};
    }
    // filter_constants (mod.rs:545-603): drops every constant whose value appeared earlier in the list and renumbers the
    // gates' operands; returns the kept nodes and value -> index of the kept constant
    static std::pair<std::vector<Node>, std::map<E, size_t, ElemLess<E>>> filter_constants(const std::vector<Node>& in) {
        constexpr size_t kDropped = ~size_t{0};
        std::map<E, size_t, ElemLess<E>> consts;
        std::vector<size_t> filtered(in.size(), kDropped);
        size_t removed = 0;
        for (size_t i = 0; i < in.size(); i++) {
            if (in[i].kind == Node::Constant) {
                if (consts.count(in[i].value)) { removed++; continue; }
                consts[in[i].value] = i - removed;
            }
            filtered[i] = i - removed;
        }
        auto operand = [&](size_t o) {
            if (o >= in.size()) throw std::runtime_error("index out of bounds: gate operand not in the node list");
            return in[o].kind == Node::Constant ? consts.at(in[o].value) : filtered[o];
        };
        std::vector<Node> out;
        out.reserve(in.size() - removed);
        for (size_t i = 0; i < in.size(); i++) {
            if (filtered[i] == kDropped) continue;
            Node nd = in[i];
            if (nd.kind == Node::Add || nd.kind == Node::Mul) { nd.l = operand(nd.l); nd.r = operand(nd.r); }
            out.push_back(std::move(nd));
        }
        return {std::move(out), std::move(consts)};
    }

    // evaluation_trace_multioutput (mod.rs:325-358) with the reference's Option: only what the outputs depend on is
    // evaluated (inner_evaluate, mod.rs:247-271), everything else that was not assigned stays unset; a needed variable
    // without a value is the reference's "Uninitialised variable" panic.  Circuits made by the builders only refer to
    // earlier nodes (one backward sweep marks what is needed, one forward sweep computes it: what the 5 M-node circuits
    // use); circuits made from an Expression (expression.hpp) number the root last and refer forwards too, and get
    // the reference's depth-first walk with an explicit stack instead of recursion.
    struct Trace {
        std::vector<E> value;
        std::vector<uint8_t> set;      // the reference's Some / None
    };
    // What does not depend on the assignment, computed once per (circuit, outputs) by callers that evaluate repeatedly
    // (LigeroInstance): whether gates only refer backwards, and which nodes the outputs need.
    struct EvalPlan {
        bool backward_only = true;
        std::vector<uint8_t> need;     // backward_only circuits: 1 = some output depends on the node
    };
    EvalPlan eval_plan(const std::vector<size_t>& outputs) const {
        EvalPlan p;
        for (size_t i = 0; i < nodes.size() && p.backward_only; i++) {
            const Node& nd = nodes[i];
            if ((nd.kind == Node::Add || nd.kind == Node::Mul) && (nd.l >= i || nd.r >= i)) p.backward_only = false;
        }
        for (size_t o : outputs)
            if (o >= nodes.size()) throw std::runtime_error("index out of bounds: output node not in the circuit");
        if (p.backward_only) {
            p.need.assign(nodes.size(), 0);
            for (size_t o : outputs) p.need[o] = 1;
            for (size_t i = nodes.size(); i-- > 0;) {
                const Node& nd = nodes[i];
                if (p.need[i] && (nd.kind == Node::Add || nd.kind == Node::Mul)) p.need[nd.l] = p.need[nd.r] = 1;
            }
        }
        return p;
    }
    Trace evaluation_trace_multioutput(const std::vector<std::pair<size_t, E>>& vars, const std::vector<size_t>& outputs) const {
        Trace t;
        evaluation_trace_into(t, eval_plan(outputs), vars, outputs);
        return t;
    }
    // the same into storage the caller keeps between evaluations (a 5 M-node trace is 170 MB: fresh memory for it costs more
    // than the field arithmetic)
    void evaluation_trace_into(Trace& t, const EvalPlan& plan, const std::vector<std::pair<size_t, E>>& vars, const std::vector<size_t>& outputs) const {
        t.value.resize(nodes.size());
        t.set.assign(nodes.size(), 0);
        for (size_t i = 0; i < nodes.size(); i++) {
            const Node& nd = nodes[i];
            if (nd.kind == Node::Constant) { t.value[i] = nd.value; t.set[i] = 1; }
        }
        for (const auto& v : vars) {
            if (v.first >= nodes.size()) throw std::runtime_error("index out of bounds: assigned node not in the circuit");
            if (nodes[v.first].kind != Node::Variable) throw std::runtime_error("Value supplied for non-variable node");
            t.value[v.first] = v.second;
            t.set[v.first] = 1;
        }
        if (plan.backward_only) {
            const std::vector<uint8_t>& need = plan.need;
            for (size_t i = 0; i < nodes.size(); i++) {
                const Node& nd = nodes[i];
                if (!need[i] || t.set[i]) continue;
                if (nd.kind == Node::Variable) throw std::runtime_error("Uninitialised variable");
                t.value[i] = nd.kind == Node::Add ? F::add(t.value[nd.l], t.value[nd.r]) : F::mul(t.value[nd.l], t.value[nd.r]);
                t.set[i] = 1;
            }
            return;
        }
        std::vector<uint8_t> open(nodes.size(), 0);          // gates whose operands are being evaluated
        std::vector<size_t> stack;
        for (size_t o : outputs) {
            stack.push_back(o);
            while (!stack.empty()) {
                const size_t i = stack.back();
                if (t.set[i]) { stack.pop_back(); continue; }
                const Node& nd = nodes[i];
                if (nd.kind == Node::Variable) throw std::runtime_error("Uninitialised variable");
                if (nd.l >= nodes.size() || nd.r >= nodes.size()) throw std::runtime_error("index out of bounds: gate operand not in the circuit");
                if (t.set[nd.l] && t.set[nd.r]) {
                    t.value[i] = nd.kind == Node::Add ? F::add(t.value[nd.l], t.value[nd.r]) : F::mul(t.value[nd.l], t.value[nd.r]);
                    t.set[i] = 1;
                    open[i] = 0;
                    stack.pop_back();
                    continue;
                }
                // an unset operand that is already open is an ancestor of this gate: the reference's recursion would not end
                open[i] = 1;
                for (size_t c : {nd.r, nd.l}) {
                    if (t.set[c]) continue;
                    if (open[c]) throw std::runtime_error("circuit has a cycle");
                    stack.push_back(c);
                }
            }
        }
    }
    Trace evaluation_trace(const std::vector<std::pair<size_t, E>>& vars, size_t node) const {   // mod.rs:279-306
        return evaluation_trace_multioutput(vars, {node});
    }
    std::vector<std::pair<size_t, E>> with_labels(const std::vector<std::pair<std::string, E>>& vars) const {   // mod.rs:313-316 etc.
        std::vector<std::pair<size_t, E>> out;
        for (const auto& v : vars) out.emplace_back(get_variable(v.first), v.second);
        return out;
    }
    E evaluate_node(const std::vector<std::pair<size_t, E>>& vars, size_t node) const {         // mod.rs:373-375
        return evaluation_trace(vars, node).value[node];
    }
    E evaluate(const std::vector<std::pair<size_t, E>>& vars) const { return evaluate_node(vars, last()); }   // mod.rs:401-403
    E evaluate_node_with_labels(const std::vector<std::pair<std::string, E>>& vars, size_t node) const { return evaluate_node(with_labels(vars), node); }
    E evaluate_with_labels(const std::vector<std::pair<std::string, E>>& vars) const { return evaluate_node(with_labels(vars), last()); }
    // mod.rs:381-387: the values of the output nodes in NODE order (each once), not in the order `outputs` lists them
    std::vector<E> evaluate_multioutput(const std::vector<std::pair<size_t, E>>& vars, const std::vector<size_t>& outputs) const {
        const Trace t = evaluation_trace_multioutput(vars, outputs);
        std::vector<uint8_t> is_out(nodes.size(), 0);
        for (size_t o : outputs) is_out[o] = 1;
        std::vector<E> out;
        for (size_t i = 0; i < nodes.size(); i++)
            if (is_out[i] && t.set[i]) out.push_back(t.value[i]);
        return out;
    }
    std::vector<E> evaluate_multioutput_with_labels(const std::vector<std::pair<std::string, E>>& vars, const std::vector<size_t>& outputs) const {
        return evaluate_multioutput(with_labels(vars), outputs);
    }
};
using ArithmeticCircuit = ArithmeticCircuitT<Fr>;

// ---------------------------------------------------------------- sparse matrix (src/matrices/mod.rs:6-126)
// The reference keeps a Vec of rows, each a Vec<(F, usize)>.  Same rows, same entry order, but stored compressed
// (row_ptr + one entry array): the constraint matrix of the 2^20-constraint circuit has 41 M rows and 45 M entries, and
// one heap allocation per row would cost more memory than the entries themselves.
template <class E>
struct SparseMatrixT {
    using F = Field<E>;
    struct Entry {            // (value, column) -- the reference's (F, usize); a plain struct so that rows move with memcpy
        E first;
        size_t second;
    };
    struct RowView {
        const Entry* b;
        const Entry* e;
        const Entry* begin() const { return b; }
        const Entry* end() const { return e; }
        size_t size() const { return (size_t)(e - b); }
        bool empty() const { return b == e; }
        const Entry& operator[](size_t i) const { return b[i]; }
    };
    size_t num_cols = 0;
    std::vector<uint64_t> row_ptr{0};
    std::vector<Entry> ent;
    explicit SparseMatrixT(size_t cols = 0) : num_cols(cols) {}
    size_t num_rows() const { return row_ptr.size() - 1; }
    RowView row(size_t i) const { return RowView{ent.data() + row_ptr[i], ent.data() + row_ptr[i + 1]}; }
    void push_row(std::initializer_list<Entry> r) { ent.insert(ent.end(), r.begin(), r.end()); row_ptr.push_back(ent.size()); }
    void push_row(const std::vector<Entry>& r) { ent.insert(ent.end(), r.begin(), r.end()); row_ptr.push_back(ent.size()); }
    void push_row1(const E& v, size_t col) { ent.push_back(Entry{v, col}); row_ptr.push_back(ent.size()); }
    void push_empty_row() { row_ptr.push_back(ent.size()); }
    void push_empty_rows(size_t n) { row_ptr.resize(row_ptr.size() + n, ent.size()); }
    static SparseMatrixT identity(size_t n) {
        SparseMatrixT m(n);
        m.ent.reserve(n);
        m.row_ptr.reserve(n + 1);
        const E one = F::one();
        for (size_t i = 0; i < n; i++) m.push_row1(one, i);
        return m;
    }
    static SparseMatrixT zero(size_t nr, size_t nc) {
        SparseMatrixT m(nc);
        m.push_empty_rows(nr);
        return m;
    }
    SparseMatrixT h_stack(const SparseMatrixT& o) && {
        if (num_rows() != o.num_rows()) throw std::runtime_error("Row number mismatch in when stacking matrices horizontally");
        SparseMatrixT out(num_cols + o.num_cols);
        out.ent.resize(ent.size() + o.ent.size());
        out.row_ptr.resize(row_ptr.size());
        Entry* w = out.ent.data();
        for (size_t i = 0; i < num_rows(); i++) {
            const size_t na = row_ptr[i + 1] - row_ptr[i], nb = o.row_ptr[i + 1] - o.row_ptr[i];
            if (na) std::memcpy(w, ent.data() + row_ptr[i], na * sizeof(Entry));
            w += na;
            const Entry* src = o.ent.data() + o.row_ptr[i];
            for (size_t j = 0; j < nb; j++) { w[j].first = src[j].first; w[j].second = src[j].second + num_cols; }
            w += nb;
            out.row_ptr[i + 1] = (uint64_t)(w - out.ent.data());
        }
        return out;
    }
    SparseMatrixT v_stack(const SparseMatrixT& o) && {
        if (num_cols != o.num_cols) throw std::runtime_error("Column number mismatch in when stacking matrices vertically");
        const uint64_t base = ent.size();
        ent.insert(ent.end(), o.ent.begin(), o.ent.end());
        row_ptr.reserve(row_ptr.size() + o.num_rows());
        for (size_t i = 1; i < o.row_ptr.size(); i++) row_ptr.push_back(base + o.row_ptr[i]);
        return std::move(*this);
    }
    SparseMatrixT neg() && {
        for (auto& e : ent) e.first = F::neg(e.first);
        return std::move(*this);
    }
    // mod.rs:100-110: result[col] += row[i] * value for every entry of row i
    std::vector<E> row_mul(const std::vector<E>& row) const {
        std::vector<E> out(num_cols);
        row_mul_into(row.data(), row.size(), out.data());
        return out;
    }
    // the same into caller-owned storage of num_cols elements
    void row_mul_into(const E* rowv, size_t row_len, E* out) const {
        for (size_t c = 0; c < num_cols; c++) out[c] = F::zero();
        const size_t n = row_len < num_rows() ? row_len : num_rows();
        const E one = F::one(), minus_one = F::neg(F::one());
        for (size_t i = 0; i < n; i++)
            for (const auto& e : row(i)) {  // almost every entry of A is +-1: add / subtract instead of multiplying
                if (F::eq(e.first, one)) out[e.second] = F::add(out[e.second], rowv[i]);
                else if (F::eq(e.first, minus_one)) out[e.second] = F::sub(out[e.second], rowv[i]);
                else out[e.second] = F::add(out[e.second], F::mul(rowv[i], e.first));
            }
    }
    size_t nnz() const { return ent.size(); }
};
using SparseMatrix = SparseMatrixT<Fr>;

// ---------------------------------------------------------------- witness files
// The reference's test reads circom/poseidon/witness.json with serde (src/ligero/tests.rs:384-390: a JSON array of
// decimal strings, wire 0 first); snarkjs' binary .wtns holds the same values ("wtns", version 2, section 1 =
// field size + prime + count, section 2 = count x 32-byte little-endian values).  Both give the wire
// values in order, Montgomery form.
inline std::vector<uint8_t> read_file_bytes(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> d;
    uint8_t buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) d.insert(d.end(), buf, buf + got);
    std::fclose(f);
    return d;
}
inline Fr fr_from_le_bytes_canonical(const uint8_t* p) {
    Fr c;
    for (int i = 0; i < 4; i++) {
        c.l[i] = 0;
        for (int j = 0; j < 8; j++) c.l[i] |= (uint64_t)p[8 * i + j] << (8 * j);
    }
    if (lg_host::geq(c, lg_host::kP)) throw std::runtime_error("witness value is not below the field modulus");
    return lg_host::to_mont(c);
}
inline std::vector<Fr> read_witness(const std::string& path) {
    const std::vector<uint8_t> d = read_file_bytes(path);
    std::vector<Fr> out;
    if (d.size() >= 12 && std::memcmp(d.data(), "wtns", 4) == 0) {
        auto u32 = [&](size_t o) { if (o > d.size() || d.size() - o < 4) throw std::runtime_error("truncated .wtns"); uint32_t v; std::memcpy(&v, &d[o], 4); return v; };
        auto u64 = [&](size_t o) { if (o > d.size() || d.size() - o < 8) throw std::runtime_error("truncated .wtns"); uint64_t v; std::memcpy(&v, &d[o], 8); return v; };
        if (u32(4) != 2) throw std::runtime_error("unsupported .wtns version");
        const uint32_t nsec = u32(8);
        size_t off = 12;
        uint64_t count = 0;
        bool have_header = false;
        for (uint32_t s = 0; s < nsec; s++) {
            const uint32_t type = u32(off);
            const uint64_t len = u64(off + 4);
            off += 12;
            if (off > d.size() || len > d.size() - off) throw std::runtime_error("truncated .wtns section");
            if (type == 1) {
                if (len < 40) throw std::runtime_error("truncated .wtns header");
                if (u32(off) != 32) throw std::runtime_error(".wtns field size is not 32 bytes");
                static const uint8_t prime[32] = {0x01, 0x00, 0x00, 0xf0, 0x93, 0xf5, 0xe1, 0x43, 0x91, 0x70, 0xb9, 0x79, 0x48, 0xe8, 0x33, 0x28,
                                                  0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};
                if (std::memcmp(&d[off + 4], prime, 32) != 0) throw std::runtime_error(".wtns prime is not BN254's scalar field");
                count = u32(off + 36);
                have_header = true;
            } else if (type == 2) {
                if (!have_header || len != count * 32) throw std::runtime_error(".wtns value section does not match its header");
                for (uint64_t i = 0; i < count; i++) out.push_back(fr_from_le_bytes_canonical(&d[off + 32 * i]));
            }
            off += len;
        }
        return out;
    }
    // JSON: ["1", "1234...", ...]
    size_t i = 0;
    auto skip = [&] { while (i < d.size() && (d[i] == ' ' || d[i] == '\n' || d[i] == '\r' || d[i] == '\t' || d[i] == ',')) i++; };
    skip();
    if (i >= d.size() || d[i] != '[') throw std::runtime_error("witness file is neither .wtns nor a JSON array");
    i++;
    const Fr ten = fr_from_u64(10);
    for (;;) {
        skip();
        if (i >= d.size()) throw std::runtime_error("unterminated JSON array");
        if (d[i] == ']') break;
        const bool quoted = d[i] == '"';
        if (quoted) i++;
        Fr v = fr_zero();
        size_t digits = 0;
        while (i < d.size() && d[i] >= '0' && d[i] <= '9') {
            v = fr_add(fr_mul(v, ten), fr_from_u64((uint64_t)(d[i] - '0')));   // values are reduced mod p like F::from_str of a canonical string
            i++;
            digits++;
        }
        if (digits == 0 || digits > 78) throw std::runtime_error("bad number in witness JSON");
        if (quoted) {
            if (i >= d.size() || d[i] != '"') throw std::runtime_error("bad string in witness JSON");
            i++;
        }
        out.push_back(v);
    }
    return out;
}

// ---------------------------------------------------------------- LigeroCircuit::new and the x/y/z/w assembly
// calculate_t of ark-poly-commit's linear_codes::utils (called at mod.rs:287-292), f64 arithmetic
inline size_t calculate_t(size_t sec_param, size_t d_num, size_t d_den, size_t codeword_len, int field_bits = 254) {
    const double residual = (double)codeword_len / std::pow(2.0, field_bits);
    const double rhs = std::log2(std::pow(2.0, -(double)sec_param) - residual);
    if (!std::isnormal(rhs)) throw std::runtime_error("For the given codeword length and the required security guarantee, the field is not big enough.");
    const double denom = std::log2(1.0 - 0.5 * (double)d_num / (double)d_den);
    if (!std::isnormal(denom)) throw std::runtime_error("The distance is wrong");
    const size_t t = (size_t)std::ceil((rhs - 1.0) / denom);
    return t < codeword_len ? t : codeword_len;
}

template <class E>
class LigeroInstanceT {
public:
    using F = Field<E>;
    using Node = NodeT<E>;
    using SparseMatrix = SparseMatrixT<E>;
    ArithmeticCircuitT<E> circuit;
    std::vector<size_t> outputs;
    size_t one_index = 0;
    bool one_found = false;
    size_t m = 0, k = 0, n = 0, t = 0;
    SparseMatrix a;

    // mod.rs:147-228
    LigeroInstanceT(ArithmeticCircuitT<E> c, std::vector<size_t> outs, size_t lambda) : circuit(std::move(c)) {
        auto it = circuit.constants.find(F::one());
        if (it != circuit.constants.end()) { one_index = it->second; one_found = true; } else { one_index = 1; one_found = false; }
        if (one_index != 0) insert_one();
        const size_t sol_vec_length = 1 + circuit.num_nodes() - circuit.num_constants() + outs.size();
        m = (size_t)std::ceil(std::sqrt((double)sol_vec_length));       // compute_dimensions, mod.rs:275-279
        k = 1;
        while (k < m) k <<= 1;
        n = 8 * k;                                                       // reed_solomon_parameters, mod.rs:283-294
        t = calculate_t(lambda, n - k + 1, n, n, F::kModulusBits);
        std::vector<size_t> index_map(circuit.nodes.size(), kNoIndex);   // mod.rs:179-194 (a HashMap there; dense here)
        index_map[0] = 0;
        size_t seen = 0;
        for (size_t i = 1; i < circuit.nodes.size(); i++) {
            if (circuit.nodes[i].kind == Node::Constant) seen++;
            else index_map[i] = i - seen;
        }
        for (size_t o : outs) outputs.push_back(bump_index(o));
        plan_ = circuit.eval_plan(outputs);
        build_program(sol_vec_length);
        a = generate_matrices(index_map, m * k);
    }

    size_t bump_index(size_t index) const {                              // mod.rs:230-242
        if (one_found) return index < one_index ? index + 1 : (index == one_index ? 0 : index);
        return index + 1;
    }

    // prove (mod.rs:449-452) + prove_inner (mod.rs:476-516): assignment by ORIGINAL node index.  Writes preenc_u as the
    // flat row-major 4m x k matrix the C ABI takes -- x, y, z, w each padded to m k (mod.rs:506-509), cut into rows of k
    // (as_matrix, mod.rs:1014-1017) and stacked [X; Y; Z; W] (mod.rs:511-516) is exactly the concatenation of the four
    // padded vectors -- without the intermediate Vecs (at 2^20 constraints they are 1.3 GB each way).
    void build_preenc_into(const std::vector<std::pair<size_t, E>>& var_assignment, E* out, bool* all_outputs_one = nullptr) const {
        std::vector<std::pair<size_t, E>> bumped;
        bumped.reserve(var_assignment.size());
        for (const auto& v : var_assignment) bumped.emplace_back(bump_index(v.first), v.second);
        build_preenc_from_formatted(bumped, out, all_outputs_one);
    }
    // prove_with_labels (mod.rs:580-611): labels resolve through the FORMATTED circuit's variable map (insert_one has
    // already bumped it, mod.rs:268-270), so the indices go to prove_inner as they are
    std::vector<std::pair<size_t, E>> resolve_labels(const std::vector<std::pair<std::string, E>>& var_assignment) const {
        std::vector<std::pair<size_t, E>> out;
        out.reserve(var_assignment.size());
        for (const auto& v : var_assignment) {
            const auto it = circuit.variables.find(v.first);
            if (it == circuit.variables.end()) throw std::runtime_error("Variable not found: " + v.first);
            out.emplace_back(it->second, v.second);
        }
        return out;
    }
    void build_preenc_with_labels_into(const std::vector<std::pair<std::string, E>>& var_assignment, E* out, bool* all_outputs_one = nullptr) const {
        build_preenc_from_formatted(resolve_labels(var_assignment), out, all_outputs_one);
    }
    // prove_inner (mod.rs:476-516): assignment by index into the formatted circuit
    // what a prover keeps between proofs of one instance: the trace storage, and which buffer already holds a preenc_u
    struct Scratch {
        typename ArithmeticCircuitT<E>::Trace trace;
        const E* filled = nullptr;
        size_t filled_begin = 0, filled_end = 0;
        void buffer_replaced() { filled = nullptr; }     // call when the buffer is reallocated or written by anything else
    };
    void build_preenc_from_formatted(const std::vector<std::pair<size_t, E>>& bumped, E* out, bool* all_outputs_one = nullptr, Scratch* scratch = nullptr) const {
        build_preenc_range_from_formatted(bumped, 0, 4 * m * k, out, all_outputs_one, scratch);
    }
    // the same, but only elements [elem_begin, elem_end) of the flat row-major 4m x k matrix, written to out[0 ..): the row
    // shard one rank of a coset-sharded proof uploads (elem = row * k; the trace is still evaluated in full)
    void build_preenc_range_from_formatted(const std::vector<std::pair<size_t, E>>& bumped, size_t elem_begin, size_t elem_end, E* out,
                                           bool* all_outputs_one = nullptr, Scratch* scratch = nullptr, std::atomic<uint64_t>* positions_done = nullptr) const {
        build_preenc_ranges_from_formatted(bumped, {{elem_begin, elem_end}}, out, all_outputs_one, scratch, positions_done);
    }
    // ... or several element ranges at once (one evaluation of the trace), written back to back into out[0 ..): the rows one rank
    // of a row-relay proof keeps in the blocks layout -- its share of each of the X, Y, Z, W blocks
    // positions_done (optional): advanced, with release semantics, to the number of leading positions of the solution vector whose
    // entries in `out` are final -- a consumer on another thread may ship those rows while the rest is still being evaluated
    // (lg_encode_commit_from_witness_progress); m k once everything, the zero padding included, is final
    void build_preenc_ranges_from_formatted(const std::vector<std::pair<size_t, E>>& bumped, const std::vector<std::pair<size_t, size_t>>& ranges, E* out,
                                            bool* all_outputs_one = nullptr, Scratch* scratch = nullptr, std::atomic<uint64_t>* positions_done = nullptr) const {
        struct Finish {     // whatever way the function is left -- an exception included -- the consumer must not wait for ever
            std::atomic<uint64_t>* p; uint64_t all;
            ~Finish() { if (p) p->store(all, std::memory_order_release); }
        } finish{positions_done, (uint64_t)(m * k)};
        const size_t mk = m * k;
        size_t total = 0;
        std::vector<size_t> out_off;
        for (size_t i = 0; i < ranges.size(); i++) {
            if (ranges[i].second > 4 * mk || ranges[i].first > ranges[i].second || (i && ranges[i].first < ranges[i - 1].second))
                throw std::runtime_error("build_preenc: element ranges outside the 4m x k matrix or out of order");
            out_off.push_back(total);
            total += ranges[i].second - ranges[i].first;
        }
        const size_t elem_begin = ranges.empty() ? 0 : ranges.front().first, elem_end = ranges.size() == 1 ? ranges.front().second : elem_begin + total;
        const bool single = ranges.size() == 1;
        // flat index of the matrix -> index into out, or npos
        auto locate = [&](size_t flat) -> size_t {
            for (size_t i = 0; i < ranges.size(); i++)
                if (flat >= ranges[i].first && flat < ranges[i].second) return out_off[i] + (flat - ranges[i].first);
            return ~size_t{0};
        };
        typename ArithmeticCircuitT<E>::Trace local;
        typename ArithmeticCircuitT<E>::Trace& trace = scratch ? scratch->trace : local;
        // the all-zero limbs are the field's zero.  WHICH elements get a value depends on the circuit alone, so a buffer that
        // already holds an earlier preenc_u of this instance (same range) needs no second pass over its 1.3 GB
        const bool zeroed = single && scratch && scratch->filled == out && scratch->filled_begin == elem_begin && scratch->filled_end == elem_end;
        if (!zeroed) std::memset(static_cast<void*>(out), 0, total * sizeof(E));
        if (scratch) { scratch->filled = single ? out : nullptr; scratch->filled_begin = elem_begin; scratch->filled_end = elem_end; }
        if (!prog_.kind.empty()) {
            // One pass over the compact program: evaluate node i and drop its value(s) into x / y / z / w at once.  (The
            // node structs are 88 bytes each; walking 5 M of them twice -- trace, then assembly -- was memory-bound.)
            std::vector<E>& val = trace.value;
            std::vector<uint8_t>& set = trace.set;
            const size_t nn = prog_.kind.size();
            val.resize(nn);
            set.assign(nn, 0);
            for (const auto& v : bumped) {
                if (v.first >= nn) throw std::runtime_error("index out of bounds: assigned node not in the circuit");
                if (prog_.kind[v.first] != Node::Variable) throw std::runtime_error("Value supplied for non-variable node");
                val[v.first] = v.second;
                set[v.first] = 1;
            }
            const bool whole = single && elem_begin == 0 && elem_end == 4 * mk;
            auto put = [&](size_t flat, const E& v) {
                if (whole) out[flat] = v;
                else if (single) { if (flat >= elem_begin && flat < elem_end) out[flat - elem_begin] = v; }
                else { const size_t at = locate(flat); if (at != ~size_t{0}) out[at] = v; }
            };
            size_t pos = 0, ci = 0;
            for (size_t i = 0; i < nn; i++) {
                if (positions_done && (i & 0xffff) == 0) positions_done->store(pos, std::memory_order_release);
                const uint8_t kd = prog_.kind[i];
                if (kd == Node::Constant) {
                    val[i] = prog_.constants[ci++];
                    if (i != 0) continue;                                   // only the leading one takes a position (mod.rs:491)
                    put(3 * mk + pos, val[i]);
                } else if (kd == Node::Variable) {
                    if (!set[i])
                        throw std::runtime_error(plan_.need[i] ? "Uninitialised variable"
                                                               : "Uninitialised variable. Make sure the circuit only contains nodes upon which the final output truly depends");
                    put(3 * mk + pos, val[i]);
                } else {
                    const E& lhs = val[prog_.l[i]];
                    const E& rhs = val[prog_.r[i]];
                    if (kd == Node::Mul) {
                        const E z = F::mul(lhs, rhs);
                        put(pos, lhs); put(mk + pos, rhs); put(2 * mk + pos, z);
                        val[i] = z;
                    } else {
                        val[i] = F::add(lhs, rhs);
                    }
                    put(3 * mk + pos, val[i]);
                }
                pos++;
            }
            if (all_outputs_one) {
                *all_outputs_one = true;
                for (size_t o : outputs)
                    if (!F::eq(val[o], F::one())) *all_outputs_one = false;
            }
            return;
        }
        // evaluation_trace_multioutput + expect on every node (mod.rs:476-478): a node the outputs do not depend on and
        // that is not an assigned variable is a panic there, not a silently evaluated gate
        circuit.evaluation_trace_into(trace, plan_, bumped, outputs);
        for (size_t i = 0; i < trace.set.size(); i++)
            if (!trace.set[i]) throw std::runtime_error("Uninitialised variable. Make sure the circuit only contains nodes upon which the final output truly depends");
        const std::vector<E>& sol = trace.value;
        if (all_outputs_one) {
            *all_outputs_one = true;
            for (size_t o : outputs)
                if (!F::eq(sol[o], F::one())) *all_outputs_one = false;
        }
        auto put = [&](size_t flat, const E& v) {
            const size_t at = locate(flat);
            if (at != ~size_t{0}) out[at] = v;
        };
        size_t pos = 0;
        for (size_t i = 0; i < circuit.nodes.size(); i++) {
            const Node& nd = circuit.nodes[i];
            if (nd.kind == Node::Constant && i != 0) continue;
            if (pos >= mk) throw std::runtime_error("solution vector longer than m * k");
            put(3 * mk + pos, sol[i]);                                                       // w
            if (nd.kind == Node::Mul) { put(pos, sol[nd.l]); put(mk + pos, sol[nd.r]); put(2 * mk + pos, sol[i]); }   // x, y, z
            pos++;
        }
    }
    // a1 on the device (include/ligero_hip.h lg_upload_gate_map): preenc_u is w plus wiring.  For every position p of the
    // solution vector (the kept nodes in node order, mod.rs:483-503) the sources of x[p] and y[p] -- the operands of the Mul
    // gate sitting there: another position of w, or a constant that has no position (mod.rs:491 keeps only the leading one)
    struct GateMap {
        static constexpr uint32_t kNone = 0xffffffffu, kConst = 0x80000000u;
        std::vector<uint32_t> left, right;
        std::vector<E> constants;
    };
    GateMap gate_map() const {
        const auto& nodes = circuit.nodes;
        if (nodes.size() >= GateMap::kConst) throw std::runtime_error("gate map: circuit too large for 31-bit positions");
        std::vector<uint32_t> src(nodes.size());           // node -> position of w, or kConst | index into constants
        GateMap g;
        uint32_t pos = 0;
        for (size_t i = 0; i < nodes.size(); i++) {
            if (nodes[i].kind == Node::Constant && i != 0) {
                src[i] = GateMap::kConst | (uint32_t)g.constants.size();
                g.constants.push_back(nodes[i].value);
            } else {
                src[i] = pos++;
            }
        }
        if (pos > m * k) throw std::runtime_error("solution vector longer than m * k");
        g.left.assign(pos, GateMap::kNone);
        g.right.assign(pos, GateMap::kNone);
        for (size_t i = 0; i < nodes.size(); i++)
            if (nodes[i].kind == Node::Mul) {               // (operands may sit after the gate in expression-made circuits)
                g.left[src[i]] = src[nodes[i].l];
                g.right[src[i]] = src[nodes[i].r];
            }
        return g;
    }
    // f3 on the device (include/ligero_hip.h lg_upload_trace_program): evaluation_trace_multioutput (arithmetic_circuit/mod.rs:325-358)
    // as a program over the POSITIONS of the solution vector, scheduled by dependency level -- every gate of a level has its operands
    // in earlier levels, so a level is one data-parallel launch and the order inside it is free.  Circuits compiled from R1CS are a
    // few levels deep whatever their size (every wire is an assigned variable: a constraint's nodes hang off variables directly);
    // expression-made circuits as deep as their longest chain.  Forward references are fine (levels, not node order, schedule).
    struct TraceProgram {
        static constexpr uint8_t kInput = 0, kAdd = 1, kMul = 2, kOne = 3;    // kOne: the leading constant (position 0, mod.rs:491)
        std::vector<uint8_t> op;             // [npos]
        std::vector<uint32_t> left, right;   // [npos]: gates only -- a position, or GateMap::kConst | index into `constants`
        std::vector<E> constants;            // the same list, in the same order, as gate_map().constants
        std::vector<uint32_t> order;         // the gates' positions, level by level
        std::vector<uint64_t> level_off;     // [levels + 1] into order
        std::vector<uint32_t> outputs;       // positions of the output nodes (each must evaluate to one, mod.rs:519)
        std::vector<uint32_t> pos_of_node;   // formatted node index -> position (GateMap::kNone: a constant without one)
        size_t num_inputs = 0;               // variables: every one of them must be assigned (mod.rs:476-478)
    };
    TraceProgram trace_program() const {
        const auto& nodes = circuit.nodes;
        const size_t nn = nodes.size();
        if (nn >= GateMap::kConst) throw std::runtime_error("trace program: circuit too large for 31-bit positions");
        TraceProgram t;
        t.pos_of_node.assign(nn, GateMap::kNone);
        std::vector<uint32_t> src(nn);
        uint32_t pos = 0;
        for (size_t i = 0; i < nn; i++) {
            if (nodes[i].kind == Node::Constant && i != 0) {
                src[i] = GateMap::kConst | (uint32_t)t.constants.size();
                t.constants.push_back(nodes[i].value);
            } else {
                src[i] = t.pos_of_node[i] = pos++;
            }
        }
        if (pos > m * k) throw std::runtime_error("solution vector longer than m * k");
        // level of every node: 0 for constants and variables, 1 + max over the operands for gates
        std::vector<uint32_t> level(nn, 0);
        if (plan_.backward_only) {
            for (size_t i = 0; i < nn; i++)
                if (nodes[i].kind == Node::Add || nodes[i].kind == Node::Mul) level[i] = 1 + std::max(level[nodes[i].l], level[nodes[i].r]);
        } else {
            std::vector<uint8_t> state(nn, 0);               // 0 untouched, 1 open (operands being visited), 2 done
            std::vector<size_t> stack;
            for (size_t root = 0; root < nn; root++) {
                if (state[root]) continue;
                stack.push_back(root);
                while (!stack.empty()) {
                    const size_t i = stack.back();
                    const Node& nd = nodes[i];
                    if (state[i] == 2) { stack.pop_back(); continue; }
                    if (nd.kind != Node::Add && nd.kind != Node::Mul) { state[i] = 2; stack.pop_back(); continue; }
                    if (nd.l >= nn || nd.r >= nn) throw std::runtime_error("index out of bounds: gate operand not in the circuit");
                    if (state[nd.l] == 2 && state[nd.r] == 2) {
                        level[i] = 1 + std::max(level[nd.l], level[nd.r]);
                        state[i] = 2;
                        stack.pop_back();
                        continue;
                    }
                    state[i] = 1;
                    for (size_t c : {nd.r, nd.l}) {
                        if (state[c] == 2) continue;
                        if (state[c] == 1) throw std::runtime_error("circuit has a cycle");
                        stack.push_back(c);
                    }
                }
            }
        }
        t.op.assign(pos, TraceProgram::kInput);
        t.left.assign(pos, GateMap::kNone);
        t.right.assign(pos, GateMap::kNone);
        uint32_t levels = 0;
        for (size_t i = 0; i < nn; i++) {
            const Node& nd = nodes[i];
            if (nd.kind == Node::Constant) { if (i == 0) t.op[0] = TraceProgram::kOne; continue; }
            if (nd.kind == Node::Variable) { t.num_inputs++; continue; }
            const uint32_t p = src[i];
            t.op[p] = nd.kind == Node::Add ? TraceProgram::kAdd : TraceProgram::kMul;
            t.left[p] = src[nd.l];
            t.right[p] = src[nd.r];
            levels = std::max(levels, level[i]);
        }
        t.level_off.assign((size_t)levels + 1, 0);
        for (size_t i = 0; i < nn; i++)
            if (level[i]) t.level_off[level[i]]++;               // level L counted into slot L; slot 0 stays 0
        for (size_t l = 1; l <= levels; l++) t.level_off[l] += t.level_off[l - 1];    // level_off[L] = end of level L = begin of level L + 1
        t.order.resize(levels ? t.level_off[levels] : 0);
        std::vector<uint64_t> cursor(t.level_off.begin(), t.level_off.end());          // cursor[L - 1] = begin of level L
        for (size_t i = 0; i < nn; i++)
            if (level[i]) t.order[cursor[level[i] - 1]++] = src[i];
        for (size_t o : outputs) {
            if (src[o] & GateMap::kConst) throw std::runtime_error("trace program: an output is a constant without a position");
            t.outputs.push_back(src[o]);
        }
        return t;
    }
    // w alone: the W block of preenc_u, m k elements (zero padded)
    void build_w_from_formatted(const std::vector<std::pair<size_t, E>>& bumped, E* out, bool* all_outputs_one = nullptr, Scratch* scratch = nullptr,
                                std::atomic<uint64_t>* positions_done = nullptr) const {
        build_preenc_range_from_formatted(bumped, 3 * m * k, 4 * m * k, out, all_outputs_one, scratch, positions_done);
    }
    std::vector<std::vector<E>> build_preenc_u(const std::vector<std::pair<size_t, E>>& var_assignment, bool* all_outputs_one = nullptr) const {
        std::vector<E> flat(4 * m * k);
        build_preenc_into(var_assignment, flat.data(), all_outputs_one);
        std::vector<std::vector<E>> rows;
        rows.reserve(4 * m);
        for (size_t i = 0; i < 4 * m; i++) rows.emplace_back(flat.begin() + i * k, flat.begin() + (i + 1) * k);
        return rows;
    }

private:
    void insert_one() {                                                  // mod.rs:244-271
        if (one_found) circuit.nodes.erase(circuit.nodes.begin() + one_index);
        circuit.nodes.insert(circuit.nodes.begin(), Node{{}, Node::Constant, 0, 0, F::one(), {}});
        for (auto& nd : circuit.nodes)
            if (nd.kind == Node::Add || nd.kind == Node::Mul) { nd.l = bump_index(nd.l); nd.r = bump_index(nd.r); }
        for (auto& kv : circuit.constants) kv.second = bump_index(kv.second);
        circuit.constants[F::one()] = 0;
        for (auto& kv : circuit.variables) kv.second = bump_index(kv.second);
    }

    typename ArithmeticCircuitT<E>::EvalPlan plan_;
    // the circuit as three flat arrays + the constants in node order, for circuits the one-pass builder can take: gates refer
    // backwards only, every gate is needed by some output (otherwise prove_inner panics, and the general path words the panic),
    // indices fit 32 bits, the solution vector fits m k
    struct Program {
        std::vector<uint8_t> kind;
        std::vector<uint32_t> l, r;
        std::vector<E> constants;
    } prog_;
    void build_program(size_t sol_vec_length) {
        const size_t nn = circuit.nodes.size();
        if (!plan_.backward_only || nn >= 0xffffffffull || sol_vec_length > m * k) return;
        for (size_t i = 0; i < nn; i++) {
            const Node& nd = circuit.nodes[i];
            if ((nd.kind == Node::Add || nd.kind == Node::Mul) && !plan_.need[i]) return;
        }
        prog_.kind.resize(nn);
        prog_.l.resize(nn);
        prog_.r.resize(nn);
        for (size_t i = 0; i < nn; i++) {
            const Node& nd = circuit.nodes[i];
            prog_.kind[i] = (uint8_t)nd.kind;
            prog_.l[i] = (uint32_t)nd.l;
            prog_.r[i] = (uint32_t)nd.r;
            if (nd.kind == Node::Constant) prog_.constants.push_back(nd.value);
        }
    }
    static constexpr size_t kNoIndex = ~size_t{0};
    static size_t at(const std::vector<size_t>& m, size_t key) {
        // the reference unwraps here (mod.rs:345 etc.): a gate whose operands are both constants panics
        if (key >= m.size() || m[key] == kNoIndex) throw std::runtime_error("called `Option::unwrap()` on a `None` value: gate operand is a constant without an index (mul/add of two constants)");
        return m[key];
    }

    SparseMatrix generate_matrices(const std::vector<size_t>& index_map, size_t num_cols) const {   // mod.rs:296-433
        const auto& nodes = circuit.nodes;
        SparseMatrix p_x(num_cols), p_y(num_cols), p_z(num_cols), p_add(num_cols);
        const E one = F::one(), minus_one = F::neg(F::one());
        auto add_row = [&](size_t l, size_t r) {
            std::vector<typename SparseMatrix::Entry> row;
            if (nodes[l].kind == Node::Constant) row = {{nodes[l].value, 0}, {one, at(index_map, r)}};
            else if (nodes[r].kind == Node::Constant) row = {{one, at(index_map, l)}, {nodes[r].value, 0}};
            else row = {{one, at(index_map, l)}, {one, at(index_map, r)}};
            return row;
        };
        auto mul_rows = [&](size_t l, size_t r) {
            if (nodes[l].kind == Node::Constant) { p_x.push_row({{nodes[l].value, 0}}); p_y.push_row({{one, at(index_map, r)}}); }
            else if (nodes[r].kind == Node::Constant) { p_x.push_row({{one, at(index_map, l)}}); p_y.push_row({{nodes[r].value, 0}}); }
            else { p_x.push_row({{one, at(index_map, l)}}); p_y.push_row({{one, at(index_map, r)}}); }
        };
        for (size_t i = 0; i < nodes.size(); i++) {
            const Node& nd = nodes[i];
            switch (nd.kind) {
                case Node::Variable:
                    p_x.push_empty_row(); p_y.push_empty_row(); p_z.push_empty_row(); p_add.push_empty_row();
                    break;
                case Node::Add: {
                    p_x.push_empty_row(); p_y.push_empty_row(); p_z.push_empty_row();
                    auto row = add_row(nd.l, nd.r);
                    row.push_back(typename SparseMatrix::Entry{minus_one, at(index_map, i)});
                    p_add.push_row(std::move(row));
                    break;
                }
                case Node::Mul:
                    p_add.push_empty_row();
                    mul_rows(nd.l, nd.r);
                    p_z.push_row({{one, at(index_map, i)}});
                    break;
                case Node::Constant:
                    if (i == 0) { p_x.push_empty_row(); p_y.push_empty_row(); p_z.push_empty_row(); p_add.push_empty_row(); }
                    break;
            }
        }
        for (size_t o : outputs) {                                       // the constraint o = 1 for each output node
            const Node& nd = nodes.at(o);
            if (nd.kind == Node::Add) {
                p_x.push_empty_row(); p_y.push_empty_row(); p_z.push_empty_row();
                auto row = add_row(nd.l, nd.r);
                row.push_back(typename SparseMatrix::Entry{minus_one, 0});
                p_add.push_row(std::move(row));
            } else if (nd.kind == Node::Mul) {
                p_add.push_empty_row();
                mul_rows(nd.l, nd.r);
                p_z.push_row({{one, 0}});
            } else {
                throw std::runtime_error("The output node must be an addition or multiplication gate");
            }
        }
        if (p_x.num_rows() > num_cols) throw std::runtime_error("attempt to subtract with overflow (more rows than m * k)");
        const size_t padding = num_cols - p_x.num_rows();
        p_x.push_empty_rows(padding); p_y.push_empty_rows(padding); p_z.push_empty_rows(padding); p_add.push_empty_rows(padding);
        // mod.rs:418-432:  upper = identity(3 m k).h_stack( -(P_x.v_stack(P_y).v_stack(P_z)) ),
        //                  lower = zero(m k, 3 m k).h_stack(P_add),   A = upper.v_stack(lower)
        // assembled in one pass into exactly-sized storage: through the stacking helpers the 2^20-constraint instance
        // (41 M rows, 46.6 M entries of 40 bytes) was copied four times, and first-touch page faults of those copies
        // were most of LigeroCircuit::new's 15 s.  Row r < 3 m k is (1, r) followed by the negated entries of its P row
        // shifted by 3 m k columns; row 3 m k + r is P_add's row r shifted likewise.
        const size_t mk = num_cols;
        SparseMatrix a(4 * mk);
        a.row_ptr.resize(4 * mk + 1);
        a.ent.resize(3 * mk + p_x.nnz() + p_y.nnz() + p_z.nnz() + p_add.nnz());
        typename SparseMatrix::Entry* w = a.ent.data();
        size_t r_out = 0;
        const SparseMatrix* upper_blocks[3] = {&p_x, &p_y, &p_z};
        for (const SparseMatrix* blk : upper_blocks)
            for (size_t r = 0; r < mk; r++, r_out++) {
                *w++ = typename SparseMatrix::Entry{one, r_out};
                for (const auto& e : blk->row(r)) *w++ = typename SparseMatrix::Entry{F::neg(e.first), e.second + 3 * mk};
                a.row_ptr[r_out + 1] = (uint64_t)(w - a.ent.data());
            }
        for (size_t r = 0; r < mk; r++, r_out++) {
            for (const auto& e : p_add.row(r)) *w++ = typename SparseMatrix::Entry{e.first, e.second + 3 * mk};
            a.row_ptr[r_out + 1] = (uint64_t)(w - a.ent.data());
        }
        return a;
    }
};
using LigeroInstance = LigeroInstanceT<Fr>;

}  // namespace ligero
