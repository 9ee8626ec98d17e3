// Expression front end of the circuit builder (src/expression/mod.rs): variables and constants combined with + - *
// into a shared DAG, then numbered into an ArithmeticCircuit.  The reference's own prove/verify tests build their
// circuits both ways (src/ligero/tests.rs:172-243), and its expression tests pin the exact node numbering
// (src/expression/tests.rs:60-70, 214-345), which test_expression.cpp restates.
//
//   Expression::{constant, variable}                    mod.rs:51-57
//   to_arithmetic_circuit / update_map                  mod.rs:59-141
//   scalar_product / sparse_scalar_product / pow        mod.rs:143-176
//   Neg, Add, Mul, Sub (+ constants on either side)     mod.rs:185-236
//   +=, -=, *=, Sum, Product                            mod.rs:238-270
//
// Identity is by node, as there (Rc pointers): a sub-expression used twice is one node of the circuit; two
// Expression::variable("x") calls are two Variable nodes with the same label (the label map keeps the higher index).
#pragma once
#include <memory>
#include <unordered_map>

#include "circuit.hpp"

namespace ligero {

template <class E>
class ExpressionT {
    using F = Field<E>;
    using Node = NodeT<E>;
    struct Inner {
        typename Node::Kind kind;
        std::string label;                       // Variable
        E value{};                               // Constant
        std::shared_ptr<const Inner> a, b;       // Add / Mul
        // operands are released iteratively: a chain of a million additions must not recurse a million destructors deep
        ~Inner() {
            std::vector<std::shared_ptr<const Inner>> pending;
            pending.push_back(std::move(a));
            pending.push_back(std::move(b));
            while (!pending.empty()) {
                std::shared_ptr<const Inner> p = std::move(pending.back());
                pending.pop_back();
                if (p && p.use_count() == 1) {
                    Inner* m = const_cast<Inner*>(p.get());
                    pending.push_back(std::move(m->a));
                    pending.push_back(std::move(m->b));
                }
            }
        }
    };
    std::shared_ptr<const Inner> p_;
    explicit ExpressionT(std::shared_ptr<const Inner> p) : p_(std::move(p)) {}
    static ExpressionT gate(typename Node::Kind k, const ExpressionT& a, const ExpressionT& b) {
        auto in = std::make_shared<Inner>();
        in->kind = k;
        in->a = a.p_;
        in->b = b.p_;
        return ExpressionT(std::move(in));
    }

public:
    static ExpressionT constant(const E& value) {                       // mod.rs:51-53
        auto in = std::make_shared<Inner>();
        in->kind = Node::Constant;
        in->value = value;
        return ExpressionT(std::move(in));
    }
    static ExpressionT variable(const std::string& label) {             // mod.rs:55-57
        auto in = std::make_shared<Inner>();
        in->kind = Node::Variable;
        in->label = label;
        return ExpressionT(std::move(in));
    }
    // F::from(i32) of the integer-on-the-left operators (mod.rs:212-220)
    static E field_from_int(long v) { return v < 0 ? F::neg(F::from_u64((uint64_t)(-v))) : F::from_u64((uint64_t)v); }

    const void* pointer() const { return p_.get(); }                    // mod.rs:109-111

    // mod.rs:59-107.  Nodes are discovered parent first, then the left operand's whole sub-DAG, then the right one's
    // (update_map, mod.rs:113-141), and numbered in the REVERSE of that order: the root is the last node, and a gate may
    // refer to nodes after it.  Duplicate constants are then dropped (filter_constants).
    ArithmeticCircuitT<E> to_arithmetic_circuit() const {
        std::unordered_map<const Inner*, size_t> found;                 // node -> discovery index
        std::vector<const Inner*> order;
        std::vector<const Inner*> stack{p_.get()};
        while (!stack.empty()) {
            const Inner* x = stack.back();
            stack.pop_back();
            if (found.count(x)) continue;
            found.emplace(x, order.size());
            order.push_back(x);
            if (x->kind == Node::Add || x->kind == Node::Mul) {
                stack.push_back(x->b.get());                            // popped after everything under a
                stack.push_back(x->a.get());
            }
        }
        const size_t n = order.size();
        std::vector<Node> nodes(n);
        for (size_t d = 0; d < n; d++) {
            const Inner* x = order[d];
            Node& nd = nodes[n - 1 - d];
            nd.kind = x->kind;
            nd.label = x->label;
            nd.value = x->value;
            if (x->kind == Node::Add || x->kind == Node::Mul) {
                nd.l = n - 1 - found.at(x->a.get());
                nd.r = n - 1 - found.at(x->b.get());
            }
        }
        auto filtered = ArithmeticCircuitT<E>::filter_constants(nodes);
        ArithmeticCircuitT<E> c;
        c.nodes = std::move(filtered.first);
        c.constants = std::move(filtered.second);
        for (size_t i = 0; i < c.nodes.size(); i++)
            if (c.nodes[i].kind == Node::Variable) c.variables[c.nodes[i].label] = i;
        return c;
    }

    static ExpressionT sum(const std::vector<ExpressionT>& v) {         // Sum, mod.rs:260-264 (reduce().unwrap())
        if (v.empty()) throw std::runtime_error("called `Option::unwrap()` on a `None` value: sum of no expressions");
        ExpressionT acc = v[0];
        for (size_t i = 1; i < v.size(); i++) acc = acc + v[i];
        return acc;
    }
    static ExpressionT product(const std::vector<ExpressionT>& v) {     // Product, mod.rs:266-270
        if (v.empty()) throw std::runtime_error("called `Option::unwrap()` on a `None` value: product of no expressions");
        ExpressionT acc = v[0];
        for (size_t i = 1; i < v.size(); i++) acc = acc * v[i];
        return acc;
    }
    static ExpressionT scalar_product(const std::vector<ExpressionT>& a, const std::vector<ExpressionT>& b) {   // mod.rs:143-145
        std::vector<ExpressionT> terms;
        for (size_t i = 0; i < a.size() && i < b.size(); i++) terms.push_back(a[i] * b[i]);
        return sum(terms);
    }
    // mod.rs:147-153: sum of b[i] * a_i over the (value, index) entries of a sparse row
    static ExpressionT sparse_scalar_product(const std::vector<std::pair<E, size_t>>& a, const std::vector<ExpressionT>& b) {
        std::vector<ExpressionT> terms;
        for (const auto& e : a) terms.push_back(b.at(e.second) * e.first);
        return sum(terms);
    }
    ExpressionT pow(size_t rhs) const {                                 // mod.rs:155-176: x^0 is x there, not 1
        if (rhs == 0) return *this;
        int top = 63;
        while (!((rhs >> top) & 1)) top--;
        ExpressionT cur = *this;
        for (int b = top - 1; b >= 0; b--) {
            cur = cur * cur;
            if ((rhs >> b) & 1) cur = cur * *this;
        }
        return cur;
    }

    friend ExpressionT operator+(const ExpressionT& a, const ExpressionT& b) { return gate(Node::Add, a, b); }   // mod.rs:193-199
    friend ExpressionT operator*(const ExpressionT& a, const ExpressionT& b) { return gate(Node::Mul, a, b); }   // mod.rs:201-207
    friend ExpressionT operator-(const ExpressionT& a) { return constant(F::neg(F::one())) * a; }                // mod.rs:185-191
    friend ExpressionT operator-(const ExpressionT& a, const ExpressionT& b) { return a + (-b); }                // mod.rs:209-215
    // constants: a field element on the right, an integer on the left (mod.rs:217-236); C++ lets both sit on either side
    friend ExpressionT operator+(const ExpressionT& a, const E& v) { return a + constant(v); }
    friend ExpressionT operator*(const ExpressionT& a, const E& v) { return a * constant(v); }
    friend ExpressionT operator-(const ExpressionT& a, const E& v) { return a - constant(v); }
    friend ExpressionT operator+(long v, const ExpressionT& a) { return constant(field_from_int(v)) + a; }
    friend ExpressionT operator*(long v, const ExpressionT& a) { return constant(field_from_int(v)) * a; }
    friend ExpressionT operator-(long v, const ExpressionT& a) { return constant(field_from_int(v)) - a; }
    ExpressionT& operator+=(const ExpressionT& b) { return *this = *this + b; }                                  // mod.rs:238-258
    ExpressionT& operator*=(const ExpressionT& b) { return *this = *this * b; }
    ExpressionT& operator-=(const ExpressionT& b) { return *this = *this - b; }
    ExpressionT& operator+=(const E& v) { return *this = *this + v; }
    ExpressionT& operator*=(const E& v) { return *this = *this * v; }
    ExpressionT& operator-=(const E& v) { return *this = *this - v; }
};
using Expression = ExpressionT<Fr>;

}  // namespace ligero
