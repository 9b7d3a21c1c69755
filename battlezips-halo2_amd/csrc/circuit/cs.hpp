// Constraint-system front end: the part of halo2_proofs 0.2.0 `plonk::ConstraintSystem` (UPSTREAM, un-vendored:
// Cargo.lock:382-385) that the reference's `configure` functions drive -- src/chips/shot.rs:179-297,
// src/chips/board.rs:194-321, src/chips/bitify.rs:55-96, src/chips/placement.rs:107-265, src/chips/transpose.rs:46-88,
// src/chips/pedersen.rs:49-62 -- restated from the published API: columns, selectors, `create_gate`, `lookup`,
// `enable_equality` / `enable_constant`, query registration in call order, degree, blinding factors, and
// `compress_selectors` (keygen).  The finished system is serialised as the circuit blob `bzh_pk_create` takes
// (format "BZC2": csrc/prove.hip).  PARITY: upstream's query order / selector combination are restated from memory of
// the public crate and are unpinned by the reference (it holds no vk or proof bytes).
#pragma once
#include <algorithm>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "hostfield.hpp"

namespace bzc {

enum ColKind : uint8_t { ADVICE = 0, FIXED = 1, INSTANCE = 2 };  // upstream `Any` order: Advice < Fixed < Instance
struct Column {
    ColKind kind;
    int index;
    bool operator==(const Column& o) const { return kind == o.kind && index == o.index; }
    bool operator<(const Column& o) const { return kind != o.kind ? kind < o.kind : index < o.index; }
};
struct Selector {
    int index;
    bool simple;
};
struct TableColumn {
    Column inner;  // a fixed column
};
struct Query {
    ColKind kind;
    int column, rotation;
    bool operator==(const Query& o) const { return kind == o.kind && column == o.column && rotation == o.rotation; }
};

// ---- Expression (plonk::Expression) ----------------------------------------------------------------------------
struct ExprNode;
typedef std::shared_ptr<const ExprNode> Expr;
enum ExprTag : uint8_t { X_CONST, X_SELECTOR, X_QUERY, X_NEG, X_SUM, X_PRODUCT, X_SCALED };
struct ExprNode {
    ExprTag tag;
    Fp value;        // X_CONST, X_SCALED
    Selector sel{};  // X_SELECTOR
    Query q{};       // X_QUERY
    Expr a, b;
};
inline Expr mk(ExprTag t, Expr a = nullptr, Expr b = nullptr) {
    auto n = std::make_shared<ExprNode>();
    n->tag = t;
    n->value = Fp::zero();
    n->a = std::move(a);
    n->b = std::move(b);
    return n;
}
inline Expr constant(const Fp& v) {
    auto n = std::make_shared<ExprNode>();
    n->tag = X_CONST;
    n->value = v;
    return n;
}
inline Expr constant_u64(uint64_t v) { return constant(Fp::from_u64(v)); }
inline Expr operator-(const Expr& a) { return mk(X_NEG, a); }
inline Expr operator+(const Expr& a, const Expr& b) { return mk(X_SUM, a, b); }
inline Expr operator-(const Expr& a, const Expr& b) { return mk(X_SUM, a, mk(X_NEG, b)); }  // upstream Sub = a + (-b)
inline Expr operator*(const Expr& a, const Expr& b) { return mk(X_PRODUCT, a, b); }
inline Expr operator*(const Expr& a, const Fp& s) {  // Expression * F = Scaled
    auto n = std::make_shared<ExprNode>();
    n->tag = X_SCALED;
    n->value = s;
    n->a = a;
    return n;
}
inline Expr square(const Expr& a) { return a * a; }

inline int degree(const Expr& e) {
    switch (e->tag) {
        case X_CONST: return 0;
        case X_SELECTOR:
        case X_QUERY: return 1;
        case X_NEG:
        case X_SCALED: return degree(e->a);
        case X_SUM: return std::max(degree(e->a), degree(e->b));
        default: return degree(e->a) + degree(e->b);
    }
}
inline bool contains_simple_selector(const Expr& e) {
    switch (e->tag) {
        case X_SELECTOR: return e->sel.simple;
        case X_CONST:
        case X_QUERY: return false;
        case X_NEG:
        case X_SCALED: return contains_simple_selector(e->a);
        default: return contains_simple_selector(e->a) || contains_simple_selector(e->b);
    }
}
// Expression::extract_simple_selector: the one simple selector of the expression (-1: none); two different ones is
// an error upstream ("two simple selectors cannot be in the same expression")
inline int extract_simple_selector(const Expr& e) {
    switch (e->tag) {
        case X_SELECTOR: return e->sel.simple ? e->sel.index : -1;
        case X_CONST:
        case X_QUERY: return -1;
        case X_NEG:
        case X_SCALED: return extract_simple_selector(e->a);
        default: {
            const int l = extract_simple_selector(e->a), r = extract_simple_selector(e->b);
            if (l >= 0 && r >= 0 && l != r) throw std::logic_error("two simple selectors cannot be in the same expression");
            return l >= 0 ? l : r;
        }
    }
}
inline void collect_queries(const Expr& e, std::vector<Query>& out) {
    switch (e->tag) {
        case X_QUERY:
            if (std::find(out.begin(), out.end(), e->q) == out.end()) out.push_back(e->q);
            return;
        case X_CONST:
        case X_SELECTOR: return;
        case X_NEG:
        case X_SCALED: collect_queries(e->a, out); return;
        default: collect_queries(e->a, out); collect_queries(e->b, out);
    }
}
inline Expr substitute_selectors(const Expr& e, const std::vector<Expr>& repl) {
    switch (e->tag) {
        case X_SELECTOR: return repl[e->sel.index];
        case X_CONST:
        case X_QUERY: return e;
        case X_NEG: return mk(X_NEG, substitute_selectors(e->a, repl));
        case X_SCALED: return substitute_selectors(e->a, repl) * e->value;
        default: return mk(e->tag, substitute_selectors(e->a, repl), substitute_selectors(e->b, repl));
    }
}

struct Gate {
    std::string name;
    std::vector<std::string> constraint_names;
    std::vector<Expr> polys;
    std::vector<Query> queried_cells;  // cells queried through VirtualCells (what MockProver lists in cell_values)
    std::vector<int> queried_selectors;
};
struct Lookup {
    std::vector<Expr> inputs, tables;
};
typedef std::vector<std::pair<std::string, Expr>> Constraints;

class ConstraintSystem;
// plonk::VirtualCells: query_* registers the query in the constraint system at the moment it is made
class VirtualCells {
   public:
    explicit VirtualCells(ConstraintSystem& m) : meta(m) {}
    Expr query_selector(Selector s);
    Expr query_advice(Column c, int rot);
    Expr query_fixed(Column c, int rot);
    Expr query_instance(Column c, int rot);
    ConstraintSystem& meta;
    std::vector<Query> cells;
    std::vector<int> selectors;
};

class ConstraintSystem {
   public:
    int num_advice = 0, num_fixed = 0, num_instance = 0, num_selectors = 0;
    std::vector<bool> selector_simple;
    std::vector<Query> advice_queries, fixed_queries, instance_queries;  // registration order (= proof evaluation order)
    std::vector<int> num_advice_queries;                                 // per advice column
    std::vector<Column> permutation;                                     // equality-enabled columns, in enable order
    std::vector<Gate> gates;
    std::vector<Lookup> lookups;
    std::vector<Column> constants;  // fixed columns enabled for constants
    int minimum_degree = 0;

    Column advice_column() {
        num_advice_queries.push_back(0);
        return Column{ADVICE, num_advice++};
    }
    Column fixed_column() { return Column{FIXED, num_fixed++}; }
    Column instance_column() { return Column{INSTANCE, num_instance++}; }
    Selector selector() {
        selector_simple.push_back(true);
        return Selector{num_selectors++, true};
    }
    Selector complex_selector() {
        selector_simple.push_back(false);
        return Selector{num_selectors++, false};
    }
    TableColumn lookup_table_column() { return TableColumn{fixed_column()}; }

    int query_index(Column c, int rot) {
        std::vector<Query>& qs = c.kind == ADVICE ? advice_queries : (c.kind == FIXED ? fixed_queries : instance_queries);
        const Query q{c.kind, c.index, rot};
        for (size_t i = 0; i < qs.size(); i++) {
            if (qs[i] == q) return (int)i;
        }
        qs.push_back(q);
        if (c.kind == ADVICE) num_advice_queries[c.index]++;
        return (int)qs.size() - 1;
    }
    // enable_equality registers the column's current-row query at once (query_any_index), then adds the column to
    // the permutation argument
    void enable_equality(Column c) {
        query_index(c, 0);
        if (std::find(permutation.begin(), permutation.end(), c) == permutation.end()) permutation.push_back(c);
    }
    void enable_constant(Column fixed) {
        if (std::find(constants.begin(), constants.end(), fixed) == constants.end()) {
            constants.push_back(fixed);
            enable_equality(fixed);
        }
    }
    void create_gate(const std::string& name, const std::function<Constraints(VirtualCells&)>& f) {
        VirtualCells vc(*this);
        Constraints cons = f(vc);
        if (cons.empty()) throw std::logic_error("Gates must contain at least one constraint.");
        Gate g;
        g.name = name;
        for (auto& c : cons) {
            g.constraint_names.push_back(c.first);
            g.polys.push_back(c.second);
        }
        g.queried_cells = vc.cells;
        g.queried_selectors = vc.selectors;
        gates.push_back(std::move(g));
    }
    // lookup: inputs must not contain simple selectors; every table column is queried at the current row when the
    // closure's pairs are mapped (after all input queries)
    size_t lookup(const std::function<std::vector<std::pair<Expr, TableColumn>>(VirtualCells&)>& f) {
        VirtualCells vc(*this);
        auto pairs = f(vc);
        Lookup lk;
        for (auto& pr : pairs) {
            if (contains_simple_selector(pr.first)) throw std::logic_error("expression containing simple selector supplied to lookup argument");
            lk.inputs.push_back(pr.first);
            lk.tables.push_back(vc.query_fixed(pr.second.inner, 0));
        }
        lookups.push_back(std::move(lk));
        return lookups.size() - 1;
    }

    // ConstraintSystem::degree (selectors count as degree 1: it is taken before selector compression)
    int degree() const {
        int d = permutation.empty() ? 1 : 3;  // permutation::Argument::required_degree
        for (auto& lk : lookups) {
            int di = 1, dt = 1;
            for (auto& e : lk.inputs) di = std::max(di, bzc::degree(e));
            for (auto& e : lk.tables) dt = std::max(dt, bzc::degree(e));
            d = std::max(d, std::max(4, 2 + di + dt));
        }
        for (auto& g : gates) {
            for (auto& p : g.polys) d = std::max(d, bzc::degree(p));
        }
        return std::max(d, minimum_degree);
    }
    int blinding_factors() const {
        int factors = 1;
        for (int c : num_advice_queries) factors = std::max(factors, c);
        factors = std::max(3, factors);
        return factors + 1 /* multiopen evaluation at x_3 */ + 1 /* off-by-one slack */;
    }
    int minimum_rows() const { return blinding_factors() + 1 + 1 + 1 + 1; }

    // compress_selectors (plonk/circuit/compress_selectors.rs): simple selectors that are never enabled on the same
    // row share a fixed column, selector i of a combination of size L being q * prod_{j in 1..=L, j != i} (j - q);
    // complex or unused selectors get a column of their own.  Returns the new fixed columns' values (appended to
    // num_fixed in order) and rewrites gates / lookups.
    std::vector<std::vector<Fp>> compress_selectors(const std::vector<std::vector<bool>>& activations) {
        if ((int)activations.size() != num_selectors) throw std::logic_error("selector count");
        const size_t n = activations.empty() ? 0 : activations[0].size();
        std::vector<int> degrees(num_selectors, 0);
        for (auto& g : gates) {
            for (auto& p : g.polys) {
                const int s = extract_simple_selector(p);
                if (s >= 0) degrees[s] = std::max(degrees[s], bzc::degree(p));
            }
        }
        const int max_degree = degree();
        std::vector<std::vector<Fp>> polys;
        std::vector<Expr> repl(num_selectors);
        auto allocate = [&]() -> Expr {
            const Column c = fixed_column();
            query_index(c, 0);
            auto nd = std::make_shared<ExprNode>();
            nd->tag = X_QUERY;
            nd->q = Query{FIXED, c.index, 0};
            return nd;
        };
        std::vector<int> simple;  // remaining (degree > 0) selectors, in index order
        for (int s = 0; s < num_selectors; s++) {
            if (degrees[s] == 0) {
                repl[s] = allocate();
                std::vector<Fp> col(n, Fp::zero());
                for (size_t r = 0; r < n; r++) {
                    if (activations[s][r]) col[r] = Fp::one();
                }
                polys.push_back(std::move(col));
            } else {
                simple.push_back(s);
            }
        }
        const size_t m = simple.size();
        std::vector<std::vector<bool>> excl(m);
        for (size_t i = 0; i < m; i++) {
            excl[i].assign(i, false);
            for (size_t j = 0; j < i; j++) {
                const auto &a = activations[simple[i]], &b = activations[simple[j]];
                for (size_t r = 0; r < n; r++) {
                    if (a[r] && b[r]) {
                        excl[i][j] = true;
                        break;
                    }
                }
            }
        }
        std::vector<bool> added(m, false);
        for (size_t i = 0; i < m; i++) {
            if (added[i]) continue;
            added[i] = true;
            if (degrees[simple[i]] > max_degree) throw std::logic_error("selector degree above the system's");
            int d = degrees[simple[i]] - 1;
            std::vector<size_t> comb{i};
            for (size_t j = i + 1; j < m; j++) {
                if (d + (int)comb.size() == max_degree) break;
                if (added[j]) continue;
                bool clash = false;
                for (size_t c : comb) {
                    if (excl[j][c]) {
                        clash = true;
                        break;
                    }
                }
                if (clash) continue;
                const int new_d = std::max(d, degrees[simple[j]] - 1);
                if (new_d + (int)comb.size() + 1 > max_degree) continue;
                d = new_d;
                comb.push_back(j);
                added[j] = true;
            }
            const Expr query = allocate();
            std::vector<Fp> col(n, Fp::zero());
            Fp assigned_root = Fp::one();
            for (size_t c : comb) {
                Expr e = query;
                Fp root = Fp::one();
                for (size_t t = 0; t < comb.size(); t++) {
                    if (root != assigned_root) e = e * (constant(root) - query);
                    root = root + Fp::one();
                }
                const auto& act = activations[simple[c]];
                for (size_t r = 0; r < n; r++) {
                    if (act[r]) col[r] = assigned_root;
                }
                repl[simple[c]] = e;
                assigned_root = assigned_root + Fp::one();
            }
            polys.push_back(std::move(col));
        }
        for (auto& g : gates) {
            for (auto& p : g.polys) p = substitute_selectors(p, repl);
        }
        for (auto& lk : lookups) {
            for (auto& e : lk.inputs) e = substitute_selectors(e, repl);
            for (auto& e : lk.tables) e = substitute_selectors(e, repl);
        }
        compressed = true;
        return polys;
    }
    bool compressed = false;
};

inline Expr VirtualCells::query_selector(Selector s) {
    selectors.push_back(s.index);
    auto n = std::make_shared<ExprNode>();
    n->tag = X_SELECTOR;
    n->sel = s;
    return n;
}
inline Expr VirtualCells::query_advice(Column c, int rot) {
    const Query q{ADVICE, c.index, rot};
    cells.push_back(q);
    meta.query_index(c, rot);
    auto n = std::make_shared<ExprNode>();
    n->tag = X_QUERY;
    n->q = q;
    return n;
}
inline Expr VirtualCells::query_fixed(Column c, int rot) {
    const Query q{FIXED, c.index, rot};
    cells.push_back(q);
    meta.query_index(c, rot);
    auto n = std::make_shared<ExprNode>();
    n->tag = X_QUERY;
    n->q = q;
    return n;
}
inline Expr VirtualCells::query_instance(Column c, int rot) {
    const Query q{INSTANCE, c.index, rot};
    cells.push_back(q);
    meta.query_index(c, rot);
    auto n = std::make_shared<ExprNode>();
    n->tag = X_QUERY;
    n->q = q;
    return n;
}

// Constraints::with_selector: every constraint multiplied by the selector (selector * constraint)
inline Constraints with_selector(const Expr& sel, Constraints cons) {
    for (auto& c : cons) c.second = sel * c.second;
    return cons;
}
// halo2_gadgets utilities: bool_check(v) = v (1 - v);  range_check(w, range) = w (1 - w) (2 - w) ... (range-1 - w)
inline Expr bool_check(const Expr& v) { return v * (constant(Fp::one()) - v); }  // range_check(value, 2)
inline Expr range_check(const Expr& word, uint64_t range) {
    Expr acc = word;
    for (uint64_t i = 1; i < range; i++) acc = acc * (constant_u64(i) - word);
    return acc;
}

// ---- blob writer ("BZC2", csrc/prove.hip) -------------------------------------------------------------------------
struct BlobWriter {
    std::vector<uint8_t> b;
    void u8(uint8_t v) { b.push_back(v); }
    void u32(uint32_t v) {
        for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i)));
    }
    void fe(const Fp& v) {
        uint8_t r[32];
        v.to_repr(r);
        b.insert(b.end(), r, r + 32);
    }
    void expr(const Expr& e) {
        switch (e->tag) {
            case X_CONST: u8(0); fe(e->value); break;
            case X_QUERY:
                u8(e->q.kind == ADVICE ? 1 : (e->q.kind == FIXED ? 2 : 3));
                u32((uint32_t)e->q.column);
                u32((uint32_t)e->q.rotation);
                break;
            case X_NEG: u8(4); expr(e->a); break;
            case X_SUM: u8(5); expr(e->a); expr(e->b); break;
            case X_PRODUCT: u8(6); expr(e->a); expr(e->b); break;
            case X_SCALED: u8(7); expr(e->a); fe(e->value); break;
            case X_SELECTOR: throw std::logic_error("selectors must be compressed before serialisation");
        }
    }
};

}  // namespace bzc
