// Circuit assignment back end + the single-pass floor planner the reference's circuits name
// (`type FloorPlanner = SimpleFloorPlanner`, src/circuits/shot.rs:24, src/circuits/board.rs:23).
//
// Restates halo2_proofs 0.2.0 (UPSTREAM, un-vendored) `circuit::floor_planner::single_pass::SingleChipLayouter` and the
// `plonk::Assignment` sinks it drives (keygen `Assembly`, prover `WitnessCollection`, dev `MockProver` region
// bookkeeping) from the published design: every `assign_region` closure runs twice (shape pass, then assignment), a
// region starts at the first row where none of its columns -- selectors count as columns -- is in use, constants are
// placed one per row in the first constants column after each region and tied to their advice cells by copy
// constraints, `assign_table` fills the rest of a table column with its first value.
#pragma once
#include <map>
#include <set>
#include <string>
#include <vector>

#include "cs.hpp"

namespace bzc {

struct SynthesisError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct RegionInfo {  // what dev::MockProver keeps per region (failure locations are reported against these)
    std::string name;
    std::set<Column> columns;
    bool has_rows = false;
    size_t row_lo = 0, row_hi = 0;
    std::vector<std::pair<int, size_t>> advice_cells;   // (advice column, absolute row) of every assign_advice, in order
    void extend(Column c, size_t row) {
        columns.insert(c);
        if (!has_rows) {
            row_lo = row_hi = row;
            has_rows = true;
        } else {
            row_lo = std::min(row_lo, row);
            row_hi = std::max(row_hi, row);
        }
    }
};
struct CopyConstraint {
    Column a;
    size_t row_a;
    Column b;
    size_t row_b;
};

// plonk::Assignment sink.  `keep_fixed`: keygen (fixed cells, selector activations, copy constraints, regions);
// `keep_advice`: witness generation (advice cells).  Both for MockProver-style checking.
struct Assembly {
    unsigned k = 0;
    size_t n = 0, usable_rows = 0;
    bool keep_fixed = true, keep_advice = true;
    std::vector<std::vector<Fp>> fixed;            // [num_fixed][n]
    std::vector<std::vector<bool>> selectors;      // [num_selectors][n]
    std::vector<CopyConstraint> copies;
    std::vector<RegionInfo> regions;
    int current_region = -1;
    // advice either as owned columns or written straight into a caller's [num_advice][stride] block (Montgomery limbs)
    std::vector<std::vector<Fp>> advice;
    Fp* advice_out = nullptr;
    size_t advice_stride = 0;  // rows per column in advice_out (only rows < stride are ever written)

    void init(const ConstraintSystem& cs, unsigned k_, bool fixed_, bool advice_, Fp* out = nullptr, size_t stride = 0) {
        k = k_;
        n = (size_t)1 << k;
        usable_rows = n - (size_t)(cs.blinding_factors() + 1);
        keep_fixed = fixed_;
        keep_advice = advice_;
        if (keep_fixed) {
            fixed.assign(cs.num_fixed, std::vector<Fp>(n, Fp::zero()));
            selectors.assign(cs.num_selectors, std::vector<bool>(n, false));
        }
        advice_out = out;
        advice_stride = stride;
        if (keep_advice && !out) advice.assign(cs.num_advice, std::vector<Fp>(n, Fp::zero()));
    }
    void check_row(size_t row) const {
        if (row >= usable_rows) throw SynthesisError("not enough rows available");
    }
    void enter_region(const std::string& name) {
        if (keep_fixed) {
            regions.push_back(RegionInfo{name, {}, false, 0, 0, {}});
            current_region = (int)regions.size() - 1;
        }
    }
    void exit_region() { current_region = -1; }
    void enable_selector(Selector s, size_t row) {
        check_row(row);
        if (keep_fixed) selectors[s.index][row] = true;
    }
    void assign_advice(Column c, size_t row, const Fp& v) {
        check_row(row);
        if (keep_fixed && current_region >= 0) {
            regions[current_region].extend(c, row);
            regions[current_region].advice_cells.push_back({c.index, row});
        }
        if (!keep_advice) return;
        if (advice_out) {
            if (row >= advice_stride) throw SynthesisError("advice row beyond the compact stride");
            advice_out[(size_t)c.index * advice_stride + row] = v;
        } else {
            advice[c.index][row] = v;
        }
    }
    void assign_fixed(Column c, size_t row, const Fp& v) {
        check_row(row);
        if (!keep_fixed) return;
        if (current_region >= 0) regions[current_region].extend(c, row);
        fixed[c.index][row] = v;
    }
    void copy(Column a, size_t row_a, Column b, size_t row_b) {
        check_row(row_a);
        check_row(row_b);
        if (keep_fixed) copies.push_back(CopyConstraint{a, row_a, b, row_b});
    }
    void fill_from_row(Column c, size_t from_row, const Fp& v) {
        if (!keep_fixed) return;
        for (size_t r = from_row; r < usable_rows; r++) fixed[c.index][r] = v;
    }
};

struct Cell {
    int region_index;
    size_t row_offset;
    Column column;
};
struct AssignedCell {
    Cell cell;
    Fp value;
};

struct RegionColumn {  // RegionColumn::{Column, Selector}
    int kind;          // 0..2 ColKind, 3 selector
    int index;
    bool operator<(const RegionColumn& o) const { return kind != o.kind ? kind < o.kind : index < o.index; }
};

class Layouter;
class Region {
   public:
    Region(Layouter& l, int index, bool shape) : lay(l), region_index(index), shape_pass(shape) {}
    void enable_selector(Selector s, size_t offset);
    AssignedCell assign_advice(Column c, size_t offset, const Fp& v);
    AssignedCell assign_advice_from_constant(Column c, size_t offset, const Fp& constant);
    void assign_fixed(Column c, size_t offset, const Fp& v);
    AssignedCell copy_advice(const AssignedCell& from, Column c, size_t offset);
    void constrain_equal(const Cell& a, const Cell& b);
    void constrain_constant(const Cell& c, const Fp& constant);

    Layouter& lay;
    int region_index;
    bool shape_pass;
    // shape pass
    std::set<RegionColumn> columns;
    size_t row_count = 0;
    // assignment pass
    std::vector<std::pair<Fp, Cell>> constants;
};

class Table {  // SimpleTableLayouter
   public:
    explicit Table(Layouter& l) : lay(l) {}
    void assign_cell(TableColumn c, size_t offset, const Fp& v);
    Layouter& lay;
    std::map<int, std::pair<bool, Fp>> defaults;          // fixed column index -> (seen, default = value at offset 0)
    std::map<int, std::vector<bool>> assigned;
};

class Layouter {
   public:
    Layouter(Assembly& a, const std::vector<Column>& constants_cols) : cs(a), constants(constants_cols) {}
    // With `known_starts` (region start rows recorded by an earlier synthesis of the same circuit) the shape pass is
    // skipped: witness generation re-runs only the assignment pass.
    const std::vector<size_t>* known_starts = nullptr;

    template <class Fn>
    auto assign_region(const std::string& name, Fn&& fn) {
        const int region_index = (int)regions.size();
        size_t region_start = 0;
        if (known_starts) {
            if ((size_t)region_index >= known_starts->size()) throw SynthesisError("region count differs from the recorded layout");
            region_start = (*known_starts)[region_index];
            regions.push_back(region_start);
        } else {
            Region shape(*this, region_index, true);
            fn(shape);
            for (auto& c : shape.columns) {
                auto it = columns.find(c);
                if (it != columns.end()) region_start = std::max(region_start, it->second);
            }
            regions.push_back(region_start);
            for (auto& c : shape.columns) columns[c] = region_start + shape.row_count;
        }
        cs.enter_region(name);
        Region region(*this, region_index, false);
        auto result = fn(region);
        cs.exit_region();
        if (!region.constants.empty()) {
            if (constants.empty()) throw SynthesisError("not enough columns for constants");
            const Column cc = constants[0];
            size_t& next = columns[RegionColumn{(int)cc.kind, cc.index}];
            for (auto& kv : region.constants) {
                cs.assign_fixed(cc, next, kv.first);
                cs.copy(cc, next, kv.second.column, regions[kv.second.region_index] + kv.second.row_offset);
                next++;
            }
        }
        return result;
    }
    template <class Fn>
    void assign_table(const std::string& name, Fn&& fn) {
        cs.enter_region(name);
        Table t(*this);
        fn(t);
        cs.exit_region();
        size_t first_unused = 0;
        bool have = false;
        for (auto& kv : t.assigned) {
            size_t len = kv.second.size();
            for (bool b : kv.second) {
                if (!b) throw SynthesisError("table column has an unassigned row");
            }
            if (have && len != first_unused) throw SynthesisError("table columns of different lengths");
            first_unused = len;
            have = true;
        }
        for (auto& kv : t.defaults) cs.fill_from_row(Column{FIXED, kv.first}, first_unused, kv.second.second);
    }
    void constrain_instance(const Cell& cell, Column instance, size_t row) {
        cs.copy(cell.column, regions[cell.region_index] + cell.row_offset, instance, row);
    }
    size_t absolute_row(const Cell& c) const { return regions[c.region_index] + c.row_offset; }

    Assembly& cs;
    std::vector<Column> constants;
    std::vector<size_t> regions;                 // start row per region
    std::map<RegionColumn, size_t> columns;      // first unused row per column / selector
};

inline void Region::enable_selector(Selector s, size_t offset) {
    if (shape_pass) {
        columns.insert(RegionColumn{3, s.index});
        row_count = std::max(row_count, offset + 1);
        return;
    }
    lay.cs.enable_selector(s, lay.regions[region_index] + offset);
}
inline AssignedCell Region::assign_advice(Column c, size_t offset, const Fp& v) {
    if (shape_pass) {
        columns.insert(RegionColumn{(int)c.kind, c.index});
        row_count = std::max(row_count, offset + 1);
    } else {
        lay.cs.assign_advice(c, lay.regions[region_index] + offset, v);
    }
    return AssignedCell{Cell{region_index, offset, c}, v};
}
inline AssignedCell Region::assign_advice_from_constant(Column c, size_t offset, const Fp& constant) {
    AssignedCell cell = assign_advice(c, offset, constant);
    constrain_constant(cell.cell, constant);
    return cell;
}
inline void Region::assign_fixed(Column c, size_t offset, const Fp& v) {
    if (shape_pass) {
        columns.insert(RegionColumn{(int)c.kind, c.index});
        row_count = std::max(row_count, offset + 1);
        return;
    }
    lay.cs.assign_fixed(c, lay.regions[region_index] + offset, v);
}
inline AssignedCell Region::copy_advice(const AssignedCell& from, Column c, size_t offset) {
    AssignedCell cell = assign_advice(c, offset, from.value);
    constrain_equal(cell.cell, from.cell);
    return cell;
}
inline void Region::constrain_equal(const Cell& a, const Cell& b) {
    if (shape_pass) return;
    lay.cs.copy(a.column, lay.regions[a.region_index] + a.row_offset, b.column, lay.regions[b.region_index] + b.row_offset);
}
inline void Region::constrain_constant(const Cell& c, const Fp& constant) {
    if (shape_pass) return;
    constants.push_back({constant, c});
}
inline void Table::assign_cell(TableColumn c, size_t offset, const Fp& v) {
    const int idx = c.inner.index;
    auto& d = defaults[idx];
    if (offset == 0) {
        if (d.first) throw SynthesisError("table default assigned twice");
        d = {true, v};
    }
    auto& as = assigned[idx];
    if (as.size() <= offset) as.resize(offset + 1, false);
    as[offset] = true;
    lay.cs.assign_fixed(c.inner, offset, v);
}

}  // namespace bzc
