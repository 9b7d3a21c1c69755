"""CPU ORACLE (test infrastructure, NOT product code): restatement of halo2_proofs 0.2.0 `dev::MockProver::verify`
(UPSTREAM, un-vendored: Cargo.lock:382-385) over a circuit given as data -- the checker behind the reference's 35
circuit tests (src/circuits/board.rs:98-877, src/circuits/shot.rs:99-878, src/chips/bitify.rs:405-531).

Given the decoded circuit blob (oracle/blob.py), the circuit description the product exports (gate / constraint
names, cells queried per gate, regions) and one witness (advice table + instance column), it reports failures in
upstream's order and vocabulary:
  ConstraintNotSatisfied {(gate index, name), (constraint index, name), region (index, name) + offset, cell values}
  Lookup                 {lookup index, region + offset}
  Permutation            {column, region + offset | outside-region row}
gates by index, then row, then constraint; then lookups; then the permutation columns in their order, by row.
Values print like upstream's `util::format_value`: "0", "1", "-1", else hex without leading zeros.
Independent of the product: big-int evaluation of the blob's polynomials, own permutation-cycle construction
(halo2_oracle.build_permutation)."""
from __future__ import annotations

import halo2_oracle as H

P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
KIND_ORDER = {'advice': 0, 'fixed': 1, 'instance': 2}


def format_value(v: int, p: int = P) -> str:
    v %= p
    if v == 0:
        return "0"
    if v == 1:
        return "1"
    if v == p - 1:
        return "-1"
    return "0x%x" % v


def _find_region(regions, row, columns):
    """FailureLocation::find: first region whose rows contain `row` and whose columns meet `columns`."""
    for i, r in enumerate(regions):
        if not r["has_rows"] or not (r["row_lo"] <= row <= r["row_hi"]):
            continue
        rc = {(k, c) for k, c in r["columns"]}
        if rc & columns:
            return {"region": [i, r["name"]], "offset": row - r["row_lo"]}
    return {"outside_row": row}


def _columns_of(e, out):
    t = e[0]
    if t in KIND_ORDER:
        out.add((t, e[1]))
    elif t in ('neg', 'scale'):
        _columns_of(e[1], out)
    elif t in ('add', 'mul'):
        _columns_of(e[1], out)
        _columns_of(e[2], out)


def _cells_of(e, out):
    t = e[0]
    if t in KIND_ORDER:
        out.add((t, e[1], e[2]))
    elif t in ('neg', 'scale'):
        _cells_of(e[1], out)
    elif t in ('add', 'mul'):
        _cells_of(e[1], out)
        _cells_of(e[2], out)


def _leading_fixed_factor(e):
    """(column) of a fixed current-row query that multiplies the whole polynomial (reached from the root through
    products / negations / scalings only), else None.  Every gate of the reference is `selector * constraint`, and a
    compressed selector is q * prod (j - q): on rows where q = 0 the polynomial vanishes and need not be evaluated."""
    while True:
        t = e[0]
        if t == 'fixed':
            return e[1] if e[2] == 0 else None
        if t in ('mul', 'neg', 'scale'):
            e = e[1]
            continue
        return None


def verify(circ, desc, advice, instance, p: int = P):
    """circ: blob.DecodedCircuit; desc: the product's circuit description (dict); advice: num_advice columns of n ints;
    instance: num_instance columns (shorter columns are zero-padded).  Returns [] or the failure list."""
    n, usable = circ.n, desc["usable_rows"]
    inst = [list(c) + [0] * (n - len(c)) for c in instance]
    while len(inst) < circ.num_instance:
        inst.append([0] * n)
    cols = {'advice': advice, 'fixed': circ.fixed, 'instance': inst}
    failures = []

    # gates
    for gi, g in enumerate(desc["gates"]):
        polys = circ.gates[g["first_poly"]:g["first_poly"] + len(g["constraints"])]
        gate_columns = set()
        for pl in polys:
            _columns_of(pl, gate_columns)
        queried = {(k, c, r) for k, c, r in g["queried_cells"]}
        lead = [_leading_fixed_factor(pl) for pl in polys]
        rows = range(usable)
        if all(c is not None for c in lead):
            rows = sorted({r for c in set(lead) for r in range(usable) if circ.fixed[c][r] % p})
        for row in rows:
            leaf = lambda t, c, r, row=row: cols[t][c][(row + r) % n]
            for ci, pl in enumerate(polys):
                if lead[ci] is not None and circ.fixed[lead[ci]][row] % p == 0:
                    continue
                if H.expr_eval(pl, leaf, p) == 0:
                    continue
                cells = set()
                _cells_of(pl, cells)
                shown = sorted((c for c in cells if c in queried), key=lambda c: (KIND_ORDER[c[0]], c[1], c[2]))
                f = {"type": "ConstraintNotSatisfied", "gate": [gi, g["name"]], "constraint": [ci, g["constraints"][ci]],
                     "cell_values": [[k, c, r, format_value(leaf(k, c, r), p)] for k, c, r in shown]}
                f.update(_find_region(desc["regions"], row, gate_columns))
                failures.append(f)

    # lookups: every usable row's input tuple must be a row of the table
    for li, (ins, tabs) in enumerate(circ.lookups):
        table = set()
        for row in range(usable):
            leaf = lambda t, c, r, row=row: cols[t][c][(row + r) % n]
            table.add(tuple(H.expr_eval(e, leaf, p) for e in tabs))
        in_columns = set()
        for e in ins:
            _columns_of(e, in_columns)
        for row in range(usable):
            leaf = lambda t, c, r, row=row: cols[t][c][(row + r) % n]
            if tuple(H.expr_eval(e, leaf, p) for e in ins) not in table:
                f = {"type": "Lookup", "lookup_index": li}
                f.update(_find_region(desc["regions"], row, in_columns))
                failures.append(f)

    # permutation: every cell equals the cell its cycle maps it to
    class _CS:
        pass
    cs = _CS()
    cs.perm_columns, cs.n = circ.perm_columns, n
    mapping = H.build_permutation(cs, circ.copies)
    for ci, (kind, idx) in enumerate(circ.perm_columns):
        for row in range(n):
            mc, mr = mapping[ci][row]
            if (mc, mr) == (ci, row):
                continue
            mk, mi = circ.perm_columns[mc]
            if cols[kind][idx][row] % p != cols[mk][mi][mr] % p:
                f = {"type": "Permutation", "column": [kind, idx]}
                f.update(_find_region(desc["regions"], row, {(kind, idx)}))
                failures.append(f)
    return failures


class LocalChecker:
    """What MockProver::verify would newly report after ONE advice cell of an otherwise satisfying witness is changed,
    without re-walking the whole table: only constraints that can see the cell are evaluated -- gate polynomials that
    query (column, rotation r) at row - r, lookup inputs likewise, and the cell's permutation cycle.  Used by the mutation
    test of the restated halo2_gadgets gates (tests/test_ecc_gate_mutation_cpu.py): a cell no constraint notices is a
    dropped or mis-stated polynomial."""

    def __init__(self, circ, desc, advice, instance, p: int = P):
        self.circ, self.desc, self.p = circ, desc, p
        self.n, self.usable = circ.n, desc["usable_rows"]
        inst = [list(c) + [0] * (self.n - len(c)) for c in instance]
        while len(inst) < circ.num_instance:
            inst.append([0] * self.n)
        self.cols = {'advice': [list(c) for c in advice], 'fixed': circ.fixed, 'instance': inst}
        # per advice column: [(gate index, constraint index, polynomial, leading fixed column or None, rotation)]
        self.by_column = {}
        for gi, g in enumerate(desc["gates"]):
            polys = circ.gates[g["first_poly"]:g["first_poly"] + len(g["constraints"])]
            for ci, pl in enumerate(polys):
                cells = set()
                _cells_of(pl, cells)
                lead = _leading_fixed_factor(pl)
                for t, c, r in cells:
                    if t == 'advice':
                        self.by_column.setdefault(c, []).append((gi, ci, pl, lead, r))
        self.lookup_by_column = {}
        self.tables = []
        for li, (ins, tabs) in enumerate(circ.lookups):
            table = set()
            for row in range(self.usable):
                table.add(tuple(H.expr_eval(e, self._leaf(row), p) for e in tabs))
            self.tables.append(table)
            cells = set()
            for e in ins:
                _cells_of(e, cells)
            for t, c, r in cells:
                if t == 'advice':
                    self.lookup_by_column.setdefault(c, []).append((li, ins, r))

        class _CS:
            pass
        cs = _CS()
        cs.perm_columns, cs.n = circ.perm_columns, self.n
        self.mapping = H.build_permutation(cs, circ.copies)
        self.perm_index = {tuple(pc): i for i, pc in enumerate(circ.perm_columns)}

    def _leaf(self, row):
        return lambda t, c, r: self.cols[t][c][(row + r) % self.n]

    def failures_after(self, column: int, row: int, new_value: int):
        """names of what fails with advice[column][row] = new_value (the cell is restored afterwards):
        ('gate', gate index, constraint index, row) / ('lookup', index, row) / ('permutation', column, row)"""
        p, old = self.p, self.cols['advice'][column][row]
        self.cols['advice'][column][row] = new_value % p
        out = []
        try:
            for gi, ci, pl, lead, r in self.by_column.get(column, ()):
                at = (row - r) % self.n
                if at >= self.usable or (lead is not None and self.circ.fixed[lead][at] % p == 0):
                    continue
                if H.expr_eval(pl, self._leaf(at), p) != 0:
                    out.append(('gate', gi, ci, at))
            for li, ins, r in self.lookup_by_column.get(column, ()):
                at = (row - r) % self.n
                if at < self.usable and tuple(H.expr_eval(e, self._leaf(at), p) for e in ins) not in self.tables[li]:
                    out.append(('lookup', li, at))
            pi = self.perm_index.get(('advice', column))
            if pi is not None:
                mc, mr = self.mapping[pi][row]
                if (mc, mr) != (pi, row):
                    mk, mi = self.circ.perm_columns[mc]
                    if self.cols[mk][mi][mr] % p != new_value % p:
                        out.append(('permutation', column, row))
        finally:
            self.cols['advice'][column][row] = old
        return out
