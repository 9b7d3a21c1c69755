"""Mutation test of the 19 halo2_gadgets gates restated in csrc/circuit/ecc.hpp.

The reference exercises those gates only through whole circuits (src/chips/pedersen.rs:49-134 wires EccChip +
LookupRangeCheckConfig; the MockProver tests of src/circuits/{board,shot}.rs show that VALID witnesses satisfy them).
Upstream's source is not in this tree, so the strongest pin left is the converse: every advice cell the Pedersen regions
assign (BoardCircuit regions 28-35, the last one "complete point addition" = 35: src/circuits/board.rs:867; ShotCircuit
regions 5-12, :684 of src/circuits/shot.rs) must be NOTICED by some gate, lookup or copy constraint -- a cell that can be
changed freely is a dropped or mis-stated polynomial.  For a valid witness every such cell is perturbed by +1 (and a
sample by a random value); oracle/mock_prover.LocalChecker evaluates exactly the constraints that can see the cell."""
import random

import pytest

import blob as B
import coracle as Cc
import mock_prover as M
import pasta as O
from helpers import real_parity as R

FP = O.FP.p


def _case(kind):
    from bzh2 import circuits as C
    if kind == "board":
        lay = C.CircuitLayout(C.BOARD, 12)
        circuits = R.board_circuits(C, 5150, 1)
        first, last_name = 28, "complete point addition"
    else:
        lay = C.CircuitLayout(C.SHOT, 11)
        circuits = R.shot_circuits(C, 5151, 1)
        first, last_name = 5, "complete point addition"
    desc, circ = lay.describe(), B.decode(lay.blob())
    adv, inst = lay.synthesize(circuits)
    lay.close()
    cols = [Cc.array_to_ints(adv[0, c]) for c in range(adv.shape[1])]
    return desc, circ, cols, [list(inst[0][0])], first, last_name


@pytest.mark.parametrize("kind", ["board", "shot"])
def test_every_cell_of_the_pedersen_regions_is_constrained(oracle_c, kind):
    desc, circ, cols, inst, first, last_name = _case(kind)
    regions = desc["regions"]
    assert len(regions) == first + 8 and regions[-1]["name"] == last_name      # the reference's asserted region map
    assert M.verify(circ, desc, cols, inst) == [], "the unperturbed witness must satisfy the circuit"
    chk = M.LocalChecker(circ, desc, cols, inst)
    rng = random.Random(kind)
    cells, uncaught, by_kind = 0, [], {"gate": 0, "lookup": 0, "permutation": 0}
    gates_hit = set()
    for ri in range(first, first + 8):
        seen = set()
        for c, row in regions[ri]["advice_cells"]:
            if (c, row) in seen:
                continue
            seen.add((c, row))
            cells += 1
            for delta in (1,) if cells % 7 else (1, rng.randrange(2, FP)):
                f = chk.failures_after(c, row, cols[c][row] + delta)
                if not f:
                    uncaught.append((ri, regions[ri]["name"], c, row - regions[ri]["row_lo"], delta == 1))
                for x in f:
                    by_kind[x[0]] += 1
                    if x[0] == "gate":
                        gates_hit.add(x[1])
    # the fixed-base multiplications and the addition assign > 1 000 cells; all of them must be noticed
    assert cells > 1000, cells
    assert not uncaught, "cells no constraint notices: %r" % uncaught[:20]
    assert by_kind["gate"] and by_kind["lookup"] and by_kind["permutation"], by_kind
    # every halo2_gadgets gate the circuits ENABLE takes part (variable-base and short fixed-base multiplication gates are
    # configured but never enabled by the reference: src/chips/pedersen.rs:104-134 uses mul_fixed full-width + base-field only)
    ecc = [gi for gi, g in enumerate(desc["gates"]) if (37 <= gi <= 55 if kind == "board" else 2 <= gi <= 20)]
    assert len(ecc) == 19
    enabled = {gi for gi in ecc if _enabled_somewhere(circ, desc, gi, rng)}
    assert enabled, "no halo2_gadgets gate is enabled?"
    assert enabled <= gates_hit, "enabled gates no mutation ever tripped: %r" % sorted(enabled - gates_hit)
    names = sorted(desc["gates"][gi]["name"] for gi in enabled)
    print("%s: %d cells of regions %d-%d perturbed, all noticed (%r); %d of the 19 gates enabled and tripped: %s"
          % (kind, cells, first, first + 7, by_kind, len(enabled), names))


def _enabled_somewhere(circ, desc, gi, rng):
    """a gate's compressed selector S multiplies all its constraints (S * C_j): it is enabled on a row iff some S * C_j is
    non-zero there for RANDOM advice values (selectors share fixed columns, so a non-zero column alone does not say which
    gate is on)"""
    g = desc["gates"][gi]
    polys = circ.gates[g["first_poly"]:g["first_poly"] + len(g["constraints"])]
    lead = M._leading_fixed_factor(polys[0])
    rows = [r for r in range(desc["usable_rows"]) if lead is None or circ.fixed[lead][r] % FP]
    for row in rows:
        noise = {}

        def leaf(t, c, r, row=row):
            if t == 'fixed':
                return circ.fixed[c][(row + r) % circ.n]
            return noise.setdefault((t, c, r), rng.randrange(FP))
        if any(M.H.expr_eval(pl, leaf, FP) for pl in polys):
            return True
    return False
