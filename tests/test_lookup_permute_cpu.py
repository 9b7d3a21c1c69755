"""bzh_permute_expression_pair (host, row N4 of SURVEY section 8) against the oracle's restatement of
halo2_proofs lookup::prover::permute_expression_pair, including upstream's fill order for repeated rows."""
import random

import pytest

import coracle as C
import pasta as O


@pytest.fixture(scope="module")
def bzh2_lib():
    import os
    import __graft_entry__ as g
    import bzh2
    if not os.path.exists(bzh2.lib_path()):
        g.build()
    return bzh2


@pytest.mark.parametrize("seed", range(6))
def test_permute_matches_oracle(bzh2_lib, seed):
    F = O.FP
    rng = random.Random(seed)
    usable = rng.choice([1, 7, 64, 1000])
    table = [rng.randrange(F.p) for _ in range(usable)]
    if seed % 2:
        table = [i % 16 for i in range(usable)]            # heavy repetition, like a 10-bit range table
    inp = [rng.choice(table) for _ in range(usable)]
    # the multiset condition: every distinct input needs one table copy; repeated rows need leftovers -> always true here
    a, s = O.permute_expression_pair(inp, table, usable, F)
    ga, gs = bzh2_lib.permute_expression_pair(0, C.ints_to_array(inp + [0, 0]), C.ints_to_array(table + [0, 0]), usable)
    assert C.array_to_ints(ga) == a and C.array_to_ints(gs) == s
    assert sorted(s) == sorted(table)
    # Montgomery in / out
    R = F.R
    gam, gsm = bzh2_lib.permute_expression_pair(0, C.ints_to_array([x * R % F.p for x in inp]),
                                                C.ints_to_array([x * R % F.p for x in table]), usable, bzh2_lib.FORM_MONTGOMERY)
    assert C.array_to_ints(gam) == [x * R % F.p for x in a] and C.array_to_ints(gsm) == [x * R % F.p for x in s]


def test_input_outside_table_is_an_error(bzh2_lib):
    with pytest.raises(bzh2_lib.BzhError) as e:
        bzh2_lib.permute_expression_pair(0, C.ints_to_array([1, 2, 99]), C.ints_to_array([1, 2, 3]), 3)
    assert e.value.status == bzh2_lib.E_RANGE


@pytest.mark.parametrize("bits", [10, 20])
def test_permute_at_the_reference_lookup_shape(bzh2_lib, bits):
    """The shape the BoardCircuit's lookup has at k = 14: 16 378 usable rows, a 10-bit range table repeated down the column,
    an input column that is zero outside the 2 399 used rows (values below 2^16 take the histogram path of the host sort,
    below 2^64 the integer sort; both against the oracle's restatement)."""
    F = O.FP
    usable = 16378
    rng = random.Random(77 + bits)
    table = [i % (1 << bits) for i in range(usable)]
    inp = [0] * usable
    for i in range(2399):
        inp[i] = rng.randrange(min(1 << bits, usable))
    a, s = O.permute_expression_pair(inp, table, usable, F)
    ga, gs = bzh2_lib.permute_expression_pair(0, C.ints_to_array(inp + [0] * 6), C.ints_to_array(table + [0] * 6), usable)
    assert C.array_to_ints(ga)[:usable] == a and C.array_to_ints(gs)[:usable] == s
