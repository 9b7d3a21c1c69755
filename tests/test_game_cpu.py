"""Host mirror of the reference's game/witness marshalling (rows a9, a2/a5 traces) against data the
reference's own tests hold (tests/golden/game_fixtures.json, extracted by make_game_golden.py) and the
hand-derived values of SURVEY App. B."""
import json
import os

import pytest

from bzh2 import game as G

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "game_fixtures.json")))
PATTERN1 = [(3, 3, True), (5, 4, False), (0, 1, False), (0, 5, True), (6, 1, False)]       # src/circuits/board.rs:101-107
PATTERN2 = [(3, 4, False), (9, 6, True), (0, 0, False), (0, 6, False), (6, 1, True)]       # src/circuits/board.rs:134-140


@pytest.mark.parametrize("case", FIX["shots"], ids=lambda c: c["test"])
def test_shot_hits_match_reference_tests(case):
    """src/circuits/shot.rs valid_* / invalid_assert_*: the shot trace's final hit count is the truth."""
    board = G.Board.from_(G.Deck.from_([None if s is None else tuple(s) for s in case["deck"]]))
    state = board.state()
    shot = G.serialize([case["shot"][0]], [case["shot"][1]])
    shot_trace, hit_trace = G.compute_shot_trace(state, shot)
    assert shot_trace[-1] == 1
    assert hit_trace[-1] == case["hit"]
    assert state.to_fp() == state.lower_u128()          # message = Fp::from_u128(lower_u128) (src/utils/binary.rs:62-72)


def test_dual_placement_cells_match_board_test():
    """src/circuits/board.rs:283,287: carrier (3,3,vertical) with DualPlacement -> H5 / V5 cell values."""
    h, v = G.Ship(0, 3, 3, True).witness(G.DUAL_PLACEMENT)
    assert hex(h.value) == FIX["dual_placement_carrier_3_3_vertical"]["H5"]
    assert hex(v.value) == FIX["dual_placement_carrier_3_3_vertical"]["V5"]
    with pytest.raises(ValueError):                      # H and V overlap nowhere here, zip must still work
        G.BinaryValue(1).zip(G.BinaryValue(1))
    assert h.zip(v).value == (h.value | v.value)


def test_states_and_witnesses_survey_appendix_b():
    b1 = G.Board.from_(G.Deck.from_(PATTERN1))
    assert b1.state().value == 0x2409025e80200031c00
    assert [w.value for w in b1.witness()] == [0x0, 0x3e00000000, 0x1e00000000000, 0x0, 0x1c00, 0x0, 0x0, 0xe0, 0x30000, 0x0]
    b2 = G.Board.from_(G.Deck.from_(PATTERN2))
    assert b2.state().value == 0x8020080207000f80004010007
    assert [w.value for w in b2.witness()] == [0xf80000000000, 0x0, 0x0, 0xf000000000000000000000000, 0x7, 0x0,
                                               0x7000000000000000, 0x0, 0x0, 0x6000000000000000]
    assert bin(b1.state().value).count("1") == 17 == bin(b2.state().value).count("1")


def test_placement_trace_rules():
    """A valid placement ends with bit_sum == S and exactly one full window (src/chips/placement.rs:380-419);
    a ship wrapped across a row edge or split in two never counts a full window."""
    for i, (x, y, z) in enumerate(PATTERN1):
        ship = G.Ship(i, x, y, z)
        bit_sum, win = G.compute_placement_trace(ship.bits(True), ship.length())
        assert bit_sum[-1] == ship.length() and win[-1] == 1
    wrapped = G.BinaryValue(0b11111 << 8)                # cells 8..12 wrap from row 0 into row 1
    assert G.compute_placement_trace(wrapped, 5)[1][-1] == 0
    h, _ = G.Ship(1, 5, 4, False).witness(G.NONCONSECUTIVE)
    assert G.compute_placement_trace(h, 4) [0][-1] == 4 and G.compute_placement_trace(h, 4)[1][-1] == 0
    h, _ = G.Ship(1, 5, 4, False).witness(G.OVERSIZED)
    assert G.compute_placement_trace(h, 4)[0][-1] == 5


def test_binary_value_basics():
    v = G.BinaryValue.from_u8(5)
    assert v.bitfield(4) == [1, 0, 1, 0] and v.to_repr()[0] == 5 and G.BinaryValue.from_repr(v.to_repr()) == v
    with pytest.raises(ValueError):
        G.BinaryValue(G.FP_MODULUS).to_fp()


def test_wasm_record_round_trip_and_canonical_check():
    """src/wasm/circuit_wasm.rs:27-31,75-83: {commitment: Vec<[u8;32]>, proof: Vec<u8>} as serde writes it, through the
    C ABI (bzh_record_*); reading back refuses non-canonical public inputs as BinaryValue::from_repr(..).to_fp() does (:86-116)."""
    import json
    from bzh2.wire import BattleZipsRecord, KIND_SHOT, record_stride
    from bzh2.game import FP_MODULUS
    rec = BattleZipsRecord([5, FP_MODULUS - 1, 1 << 53, 1], bytes(range(200)), KIND_SHOT, 77)
    text = rec.to_json()
    d = json.loads(text)
    assert set(d) == {"commitment", "proof"}
    assert [len(c) for c in d["commitment"]] == [32] * 4 and d["commitment"][0][:2] == [5, 0] and d["proof"] == list(range(200))
    assert d["commitment"][1] == list((FP_MODULUS - 1).to_bytes(32, "little"))
    back = BattleZipsRecord.from_json(text, KIND_SHOT, 77)
    assert back == rec
    # serde's own spelling with whitespace parses too; fields in either order
    assert BattleZipsRecord.from_json(json.dumps({"proof": d["proof"], "commitment": d["commitment"]}, indent=1), KIND_SHOT, 77) == rec
    d2 = json.loads(text)
    d2["commitment"][1] = list(FP_MODULUS.to_bytes(32, "little"))            # p itself is not a canonical Fp
    with pytest.raises(ValueError):
        BattleZipsRecord.from_json(json.dumps(d2))
    for bad in ('{"commitment":[[1,2,3]],"proof":[]}', '{"commitment":[],"proof":[256]}', '{"commitment":[],"proof":[1.5]}',
                '{"commitment":[]}', '{"commitment":[],"proof":[],"x":1}', '{"commitment":[],"proof":[1,]}', text + "x", ""):
        with pytest.raises(ValueError):
            BattleZipsRecord.from_json(bad)
    with pytest.raises(ValueError):                                         # five public inputs: no circuit of the reference has them
        BattleZipsRecord.from_json(json.dumps({"commitment": [[0] * 32] * 5, "proof": []}))
    assert BattleZipsRecord.from_json('{"commitment":[],"proof":[]}') == BattleZipsRecord([], b"")
    # fixed-stride form: what the multi-GPU gather carries
    fixed = rec.to_fixed(256)
    assert len(fixed) == record_stride(256) == 144 + 256
    assert BattleZipsRecord.from_fixed(fixed) == rec and BattleZipsRecord.proof_from_fixed(fixed) == rec.proof
    with pytest.raises(ValueError):
        rec.to_fixed(100)                                                   # proof longer than the stride
    with pytest.raises(ValueError):
        BattleZipsRecord([FP_MODULUS], b"").to_fixed(8)                     # non-canonical input refused on encode
    corrupt = bytearray(fixed)
    corrupt[0:4] = (257).to_bytes(4, "little")                              # length beyond the stride
    with pytest.raises(ValueError):
        BattleZipsRecord.from_fixed(bytes(corrupt))
    corrupt = bytearray(fixed)
    corrupt[16 + 32:16 + 64] = FP_MODULUS.to_bytes(32, "little")            # a gathered record with a non-canonical input
    with pytest.raises(ValueError):
        BattleZipsRecord.from_fixed(bytes(corrupt))
