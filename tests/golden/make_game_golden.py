#!/usr/bin/env python3
"""Extract game-level known answers (DATA only) from the reference's own tests:
  * src/circuits/shot.rs: for each valid_* / invalid_assert_* test the deck tuples, the shot (x, y)
    and whether the shot truly hits (the asserted `hit` for valid tests, its negation for the two
    invalid_assert_* tests);
  * src/circuits/board.rs:283,287: the H5 / V5 cell values of the DualPlacement carrier (3,3,vertical).
Run in the build container:  python tests/golden/make_game_golden.py
"""
import json
import os
import re

REF = "/root/reference/src/circuits"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "game_fixtures.json")


def tests_of(path):
    src = open(path).read()
    parts = re.split(r"#\[test\]\s*fn\s+(\w+)\s*\(\)", src)
    return {parts[i]: parts[i + 1] for i in range(1, len(parts) - 1, 2)}


def deck_of(body):
    m = re.search(r"Deck::from\(\[(.*?)\]\)", body, re.S)
    ships = []
    for t in re.findall(r"Some\(\((\d+),\s*(\d+),\s*(true|false)\)\)|None", m.group(1)):
        ships.append(None if t[0] == "" else [int(t[0]), int(t[1]), t[2] == "true"])
    return ships


def main():
    shot_tests = tests_of(os.path.join(REF, "shot.rs"))
    cases = []
    for name in ("valid_hit_0", "valid_hit_1", "valid_miss_0", "valid_miss_1", "invalid_assert_hit_when_miss",
                 "invalid_assert_miss_when_hit"):
        body = shot_tests[name]
        sx, sy = re.search(r"serialize::<1>\(\[(\d+)\],\s*\[(\d+)\]\)", body).groups()
        asserted = int(re.search(r"BinaryValue::from_u8\((\d+)\)", body).group(1))
        truth = asserted if name.startswith("valid") else 1 - asserted
        cases.append({"test": name, "deck": deck_of(body), "shot": [int(sx), int(sy)], "hit": truth})
    board_src = open(os.path.join(REF, "board.rs")).read()
    dual = re.search(r'"0x(200000000)".*?"0x(3c00000000)"', board_src, re.S)
    out = {"source": "src/circuits/shot.rs tests; src/circuits/board.rs:283,287 (data only)", "shots": cases,
           "dual_placement_carrier_3_3_vertical": {"H5": "0x" + dual.group(1), "V5": "0x" + dual.group(2)}}
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, len(cases), "shot cases")


if __name__ == "__main__":
    main()
