#!/usr/bin/env python3
"""Extract the reference's MockProver assertions as DATA (tests/golden/mock_fixtures.json).

Reads, as text, the #[test] functions of
  /root/reference/src/circuits/board.rs   (13 MockProver tests)
  /root/reference/src/circuits/shot.rs    (14 MockProver tests)
  /root/reference/src/chips/bitify.rs     (8 MockProver tests)
and records, per test: the inputs (deck tuples, WitnessOption list, shot coordinates, asserted hit, which public
input is off by one, bit widths / literal values of the bitify tests, k) and the asserted outcome -- Ok, or the exact
`VerifyFailure` vector: gate index + name, constraint index + name, region index + name, offset, cell values;
permutation failures with their column and location.  Numbers and strings only; no source text is kept.

Run in the build container:  python tests/golden/make_mock_golden.py
"""
import json
import os
import re

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mock_fixtures.json")


def strip_comments(src):
    return "\n".join(line.split("//")[0] for line in src.splitlines())


def tests_of(path):
    src = strip_comments(open(path).read())
    parts = re.split(r"#\[test\]\s*fn\s+(\w+)\s*\(\)", src)
    return {parts[i]: parts[i + 1] for i in range(1, len(parts) - 1, 2)}


def deck_of(body):
    m = re.search(r"Deck::from\(\[(.*?)\]\)", body, re.S)
    ships = []
    for t in re.finditer(r"Some\(\((\d+),\s*(\d+),\s*(true|false)\)\)|None", m.group(1)):
        ships.append(None if t.group(1) is None else [int(t.group(1)), int(t.group(2)), t.group(3) == "true"])
    assert len(ships) == 5
    return ships


def options_of(body):
    m = re.search(r"let witness_options = \[(.*?)\];", body, re.S)
    if not m:
        return ["Default"] * 5
    opts = re.findall(r"WitnessOption::(\w+)", m.group(1))
    assert len(opts) == 5
    return opts


def failures_of(body):
    m = re.search(r"Err\(vec!\[(.*)\]\)", body, re.S)
    if not m:
        assert "Ok(())" in body or "assert_satisfied" in body
        return None
    out = []
    for chunk in re.split(r"VerifyFailure::", m.group(1))[1:]:
        if chunk.startswith("ConstraintNotSatisfied"):
            g = re.search(r'\((\d+),\s*"([^"]+)"\)\s*\.into\(\),\s*(\d+),\s*"([^"]+)"', chunk)
            r = re.search(r'region:\s*\((\d+),\s*"([^"]+)"\)\s*\.into\(\),\s*offset:\s*(\d+)', chunk)
            cells = [[kind.lower(), int(col), int(rot), val] for kind, col, rot, val in
                     re.findall(r'\(\s*\(\s*\(Any::(\w+),\s*(\d+)\)\s*\.into\(\),\s*(-?\d+)\)\s*\.into\(\),\s*String::from\("([^"]+)"\)', chunk)]
            out.append({"type": "ConstraintNotSatisfied", "gate": [int(g.group(1)), g.group(2)],
                        "constraint": [int(g.group(3)), g.group(4)], "region": [int(r.group(1)), r.group(2)],
                        "offset": int(r.group(3)), "cell_values": cells})
        elif chunk.startswith("Permutation"):
            c = re.search(r"\(Any::(\w+),\s*(\d+)\)", chunk)
            f = {"type": "Permutation", "column": [c.group(1).lower(), int(c.group(2))]}
            r = re.search(r'region:\s*\((\d+),\s*"([^"]+)"\)\s*\.into\(\),\s*offset:\s*(\d+)', chunk)
            if r:
                f["region"] = [int(r.group(1)), r.group(2)]
                f["offset"] = int(r.group(3))
            else:
                f["outside_row"] = int(re.search(r"OutsideRegion\s*\{\s*row:\s*(\d+)", chunk).group(1))
            out.append(f)
        else:
            raise ValueError("unknown failure kind: " + chunk[:40])
    return out


def public_tweak(body, names):
    """index of the public input that the test bumps by one (None: all honest)"""
    if re.search(r"\.x\(\)\.to_owned\(\)\s*\+\s*pallas::Base::one\(\)", body):
        return 0
    m = re.search(r"let public_outputs = vec!\[(.*?)\];", body, re.S)
    if m:
        entries = [e.strip() for e in m.group(1).split(",\n") if e.strip()]
        for i, e in enumerate(entries):
            if "+ pallas::Base::one()" in e:
                return i
    return None


def board_cases():
    out = []
    for name, body in tests_of(os.path.join(REF, "circuits/board.rs")).items():
        if name == "production":
            continue
        out.append({"test": name, "k": int(re.search(r"MockProver::run\((\d+)", body).group(1)), "deck": deck_of(body),
                    "options": options_of(body), "public_plus_one": public_tweak(body, 2), "expect": failures_of(body)})
    return out


def shot_cases():
    out = []
    for name, body in tests_of(os.path.join(REF, "circuits/shot.rs")).items():
        if name == "production":
            continue
        m = re.search(r"serialize::<(\d+)>\(\[([\d,\s]*)\],\s*\[([\d,\s]*)\]\)", body)
        if m:
            xs = [int(v) for v in m.group(2).split(",") if v.strip()]
            ys = [int(v) for v in m.group(3).split(",") if v.strip()]
            shots = [[x, y] for x, y in zip(xs, ys)]
        else:
            assert "let shot = BinaryValue::empty()" in body
            shots = []
        out.append({"test": name, "k": int(re.search(r"MockProver::run\((\d+)", body).group(1)), "deck": deck_of(body),
                    "shots": shots, "hit": int(re.search(r"let hit = BinaryValue::from_u8\((\d+)\)", body).group(1)),
                    "public_plus_one": public_tweak(body, 4), "expect": failures_of(body)})
    return out


def bitify_cases():
    out = []
    tests = tests_of(os.path.join(REF, "chips/bitify.rs"))
    default_bits = int(re.search(r"const DEFAULT_BITS: usize = (\d+);", open(os.path.join(REF, "chips/bitify.rs")).read()).group(1))
    for name, body in tests.items():
        if name == "test_battlezips":
            ship = re.search(r"Ship::new\(ShipType::(\w+),\s*(\d+),\s*(\d+),\s*(true|false)\)", body)
            runs = re.findall(r"MockProver::run\((\w+),", body)
            ksize = int(re.search(r"const CIRCUIT_SIZE: u32 = (\d+);", open(os.path.join(REF, "chips/bitify.rs")).read()).group(1))
            ks = [ksize if r == "CIRCUIT_SIZE" else int(r) for r in runs]
            base = {"test": name, "circuit": "num2bits", "bits": 100,
                    "ship": [ship.group(1), int(ship.group(2)), int(ship.group(3)), ship.group(4) == "true"]}
            out.append(dict(base, k=ks[0], value_plus=0, expect=None))
            out.append(dict(base, k=ks[1], value_plus=1, expect=failures_of(body)))
            continue
        circuit = "num2bits" if "Num2BitsCircuit" in body else "bits2num"
        b = re.search(r"(Num2Bits|Bits2Num)Circuit::<(\w+)>", body).group(2)
        bits = default_bits if b == "DEFAULT_BITS" else int(b)
        k = int(re.search(r"MockProver::run\((\d+)", body).group(1))
        case = {"test": name, "circuit": circuit, "bits": bits, "k": k, "expect": failures_of(body)}
        m = re.search(r"Fp::from\((\d+)u64\)", body)
        h = re.search(r'hex::decode\("([0-9a-fA-F]+)"\)', body)
        if m:
            case["value"] = m.group(1)
            case["binary"] = "value"                    # BinaryValue::from_fp(value)
        elif h:
            case["value"] = "0"                          # Fp::zero()
            case["binary"] = "0x" + h.group(1)           # big-endian hex of the (reversed) repr bytes
        elif "Fp::zero().sub(&Fp::one())" in body:
            case["value"] = "-1"
            case["binary"] = "value"
        elif "Fp::zero()" in body:
            case["value"] = "0"
            case["binary"] = "value"
        else:
            raise ValueError(name)
        out.append(case)
    return out


def main():
    out = {"source": "MockProver assertions of src/circuits/board.rs, src/circuits/shot.rs, src/chips/bitify.rs (data only)",
           "board": board_cases(), "shot": shot_cases(), "bitify": bitify_cases()}
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, {k: len(v) for k, v in out.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
