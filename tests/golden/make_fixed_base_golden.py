#!/usr/bin/env python3
"""Extract known-answer DATA from the reference's fixed-base constant tables.

Reads (as text) the constant tables the reference's own tests check
(`generator`, `z`, `lagrange_coeffs` tests at
src/utils/constants/fixed_bases/board_commit_{v,r}.rs:2940-2960) and writes a
fixture: both GENERATOR pairs, all 85 Z values (u64 each) and ALL 680 U values per
base ("U": 85 windows x 8, the 1 360 y-coordinate known answers behind the
reference's `z` tests), plus the 48-row sample "U_rows" (always including the last
window, which uses the offset scalar) the slow big-int check walks.  Only numbers
are emitted -- no reference source text.

Run in the build container (needs /root/reference):
    python tests/golden/make_fixed_base_golden.py
"""
import json
import os
import random
import re

REF = "/root/reference/src/utils/constants/fixed_bases"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixed_bases.json")


def ints(text):
    return [int(t) for t in re.findall(r"\b\d+\b", text)]


def parse(path):
    src = open(path).read()
    g = src.index("pub const GENERATOR")
    z = src.index("pub const Z")
    u = src.index("pub const U")
    end = src.index("pub fn generator")
    gen = ints(src[src.index("=", g):z].split(");")[0])
    # strip the "32" array-length literals that appear in the type annotation
    gen = gen[-64:]
    zs = ints(src[src.index("=", z):u].split("];")[0])
    us = ints(src[src.index("=", u):end])
    assert len(gen) == 64 and len(zs) == 85 and len(us) == 85 * 8 * 32, (len(gen), len(zs), len(us))
    gx = int.from_bytes(bytes(gen[:32]), "little")
    gy = int.from_bytes(bytes(gen[32:]), "little")
    U = [[int.from_bytes(bytes(us[(w * 8 + k) * 32:(w * 8 + k + 1) * 32]), "little") for k in range(8)]
         for w in range(85)]
    return gx, gy, zs, U


def main():
    rng = random.Random(0xBA771E)
    out = {"source": "board_commit_{v,r}.rs GENERATOR/Z/U (data only)", "num_windows": 85, "H": 8, "bases": {}}
    for name, msg in (("v", "v"), ("r", "r")):
        gx, gy, zs, U = parse(os.path.join(REF, "board_commit_%s.rs" % name))
        rows = {(84, k) for k in range(8)} | {(0, k) for k in range(8)}
        while len(rows) < 48:
            rows.add((rng.randrange(85), rng.randrange(8)))
        out["bases"][name] = {
            "hash_to_curve": {"domain": "battlezips:hash2curve", "message": msg, "curve": "pallas"},
            "generator": [hex(gx), hex(gy)],
            "Z": zs,
            "U_rows": [[w, k, hex(U[w][k])] for (w, k) in sorted(rows)],
            "U": [[hex(U[w][k]) for k in range(8)] for w in range(85)],
        }
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
