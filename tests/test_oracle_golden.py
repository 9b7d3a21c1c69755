"""The oracle against the reference's own known-answer data.

Golden data: tests/golden/fixed_bases.json, extracted (numbers only) from
src/utils/constants/fixed_bases/board_commit_{v,r}.rs by
tests/golden/make_fixed_base_golden.py.  The reference checks the same data in
its `generator`, `z` and `lagrange_coeffs` tests (board_commit_*.rs:2940-2960).
"""
import json
import os
import random

import numpy as np
import pytest

import coracle as C
import pasta as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fixed_bases.json")))


def test_modulus_literal():
    # src/chips/bitify.rs:461 spells the Fp modulus in hex
    assert O.P == int("40000000000000000000000000000000224698fc094cf91b992d30ed00000001", 16)
    assert O.P.bit_length() == 255 and O.Q.bit_length() == 255


@pytest.mark.parametrize("name", ["v", "r"])
def test_generators_on_pallas(name):
    g = tuple(int(x, 16) for x in GOLD["bases"][name]["generator"])
    assert O.PALLAS.is_on_curve(g)


@pytest.mark.parametrize("name", ["v", "r"])
def test_hash_to_curve_generators(name):
    """board_commit_{v,r}.rs:2941-2948: hash_to_curve("battlezips:hash2curve")(b"v"|b"r") == GENERATOR."""
    b = GOLD["bases"][name]
    h = b["hash_to_curve"]
    got = O.hash_to_curve(h["curve"], h["domain"], h["message"].encode())
    assert got == tuple(int(x, 16) for x in b["generator"])


@pytest.mark.parametrize("name", ["v", "r"])
def test_window_table_relation_python(name):
    """U[w][k]^2 - Z[w] == y([(k+2) 8^w] B)  (last window: offset scalar), SURVEY App. A.3."""
    b = GOLD["bases"][name]
    G = tuple(int(x, 16) for x in b["generator"])
    for w, k, u in b["U_rows"]:
        u, z = int(u, 16), b["Z"][w]
        s = (k + 2) * 8 ** w if w < 84 else k * 8 ** 84 - sum(2 * 8 ** j for j in range(84))
        pt = O.PALLAS.mul(s % O.Q, G)
        assert (u * u - z - pt[1]) % O.P == 0, (name, w, k)


@pytest.mark.parametrize("name", ["v", "r"])
def test_window_table_relation_c_oracle(oracle_c, name):
    """ALL 680 U values per base (the 1 360 known answers of the reference's `z` tests, board_commit_{v,r}.rs:28-2927,
    :2956-2960) through the C restatement (Montgomery 4x64, Jacobian)."""
    b = GOLD["bases"][name]
    G = C.points_to_array([tuple(int(x, 16) for x in b["generator"])])[0]
    assert len(b["U"]) == 85 and all(len(row) == 8 for row in b["U"])
    for w, k, u in [(w, k, b["U"][w][k]) for w in range(85) for k in range(8)]:
        u, z = int(u, 16), b["Z"][w]
        s = ((k + 2) * 8 ** w if w < 84 else k * 8 ** 84 - sum(2 * 8 ** j for j in range(84))) % O.Q
        out = np.zeros(8, dtype=np.uint64)
        C.lib().orc_point_mul(1, C._p(C.int_to_limbs(s)), C._p(G), C._p(out))
        y = C.limbs_to_int(out[4:])
        assert (u * u - z - y) % O.P == 0, (name, w, k)


def test_pedersen_commit_matches_msm():
    """src/utils/pedersen.rs:17-28 is a 2-term MSM over the hashed generators."""
    rng = random.Random(11)
    m, t = rng.randrange(1 << 100), rng.randrange(O.Q)
    V = tuple(int(x, 16) for x in GOLD["bases"]["v"]["generator"])
    R = tuple(int(x, 16) for x in GOLD["bases"]["r"]["generator"])
    assert O.pedersen_commit(m, t) == O.PALLAS.msm_naive([m, t], [V, R])


def test_c_oracle_fields_match_python(oracle_c):
    rng = random.Random(5)
    for fid, F in O.FIELD_BY_ID.items():
        for _ in range(20):
            a, b = rng.randrange(F.p), rng.randrange(F.p)
            assert C.field_mul(fid, a, b) == a * b % F.p
        for a in (1, F.p - 1, rng.randrange(1, F.p)):
            assert C.field_inv(fid, a) == F.inv(a)


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_c_oracle_msm_matches_definition(oracle_c, cid):
    rng = random.Random(100 + cid)
    cv = O.CURVE_BY_ID[cid]
    pts = [cv.random_point(rng) for _ in range(40)] + [None]
    pts += [pts[0], cv.neg(pts[1])]
    sc = [rng.randrange(cv.scalar.p) for _ in pts]
    sc[3], sc[4], sc[5] = 0, cv.scalar.p - 1, 1
    want = cv.msm_naive(sc, pts)
    S, Pn = C.ints_to_array(sc), C.points_to_array(pts)
    assert C.array_to_point(C.msm(cid, S, Pn, 1)) == want
    assert C.array_to_point(C.msm(cid, S, Pn, 4)) == want
    assert C.array_to_point(C.msm_naive(cid, S, Pn)) == want
    assert cv.msm_pippenger(sc, pts) == want


@pytest.mark.parametrize("fid", [0, 1, 2])
def test_c_oracle_ntt_matches_definition(oracle_c, fid):
    rng = random.Random(200 + fid)
    F = O.FIELD_BY_ID[fid]
    for k in (0, 1, 3, 6):
        v = [rng.randrange(F.p) for _ in range(1 << k)]
        w = F.omega(k)
        A = C.ints_to_array(v)
        got = C.array_to_ints(C.ntt(fid, A, w))
        assert got == O.dft_naive(v, w, F) == O.ntt(v, w, F)
        assert C.array_to_ints(C.ntt(fid, C.ntt(fid, A, w), w, inverse=True)) == v
        cs = C.array_to_ints(C.ntt(fid, A, w, coset_shift=F.g))
        assert cs == O.coset_ntt(v, w, F.g, F)
        assert C.array_to_ints(C.ntt(fid, C.ints_to_array(cs), w, inverse=True, coset_shift=F.g)) == v


def test_c_oracle_ntt_large_threads(oracle_c):
    rng = np.random.default_rng(3)
    F = O.FP
    n = 1 << 13
    A = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    A[:, 3] &= (1 << 61) - 1
    w = F.omega(13)
    one = C.ntt(0, A, w, threads=1)
    many = C.ntt(0, A, w, threads=4)
    assert (one == many).all()
    # spot-check three outputs against the definition
    v = C.array_to_ints(A)
    for i in (0, 1, 4097):
        assert C.limbs_to_int(one[i]) == sum(x * pow(w, i * j, F.p) for j, x in enumerate(v)) % F.p


def test_eval_polynomial(oracle_c):
    rng = random.Random(9)
    F = O.FP
    cs = [rng.randrange(F.p) for _ in range(33)]
    x = rng.randrange(F.p)
    assert C.eval_poly(0, C.ints_to_array(cs), x) == O.eval_polynomial(cs, x, F)


def test_c_oracle_gate_eval_and_generator_collapse_match_the_bigint_oracle():
    """The two C restatements the CPU baseline times (oracle/oracle.c: orc_gate_eval, orc_generator_collapse) against the
    big-int oracle: y-folded constraint polynomials of the real ShotCircuit on random columns, and g_lo + [u] g_hi."""
    import random
    import numpy as np
    import blob as Bm
    import coracle as Cc
    import halo2_oracle as Hh
    import pasta as Oo
    from bzh2 import circuits as Cm
    lay = Cm.CircuitLayout(Cm.SHOT, 11)
    circ = Bm.decode(lay.blob())
    lay.close()
    F = Oo.FP
    rng = random.Random(3)
    gates = circ.gates[:40]
    prog, consts, colmap = Cc.compile_gates(gates)
    size = 64
    cols_int = {key: [rng.randrange(F.p) for _ in range(size)] for key in colmap}
    cols = [None] * len(colmap)
    for key, idx in colmap.items():
        cols[idx] = Cc.ints_to_array(cols_int[key])
    y = rng.randrange(F.p)
    got = Cc.array_to_ints(Cc.gate_eval(0, prog, consts, cols, y, 5, 9, threads=2, rot_scale=8))
    for r, g in zip(range(5, 9), got):
        acc = 0
        for e in gates:
            v = Hh.expr_eval(e, lambda t, c, rot, r=r: cols_int[(t, c)][(r + 8 * rot) % size], F.p)
            acc = (acc * y + v) % F.p
        assert g == acc
    cv = Oo.VESTA
    pts = [cv.random_point(rng) for _ in range(8)]
    u = rng.randrange(F.p)
    want = [cv.add(pts[i], cv.mul(u, pts[4 + i])) for i in range(4)]
    got = Cc.generator_collapse(0, Cc.points_to_array(pts), u, threads=2)
    assert [Cc.array_to_point(got[i]) for i in range(4)] == want
