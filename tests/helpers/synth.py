"""TEST HELPER (not product code; the product path is libbzh2.so behind include/bzh2.h).
Synthetic circuits with the SHAPE of the reference's Shot / Board circuits, for benchmarking the prover.

The real gate polynomials cannot be restated: 19 of Shot's 24 / Board's 57 gates come from the `halo2_gadgets`
crate, which is not on disk.  What fixes the prover's cost is the shape:
11 advice + 8 fixed + 1 instance columns, 13 permutation columns, constraint degree 9 (extended domain 2^(k+3)),
one 10-bit range lookup, a few dozen gates mixing boolean / running-sum rows (src/chips/bitify.rs:63-88),
multiplications, and a degree-9 membership product like the placement chip's interpolated window indicator
(src/chips/placement.rs:187-204).  The witness below satisfies every gate, copy and lookup.
"""
from __future__ import annotations

import random

from bzh2.circuit_data import Circuit, MODULI


def battlezips_shaped(k: int, seed: int = 1, field: int = 0):
    p = MODULI[field]
    rng = random.Random(seed)
    n = 1 << k
    A = lambda c, r=0: ('advice', c, r)
    F = lambda c, r=0: ('fixed', c, r)
    mul = lambda a, b: ('mul', a, b)
    add = lambda a, b: ('add', a, b)
    sub = lambda a, b: ('add', a, ('neg', b))
    const = lambda v: ('const', v)
    one = const(1)
    # fixed: 0 constants column (in the permutation, like fixed[0] of the chips), 1 q_mul, 2 q_bits, 3 q_add, 4 q_set,
    #        5 q_lookup, 6 q_bool, 7 range table
    Q_MUL, Q_BITS, Q_ADD, Q_SET, Q_LK, Q_BOOL, TABLE = 1, 2, 3, 4, 5, 6, 7
    member = A(10)
    for v in range(1, 8):
        member = mul(member, sub(A(10), const(v)))                    # a10 (a10-1) ... (a10-7): degree 8
    gates = [
        mul(F(Q_MUL), sub(mul(A(0), A(1)), A(2))),
        mul(F(Q_MUL), sub(mul(A(3), A(4)), A(5))),
        mul(F(Q_BITS), mul(A(6), sub(one, A(6)))),
        mul(F(Q_BITS), sub(A(8, 1), add(A(8), mul(A(6), A(7))))),
        mul(F(Q_BITS), sub(A(7, 1), ('scale', A(7), 2))),
        mul(F(Q_ADD), sub(A(9), add(A(0), A(1)))),
        mul(F(Q_SET), member),                                          # degree 9
    ]
    for col in (0, 1, 2, 3, 4, 5, 9, 10):                               # boolean rows on several columns, like the chips' bit cells
        gates.append(mul(F(Q_BOOL), mul(A(col), sub(one, A(col)))))
    for col in (0, 3):                                                  # products of neighbouring rows
        gates.append(mul(F(Q_BOOL), sub(A(col + 2, 1), mul(A(col), A(col + 1, 1)))))
    while len(gates) < 24:                                              # Shot has 24 gates
        j = len(gates) % 5
        gates.append(mul(F(Q_ADD), sub(add(A(j), A(j + 1)), add(A(j + 1), A(j)))))   # satisfied identically
    lookups = [([mul(F(Q_LK), A(9))], [F(TABLE)])]
    perm_columns = [('advice', i) for i in range(11)] + [('instance', 0), ('fixed', 0)]
    fixed = [[0] * n for _ in range(8)]
    adv = [[0] * n for _ in range(11)]
    circ = Circuit(k, 11, 8, 1, gates, perm_columns, lookups, fixed, [], degree=9)
    usable = circ.usable_rows
    tsize = min(1024, usable)
    for r in range(usable):
        fixed[TABLE][r] = r % tsize
    copies = []
    row = 0
    # bit-decomposition blocks: 100 bit rows + the closing row, until half of the table is used
    recomposed = []
    while row + 101 <= usable // 2:
        value = rng.getrandbits(100)
        lc, e2 = 0, 1
        for i in range(100):
            bit = (value >> i) & 1
            fixed[Q_BITS][row + i] = 1
            adv[6][row + i], adv[7][row + i], adv[8][row + i] = bit, e2 % p, lc % p
            lc, e2 = lc + bit * e2, 2 * e2
        adv[7][row + 100], adv[8][row + 100] = e2 % p, lc % p
        recomposed.append((row + 100, lc % p))
        row += 101
    # the rest: mul/add rows, membership rows, lookup rows, boolean-pair rows
    while row < usable:
        kind = row % 4
        if kind == 0:
            fixed[Q_MUL][row] = fixed[Q_ADD][row] = 1
            x, y, z, t = (rng.randrange(p) for _ in range(4))
            adv[0][row], adv[1][row], adv[2][row] = x, y, x * y % p
            adv[3][row], adv[4][row], adv[5][row] = z, t, z * t % p
            adv[9][row] = (x + y) % p
        elif kind == 1:
            fixed[Q_SET][row] = 1
            adv[10][row] = rng.randrange(8)
        elif kind == 2:
            fixed[Q_LK][row] = 1
            adv[9][row] = rng.randrange(tsize)
        row += 1
    # copy constraints: every recomposed value is copied into a multiplication row's a0 (if one exists) and the
    # first one is exposed through the instance column (constrain_instance, src/chips/shot.rs:349-352)
    mul_rows = [r for r in range(usable) if fixed[Q_MUL][r]]
    for i, (rr, val) in enumerate(recomposed):
        if i < len(mul_rows):
            mr = mul_rows[i]
            adv[0][mr] = val
            adv[2][mr] = val * adv[1][mr] % p
            adv[9][mr] = (val + adv[1][mr]) % p
            copies.append(((8, rr), (0, mr)))
    instance = [[recomposed[0][1] if recomposed else 0]]
    if recomposed:
        copies.append(((8, recomposed[0][0]), (11, 0)))
    fixed[0][0] = 1                                                     # a constant cell of the permutation's fixed column
    circ.fixed, circ.copies = fixed, copies
    return circ, adv, instance
