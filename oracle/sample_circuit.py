"""A small satisfiable PLONKish circuit for the prover tests (test infrastructure): BattleZips-shaped
building blocks -- boolean / running-sum gates in the style of src/chips/bitify.rs:63-88, a multiplication
gate, copy constraints (incl. to an instance cell, like constrain_instance in src/chips/shot.rs:349-352)
and a range-table lookup like the 10-bit table of src/chips/pedersen.rs:56-57."""
import random

import halo2_oracle as H


def build(k=4, seed=1, with_lookup=True, degree=None, F=None):
    import pasta as O
    F = F or O.FP
    p = F.p
    rng = random.Random(seed)
    n = 1 << k
    A = lambda c, r=0: ('advice', c, r)
    Fx = lambda c, r=0: ('fixed', c, r)
    mul, add, neg, const = (lambda a, b: ('mul', a, b)), (lambda a, b: ('add', a, b)), (lambda a: ('neg', a)), (lambda v: ('const', v))
    sub = lambda a, b: add(a, neg(b))
    # fixed 0: q_mul, fixed 1: q_bits, fixed 2: lookup table (0..T-1), fixed 3: q_lookup
    gates = [
        mul(Fx(0), sub(mul(A(0), A(1)), A(2))),                                  # a0*a1 = a2
        mul(Fx(1), mul(A(0), sub(const(1), A(0)))),                               # bit is boolean
        mul(Fx(1), sub(A(2, 1), add(A(2), mul(A(0), A(1))))),                     # lc' = lc + bit*e2
        mul(Fx(1), sub(A(1, 1), ('scale', A(1), 2))),                             # e2' = 2*e2
    ]
    lookups = [([mul(Fx(3), A(0))], [Fx(2)])] if with_lookup else []
    perm_columns = [('advice', 0), ('advice', 1), ('advice', 2), ('instance', 0)]
    cs = H.ConstraintSystem(k, 3, 4, 1, gates, perm_columns, lookups, degree=degree)
    usable = cs.usable_rows
    fixed = [[0] * n for _ in range(4)]
    adv = [[0] * n for _ in range(3)]
    T = min(8, usable)
    for i in range(usable):
        fixed[2][i] = i % T
    copies = []
    nb = max(1, min(3, usable - 4))                                               # rows 0..nb-1: bit decomposition
    value = rng.randrange(1 << nb)
    lc, e2 = 0, 1
    for r in range(nb):
        bit = (value >> r) & 1
        fixed[1][r] = 1
        adv[0][r], adv[1][r], adv[2][r] = bit, e2, lc
        lc, e2 = lc + bit * e2, 2 * e2
    adv[1][nb], adv[2][nb] = e2, lc                                               # row after the last bit row
    row = nb + 1
    if row < usable:                                                              # one multiplication row
        fixed[0][row] = 1
        x, y = rng.randrange(p), rng.randrange(p)
        adv[0][row], adv[1][row], adv[2][row] = x, y, x * y % p
        copies.append(((2, nb), (0, row + 1 if row + 1 < usable else row)))        # recomposed value copied elsewhere
        if row + 1 < usable:
            adv[0][row + 1] = lc
            fixed[3][row + 1] = 1                                                 # ... and range-checked by the lookup
        row += 2
    instance = [[lc]]
    copies.append(((2, nb), (3, 0)))                                              # constrain_instance
    return cs, fixed, copies, adv, instance
