"""GPU parity of the gate-expression evaluator: compiled straight-line
programs on the GPU against direct recursive evaluation of the expression tree with Python integers.
Reference seam: poly::Evaluator under vanishing::Argument::construct (create_proof step 6,
benches/shot.rs:68); gate shapes as built by the reference's create_gate calls
(src/chips/bitify.rs:63-88, src/chips/transpose.rs:60-88)."""
import random

import numpy as np
import pytest

import coracle as C
import pasta as O

pytestmark = pytest.mark.gpu


def rand_tree(rng, ncols, depth, p):
    from helpers.expr import Constant, Negated, Product, Query, Scaled, Sum
    if depth == 0 or rng.random() < 0.15:
        if rng.random() < 0.75:
            return Query(rng.randrange(ncols), rng.choice([0, 0, 8, -8, 16, -24]))
        return Constant(rng.randrange(p))
    k = rng.randrange(6)
    if k == 0:
        return Negated(rand_tree(rng, ncols, depth - 1, p))
    if k == 1:
        return Scaled(rand_tree(rng, ncols, depth - 1, p), rng.randrange(p))
    if k in (2, 3):
        return Sum(rand_tree(rng, ncols, depth - 1, p), rand_tree(rng, ncols, depth - 1, p))
    return Product(rand_tree(rng, ncols, depth - 1, p), rand_tree(rng, ncols, depth - 1, p))


@pytest.mark.parametrize("seed", range(8))
def test_random_expression_trees(gpu_ctx, seed):
    from helpers import expr
    F = O.FP
    rng = random.Random(seed)
    size, ncols = 256, 6
    cols = [[rng.randrange(F.p) for _ in range(size)] for _ in range(ncols)]
    tree = rand_tree(rng, ncols, 6, F.p)
    prog = expr.compile_expression(tree, F.p)
    got = C.array_to_ints(gpu_ctx.expr_eval(0, prog, [C.ints_to_array(c) for c in cols]))
    want = [expr.evaluate_tree(tree, cols, r, size, F.p) for r in range(size)]
    assert got == want


def test_num2bits_and_permutation_shaped_gates_folded_with_y(gpu_ctx):
    """Gate shapes of the reference folded Horner-style with a challenge y, as vanishing::construct does:
    bitify (src/chips/bitify.rs:63-88): bit*(1-bit), e2' - 2*e2, lc1' - lc1 - bit*e2;
    plus a permutation-product shaped term z(wX)*(a+beta*s+gamma) - z(X)*(a+beta*id+gamma)."""
    from helpers.expr import Constant, Product, Query, Sum, compile_expression, evaluate_tree
    F = O.FQ
    rng = random.Random(42)
    size, ext = 1 << 10, 8
    cols = [[rng.randrange(F.p) for _ in range(size)] for _ in range(7)]
    bit, lc1, e2, z, a, sig, idc = (Query(i) for i in range(7))
    nxt = lambda q: Query(q.column, ext)                       # Rotation::next() on the extended domain
    beta, gamma, y = (Constant(rng.randrange(F.p)) for _ in range(3))
    one = Constant(1)
    gates = [
        bit * (one - bit),
        nxt(e2) - e2 * 2,
        nxt(lc1) - lc1 - bit * e2,
        nxt(z) * (a + beta * sig + gamma) - z * (a + beta * idc + gamma),
    ]
    folded = gates[0]
    for g in gates[1:]:
        folded = folded * y + g
    prog = compile_expression(folded, F.p)
    got = C.array_to_ints(gpu_ctx.expr_eval(1, prog, [C.ints_to_array(c) for c in cols]))
    assert got == [evaluate_tree(folded, cols, r, size, F.p) for r in range(size)]


def test_program_validation(gpu_ctx):
    import bzh2
    from helpers import expr
    prog = expr.compile_expression(expr.Query(0) * expr.Query(3), O.P)   # column 3 does not exist below
    with pytest.raises(bzh2.BzhError):
        gpu_ctx.expr_eval(0, prog, [np.zeros((8, 4), dtype=np.uint64)])
