"""Host logic of the gate-expression compiler (tests/helpers/expr.py: compiles expression trees into the bzh_expr_op programs of the public bzh_expr_eval entry point): the
straight-line program interpreted on the CPU must equal direct tree evaluation, with challenges bound late through
Symbol leaves, within the evaluator's slot budget.  Mirrors halo2_proofs::plonk::Expression evaluation as used
by the reference's gates (src/chips/bitify.rs:63-88, src/chips/placement.rs:126-265)."""
import random

import pytest

P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001


def interpret(prog, consts, columns, row, size):
    from helpers import expr as X
    slots = [None] * X.MAX_SLOTS

    def operand(o):
        kind, idx, rot = o
        if kind == X.SLOT:
            return slots[idx]
        if kind == X.CONST:
            return consts[idx] % P
        return columns[idx][(row + rot) % size]
    for op, dst, a, b in prog.ops:
        va = operand(a)
        if op == X.NEG:
            v = -va % P
        elif op == X.COPY:
            v = va
        else:
            vb = operand(b)
            v = (va + vb) % P if op == X.ADD else ((va - vb) % P if op == X.SUB else va * vb % P)
        slots[dst] = v
    return slots[prog.result_slot]


def random_tree(rng, depth, ncols):
    from helpers import expr as X
    if depth == 0 or rng.random() < 0.15:
        r = rng.random()
        if r < 0.5:
            return X.Query(rng.randrange(ncols), rng.choice([-2, -1, 0, 0, 1, 3]))
        if r < 0.75:
            return X.Constant(rng.randrange(P))
        return X.Symbol(rng.choice(["theta", "beta", ("bd", 3)]))
    r = rng.random()
    if r < 0.1:
        return X.Negated(random_tree(rng, depth - 1, ncols))
    if r < 0.2:
        return X.Scaled(random_tree(rng, depth - 1, ncols), rng.randrange(P))
    a, b = random_tree(rng, depth - 1, ncols), random_tree(rng, depth - 1, ncols)
    return X.Sum(a, b) if r < 0.6 else X.Product(a, b)


@pytest.mark.parametrize("seed", range(6))
def test_compiled_program_matches_tree_evaluation_with_late_bound_symbols(seed):
    from helpers import expr as X
    rng = random.Random(seed)
    size, ncols = 16, 5
    cols = [[rng.randrange(P) for _ in range(size)] for _ in range(ncols)]
    tree = random_tree(rng, 7, ncols)
    prog = X.compile_expression(tree, P)
    assert all(o[1] < X.MAX_SLOTS for o in prog.ops)
    for trial in range(2):                       # one program, two bindings of the challenges
        env = {"theta": rng.randrange(P), "beta": rng.randrange(P), ("bd", 3): rng.randrange(P)}
        consts = prog.bind(env)
        for row in range(size):
            assert interpret(prog, consts, cols, row, size) == X.evaluate_tree(tree, cols, row, size, P, env)


def test_horner_chain_of_many_terms_stays_within_the_slot_budget():
    """The quotient numerator is a Horner chain in y over dozens of gate terms: depth grows, live intermediates must not."""
    from helpers import expr as X
    rng = random.Random(9)
    size = 8
    cols = [[rng.randrange(P) for _ in range(size)] for _ in range(4)]
    y = X.Symbol("y")
    acc = X.Product(X.Query(0), X.Query(1, 1))
    for i in range(200):
        term = X.Product(X.Query(i % 4), X.Sum(X.Query((i + 1) % 4, -1), X.Constant(i)))
        acc = X.Sum(X.Product(acc, y), term)
    prog = X.compile_expression(acc, P)
    env = {"y": rng.randrange(P)}
    assert max(o[1] for o in prog.ops) < 4       # a Horner chain needs a handful of slots, whatever its length
    for row in range(size):
        assert interpret(prog, prog.bind(env), cols, row, size) == X.evaluate_tree(acc, cols, row, size, P, env)


def test_serialized_circuit_blob_layout():
    """bzh2.circuit_data.serialize_circuit: the header and section counts of the blob bzh_pk_create parses (csrc/prove.hip)."""
    import halo2_oracle  # noqa: F401  (oracle path on sys.path via conftest)
    import sample_circuit as S
    from helpers import prover as PR
    cs, fixed, copies, adv, inst = S.build(k=4, seed=1, with_lookup=True)
    circ = PR.Circuit(cs.k, cs.num_advice, cs.num_fixed, cs.num_instance, cs.gates, cs.perm_columns, cs.lookups, fixed, copies)
    blob = PR.serialize_circuit(circ, P, vk_repr=7)
    u32 = lambda o: int.from_bytes(blob[o:o + 4], "little")
    assert blob[:4] == b"BZC1"
    assert [u32(4 + 4 * i) for i in range(5)] == [cs.k, cs.num_advice, cs.num_fixed, cs.num_instance, circ.degree]
    assert int.from_bytes(blob[24:56], "little") == 7
    assert u32(56) == len(cs.gates)
    # the fixed columns close the blob: num_fixed x (u32 len, len x 32 bytes)
    tail = sum(4 + 32 * min(len(col), circ.n) for col in fixed)
    o = len(blob) - tail
    for col in fixed:
        ln = u32(o)
        assert ln == min(len(col), circ.n)
        assert int.from_bytes(blob[o + 4:o + 36], "little") == col[0] % P
        o += 4 + 32 * ln
    assert o == len(blob)
