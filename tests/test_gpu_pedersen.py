"""Row a8 / f4: batched native Pedersen commitments on the GPU against the oracle's restatement of
src/utils/pedersen.rs:17-28 (whose generators are pinned by the reference's `generator` KATs)."""
import json
import os
import random

import pytest

import pasta as O

pytestmark = pytest.mark.gpu


def test_generator_constants_are_the_hashed_points():
    from bzh2 import game as G
    assert O.hash_to_curve("pallas", "battlezips:hash2curve", b"v") == G.PEDERSEN_V
    assert O.hash_to_curve("pallas", "battlezips:hash2curve", b"r") == G.PEDERSEN_R
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fixed_bases.json")))
    assert tuple(int(x, 16) for x in fx["bases"]["v"]["generator"]) == G.PEDERSEN_V
    assert tuple(int(x, 16) for x in fx["bases"]["r"]["generator"]) == G.PEDERSEN_R


def test_pedersen_commit_batch_matches_oracle(gpu_ctx):
    from bzh2 import game as G
    rng = random.Random(8)
    boards = [G.Board.from_(G.Deck.from_([(3, 3, True), (5, 4, False), (0, 1, False), (0, 5, True), (6, 1, False)])),
              G.Board.from_(G.Deck.from_([(3, 4, False), (9, 6, True), (0, 0, False), (0, 6, False), (6, 1, True)]))]
    msgs = [b.state().to_fp() for b in boards] + [0, 1, rng.randrange(1 << 100)] + [rng.randrange(O.Q) for _ in range(27)]
    traps = [rng.randrange(O.Q) for _ in msgs]
    traps[2] = 0                                          # m = 0, t = 0 -> identity
    c = G.PedersenCommitter(gpu_ctx)
    try:
        got = c.commit_batch(msgs, traps)
        assert got == [O.pedersen_commit(m, t) for m, t in zip(msgs, traps)]
        assert got[2] is None
        with pytest.raises(ValueError):                   # Fp value >= q cannot be re-read as Fq (unwrap panics upstream)
            c.commit_batch([O.Q], [1])
    finally:
        c.close()
    assert G.pedersen_commit(gpu_ctx, msgs[0], traps[0]) == got[0]


def _oracle_commitments(oracle_c, msgs, traps):
    """[m]V + [t]R through the C oracle's naive double-and-add MSM on Pallas (V, R pinned by the reference's `generator` KATs)"""
    import numpy as np
    from bzh2 import game as G
    bases = oracle_c.points_to_array([G.PEDERSEN_V, G.PEDERSEN_R])
    out = []
    for m, t in zip(msgs, traps):
        out.append(oracle_c.array_to_point(oracle_c.msm_naive(1, oracle_c.ints_to_array([m, t]), bases)))
    return out


@pytest.mark.parametrize("n", [1, 30, 2816])
def test_pedersen_commit_batch_c_entry_point_matches_the_oracle(gpu_ctx, oracle_c, n):
    """bzh_pedersen_commit_batch (src/utils/pedersen.rs:17-28 batched, called twice per Shot prove:
    src/chips/shot.rs:319) at one commitment, a wasm-call-sized handful and BASELINE configs[3]'s 2 816 proofs, against the
    oracle and the product's own host path.  Edge scalars: 0 (identity with t = 0), 1, q - 1, all-0xff-byte runs and 0x80
    bytes (the signed-digit recoding's carry chains and its |d| = 128 digit), a 100-bit board state."""
    from bzh2 import circuits as Cm
    rng = random.Random(1000 + n)
    q = O.Q
    edge = [(0, 0), (1, 0), (0, 1), (q - 1, q - 1), (int.from_bytes(b"\xff" * 31 + b"\x3f", "little") % q, 1),
            (int.from_bytes(b"\x80" * 31 + b"\x00", "little"), int.from_bytes(b"\x81\x7f" * 15 + b"\x80\x00", "little")),
            ((1 << 100) - 1, rng.randrange(q)), (0x2409025e80200031c00, rng.randrange(q))]
    pairs = (edge + [(rng.randrange(q), rng.randrange(q)) for _ in range(max(n - len(edge), 0))])[:n] if n > 1 else [(0x2409025e80200031c00, rng.randrange(q))]
    msgs, traps = [m for m, _ in pairs], [t for _, t in pairs]
    got = Cm.pedersen_commit_batch(gpu_ctx, msgs, traps)
    want = _oracle_commitments(oracle_c, msgs, traps)
    assert got == want
    if n > 1:
        assert got[0] is None
    for i in sorted({0, n // 2, n - 1}):
        if got[i] is not None:
            assert Cm.pedersen_commit_host(msgs[i] % O.P, traps[i]) == got[i] or msgs[i] >= O.P
    assert Cm.pedersen_commit_batch(gpu_ctx, msgs[-1:], traps[-1:]) == got[-1:]     # the table is reused; batch-size independent


def test_pedersen_commit_batch_refuses_non_canonical_scalars(gpu_ctx):
    import bzh2
    from bzh2 import circuits as Cm
    for m, t in ((O.Q, 1), (1, O.Q), ((1 << 256) - 1, 1)):
        with pytest.raises(bzh2.BzhError) as e:
            Cm.pedersen_commit_batch(gpu_ctx, [5, m], [7, t])
        assert e.value.status == -4
