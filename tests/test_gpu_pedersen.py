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
