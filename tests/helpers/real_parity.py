"""Shared by tests/test_gpu_real_circuit_parity.py and tests/golden/make_real_proof_golden.py: the ORACLE side of a
byte comparison of the reference's real circuits (src/circuits/shot.rs:880-941, src/circuits/board.rs:879-933,
benches/shot.rs:22,68, benches/board.rs:22,61-68) -- oracle keys from a BZC2 circuit blob, the shared randomness
stream, deterministic witnesses.  Test infrastructure: imports oracle/."""
import hashlib
import os
import random

import accel as A
import blob as B
import coracle as C
import halo2_oracle as H
import pasta as O

THREADS = max(1, min(32, os.cpu_count() or 1))
PATTERN_1 = [(3, 3, True), (5, 4, False), (0, 1, False), (0, 5, True), (6, 1, False)]      # src/circuits/board.rs:101-107
PATTERN_2 = [(3, 4, False), (9, 6, True), (0, 0, False), (0, 6, False), (6, 1, True)]     # src/circuits/board.rs:134-140


def rng_stream(tag: str, nbytes: int) -> bytes:
    """the randomness both provers consume (the reference draws it from OsRng inside create_proof): SHAKE-256 of a tag"""
    return hashlib.shake_256(tag.encode()).digest(nbytes)


def stream_scalars(stream: bytes, F=O.FP):
    """upstream's Field::random: 64 bytes -> 512-bit little-endian integer reduced mod p, one draw per 64 bytes"""
    return (O.from_u512(stream[o:o + 64], F) for o in range(0, len(stream) - 63, 64))


def points_of(g_arr):
    return [C.array_to_point(g_arr[i]) for i in range(g_arr.shape[0])]


def oracle_keys(blob: bytes, g, w, u, verifier_only=False):
    """call inside `with accel.accelerated(...)`: the keys' NTTs / commitments go through the C oracle"""
    circ = B.decode(blob)
    cs = H.ConstraintSystem(circ.k, circ.num_advice, circ.num_fixed, circ.num_instance, circ.gates, circ.perm_columns, circ.lookups,
                            degree=circ.min_degree, queries=circ.queries)
    return H.Keys(cs, H.Domain(cs, O.FP), O.VESTA, g, w, u, circ.fixed, circ.copies, vk_repr=circ.vk_repr, verifier_only=verifier_only)


def oracle_prove(keys, adv_one, inst, stream: bytes) -> bytes:
    """adv_one: (columns, n, 4) canonical limbs of ONE proof's advice table (as bzh_synthesize_* wrote it)"""
    cols = [C.array_to_ints(adv_one[c]) for c in range(adv_one.shape[0])]
    return H.create_proof(keys, cols, inst, stream_scalars(stream), O.Blake2bTranscript(O.FP))


def random_deck(rng):
    """a valid random fleet: rejection-sample non-overlapping in-bounds placements"""
    lens = (5, 4, 3, 3, 2)
    while True:
        used, deck = set(), []
        for L in lens:
            for _ in range(200):
                z = rng.random() < 0.5
                x, y = rng.randrange(10 - (0 if z else L - 1)), rng.randrange(10 - (L - 1 if z else 0))
                cells = {(x, y + i) if z else (x + i, y) for i in range(L)}
                if not cells & used:
                    used |= cells
                    deck.append((x, y, z))
                    break
            else:
                break
        if len(deck) == 5:
            return deck, used


def board_circuits(Cm, seed, count):
    rng = random.Random(seed)
    out = []
    for i in range(count):
        deck = PATTERN_1 if i == 0 else PATTERN_2 if i == 1 else random_deck(rng)[0]
        ships, state = Cm.board_witness(deck, None)
        out.append(Cm.BoardCircuit(ships, state, rng.randrange(O.FQ.p)))
    return out


def shot_circuits(Cm, seed, count):
    from bzh2.game import BinaryValue
    rng = random.Random(seed)
    out = []
    for _ in range(count):
        deck, used = random_deck(rng)
        _, state = Cm.board_witness(deck, None)
        x, y = rng.randrange(10), rng.randrange(10)
        out.append(Cm.ShotCircuit(state, rng.randrange(O.FQ.p), Cm.shot_serialize([x], [y]), BinaryValue.from_u8(1 if (x, y) in used else 0)))
    return out


def accelerated_oracle(fn):
    """Decorator for tests whose oracle side (keys, create_proof, verify_proof of oracle/halo2_oracle.py) would spend tens of
    seconds in big-int group arithmetic: run the test inside `accel.accelerated(THREADS)`, i.e. with the oracle's MSMs, NTTs,
    Horner sums and generator folds handed to the C oracle.  tests/test_oracle_accel_cpu.py pins the accelerated oracle to the
    big-int one byte for byte; every protocol decision stays in halo2_oracle."""
    import functools

    import accel

    @functools.wraps(fn)
    def run(*a, **kw):
        with accel.accelerated(THREADS):
            return fn(*a, **kw)
    return run
