#!/usr/bin/env python3
"""Write a case file for examples/prove_batch.cpp: SRS, circuit blob (bzh2.prover.serialize_circuit), witness, instances
and randomness of a BattleZips-shaped circuit.  Usage: export_case.py out.bin [k] [batch]   (no GPU needed)."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "battlezips-halo2_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))   # only for cheap SRS points (generator walk); not a product dependency

import numpy as np  # noqa: E402


def fnv1a(proofs) -> str:
    h = 1469598103934665603
    for pr in proofs:
        for b in pr:
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def build(k: int, batch: int, seed: int = 7):
    import random
    import coracle as C
    import pasta as O
    from bzh2 import prover as P, synth
    circ, adv, inst = synth.battlezips_shaped(k, seed)
    n = 1 << k
    g0 = O.VESTA.random_point(random.Random(seed))
    srs = C.point_walk(0, C.points_to_array([g0])[0], n + 2)              # G_i = [i + 1] G_0; last two play U and W
    p = P.MODULI[0]
    blob = P.serialize_circuit(circ, p)
    advice = np.stack([np.stack([C.ints_to_array(list(col) + [0] * (n - len(col))) for col in adv])] * batch)
    rows = max(len(c) for c in inst)
    instances = np.zeros((batch, len(inst), rows, 4), dtype=np.uint64)
    for b in range(batch):
        for i, col in enumerate(inst):
            instances[b, i, :len(col)] = C.ints_to_array([v % p for v in col])
    return circ, srs, blob, advice, instances, rows


def main():
    out = sys.argv[1]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    circ, srs, blob, advice, instances, rows = build(k, batch)
    n = 1 << k
    draws = circ.num_advice * (circ.blinding_factors + 2) + 3 * n + 4096    # generous: bzh_pk_info reports the exact need
    rng_stride = 64 * draws
    rng = np.random.default_rng(99).bytes(batch * rng_stride)
    parts = [np.ascontiguousarray(srs).tobytes(), blob, np.ascontiguousarray(advice).tobytes(), np.ascontiguousarray(instances).tobytes(), rng]
    with open(out, "wb") as f:
        f.write(b"BZX1")
        f.write(struct.pack("<5I", k, circ.num_advice, circ.num_instance, rows, batch))
        f.write(struct.pack("<Q", rng_stride))
        f.write(struct.pack("<5Q", *[len(x) for x in parts]))
        for x in parts:
            f.write(x)
    print("wrote", out, sum(len(x) for x in parts), "bytes")


if __name__ == "__main__":
    main()
