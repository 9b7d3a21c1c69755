"""One process = one setting of the library's tuning / fallback environment variables (they are read once per process).
Prints a digest of (a) a window-table MSM and a paired-free plain MSM on fixed inputs, (b) the proofs of two fixed
ShotCircuit witnesses under fixed seeds.  tests/test_gpu_env_paths.py compares the digests across settings: every path must
produce the same group elements and the same proof bytes."""
import hashlib
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "battlezips-halo2_amd"))

import numpy as np  # noqa: E402

import bzh2  # noqa: E402
from bzh2 import circuits as Cm, native as N, params as Pm  # noqa: E402
from bzh2.game import BinaryValue  # noqa: E402


def main():
    ctx = bzh2.Context(0)
    h = hashlib.sha256()
    # (a) MSM: 2^13 SRS points, 3 scalar vectors, window table (the prover's regime) and plain bases
    k = 11
    prm = Pm.Params(ctx, k)
    rng = np.random.default_rng(5)
    n = 1 << k
    sc = rng.integers(0, 1 << 63, size=(3, n + 2, 4), dtype=np.uint64)
    sc[..., 3] &= (1 << 60) - 1
    out = ctx.msm(prm.bases, sc)
    aff = bzh2.jacobian_to_affine(bzh2.CURVE_VESTA, out)
    h.update(np.ascontiguousarray(aff).tobytes())
    # (b) two real ShotCircuit proofs, fixed witnesses and seeds
    lay = Cm.CircuitLayout(Cm.SHOT, k)
    pk = N.NativeProvingKey(ctx, lay.blob(), bzh2.CURVE_VESTA, params=prm)
    r = random.Random(77)
    deck = [(3, 3, True), (5, 4, False), (0, 1, False), (0, 5, True), (6, 1, False)]
    _, state = Cm.board_witness(deck, None)
    circuits = [Cm.ShotCircuit(state, r.randrange(1 << 250), Cm.shot_serialize([3], [5]), BinaryValue.from_u8(1)),
                Cm.ShotCircuit(state, r.randrange(1 << 250), Cm.shot_serialize([9], [9]), BinaryValue.from_u8(0))]
    adv, insts = lay.synthesize(circuits)
    proofs = pk.prove_batch(adv, insts, None, seeds=[bytes([i]) * 32 for i in (1, 2)])
    assert pk.verify_batch(insts, proofs) == [True, True]
    for p in proofs:
        h.update(p)
    # (c) a batch of 8 (the size from which the opening collapses its generators by itself), fresh trapdoors
    fleet = {(3, 3), (3, 4), (3, 5), (3, 6), (3, 7), (5, 4), (6, 4), (7, 4), (8, 4), (0, 1), (1, 1), (2, 1), (0, 5), (0, 6), (0, 7), (6, 1), (7, 1)}
    circuits8 = [Cm.ShotCircuit(state, r.randrange(1 << 250), Cm.shot_serialize([i], [9 - i]), BinaryValue.from_u8(1 if (i, 9 - i) in fleet else 0))
                 for i in range(8)]
    adv8, insts8 = lay.synthesize(circuits8)
    proofs8 = pk.prove_batch(adv8, insts8, None, seeds=[bytes([40 + i]) * 32 for i in range(8)])
    assert pk.verify_batch(insts8, proofs8) == [True] * 8
    for p in proofs8:
        h.update(p)
    print("DIGEST", h.hexdigest())


if __name__ == "__main__":
    main()
