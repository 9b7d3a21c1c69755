"""bzh_pk_create / bzh_prove_batch (csrc/prove.hip): the whole create_proof behind the C ABI, proving a batch of
witnesses in lockstep.  Every proof must be byte-identical to the big-int oracle prover's (oracle/halo2_oracle.py)
under that proof's randomness stream.  Reference call sites: create_proof benches/shot.rs:68,
benches/board.rs:61-68; keygen_pk benches/shot.rs:60, benches/board.rs:53."""
import random

import numpy as np
import pytest

import coracle as C
import halo2_oracle as H
import pasta as O
import sample_circuit as S

pytestmark = pytest.mark.gpu


def _setup(cs, seed):
    cv = O.VESTA
    rng = random.Random(seed)
    g = [cv.random_point(rng) for _ in range(cs.n)]
    return rng, g, cv.random_point(rng), cv.random_point(rng)


def _adv_array(adv, n):
    return np.stack([C.ints_to_array(list(col) + [0] * (n - len(col))) for col in adv])


@pytest.mark.parametrize("k,with_lookup,degree,batch", [(4, False, None, 1), (5, True, None, 3), (6, True, 9, 2)])
def test_native_prove_batch_matches_oracle(gpu_ctx, oracle_c, k, with_lookup, degree, batch):
    import bzh2
    from bzh2 import native as N, prover as P
    cv, F = O.VESTA, O.FP
    cases = [S.build(k=k, seed=500 + 11 * b + k, with_lookup=with_lookup, degree=degree) for b in range(batch)]
    cs, fixed, copies = cases[0][:3]
    rng, g, w, u = _setup(cs, 4000 + k)
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, fixed, copies)
    circ = P.Circuit(cs.k, cs.num_advice, cs.num_fixed, cs.num_instance, cs.gates, cs.perm_columns, cs.lookups, fixed, copies,
                     degree=degree)
    pk = N.NativeProvingKey(gpu_ctx, circ, bzh2.CURVE_VESTA, g, w, u)
    try:
        assert (pk.n, pk.usable_rows, pk.num_advice) == (cs.n, cs.usable_rows, cs.num_advice)
        ndraws = pk.rng_bytes // 64
        rbs, want = [], []
        for b in range(batch):
            rbytes = bytes(rng.getrandbits(8) for _ in range(64 * ndraws))
            rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(ndraws)]
            want.append(H.create_proof(keys, cases[b][3], cases[b][4], rs, O.Blake2bTranscript(F)))
            rbs.append(rbytes)
        adv = np.stack([_adv_array(cse[3], cs.n) for cse in cases])
        for _ in range(2):                      # the second call runs on cached programs and a warm arena
            got = pk.prove_batch(adv, [cse[4] for cse in cases], rbs)
            assert got == want
        for b in range(batch):
            assert H.verify_proof(keys, cases[b][4], got[b], O.Blake2bTranscript(F))
            assert len(got[b]) <= pk.max_proof_bytes
    finally:
        pk.close()


def test_native_prover_battlezips_shaped_and_unsatisfied_witness(gpu_ctx, oracle_c):
    """The benchmark circuit (bzh2/synth.py) at k = 7 through the native entry point; a broken witness must come back
    as BZH_E_RANGE (surplus quotient coefficients) or as a proof the oracle verifier rejects."""
    import bzh2
    from bzh2 import native as N, synth
    cv, F = O.VESTA, O.FP
    built = [synth.battlezips_shaped(7, seed=60 + b) for b in range(2)]
    circ = built[0][0]
    cs = H.ConstraintSystem(circ.k, 11, 8, 1, circ.gates, circ.perm_columns, circ.lookups, degree=9)
    rng, g, w, u = _setup(cs, 79)
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, circ.fixed, circ.copies)
    pk = N.NativeProvingKey(gpu_ctx, circ, bzh2.CURVE_VESTA, g, w, u)
    try:
        ndraws = pk.rng_bytes // 64
        rbs, want = [], []
        for b in range(2):
            rbytes = bytes(rng.getrandbits(8) for _ in range(64 * ndraws))
            rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(ndraws)]
            want.append(H.create_proof(keys, built[b][1], built[b][2], rs, O.Blake2bTranscript(F)))
            rbs.append(rbytes)
        adv = np.stack([_adv_array(built[b][1], cs.n) for b in range(2)])
        assert pk.prove_batch(adv, [built[b][2] for b in range(2)], rbs) == want
        bad = adv.copy()
        bad[1, 2, 0, 0] ^= 1                   # a2 of a multiplication row of the second witness: a0 * a1 != a2
        try:
            proofs = pk.prove_batch(bad, [built[b][2] for b in range(2)], rbs)
        except bzh2.BzhError as e:
            assert e.status == bzh2.E_RANGE
        else:
            assert proofs[0] == want[0]
            assert not H.verify_proof(keys, built[1][2], proofs[1], O.Blake2bTranscript(F))
    finally:
        pk.close()
