"""oracle/accel.py hands the oracle prover's bulk arithmetic to the C oracle so that it finishes the reference's real
circuits (tests/test_gpu_real_circuit_parity.py).  Here: the accelerated prover emits the big-int prover's bytes, and the
accelerated verifier agrees, on circuits small enough for the big-int code (gates, permutation, lookup, several
permutation sets)."""
import random

import pytest

import accel as A
import halo2_oracle as H
import pasta as O
import sample_circuit as S


@pytest.mark.parametrize("k,seed,lookup", [(4, 3, True), (5, 4, True), (5, 5, False)])
def test_accelerated_oracle_prover_emits_the_bigint_provers_bytes(oracle_c, k, seed, lookup):
    cv, F = O.VESTA, O.FP
    cs, fixed, copies, advice, instance = S.build(k=k, seed=seed, with_lookup=lookup)
    r = random.Random(seed)
    g = [cv.random_point(r) for _ in range(cs.n)]
    w, u = cv.random_point(r), cv.random_point(r)
    rs = [r.randrange(F.p) for _ in range(40 * cs.n)]
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, fixed, copies)
    want = H.create_proof(keys, advice, instance, iter(rs), O.Blake2bTranscript(F))
    with A.accelerated(threads=2):
        keys2 = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, fixed, copies)
        got = H.create_proof(keys2, advice, instance, iter(rs), O.Blake2bTranscript(F))
        assert H.verify_proof(keys2, instance, got, O.Blake2bTranscript(F))
    assert keys2.fixed_cosets == keys.fixed_cosets and keys2.l_blind == keys.l_blind
    assert got == want
    assert H.verify_proof(keys, instance, got, O.Blake2bTranscript(F))
    # the swap is undone on exit
    assert H.quotient_evals.__module__ == "halo2_oracle" and O.fold_bases.__module__ == "pasta"
