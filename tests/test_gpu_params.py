"""Params::new on the GPU box (csrc/params.hip): g_lagrange from the device group FFT against its definition, the
on-disk cache, and the Lagrange-basis commitments of bzh_prove_batch -- the same proof bytes as coefficient-basis
commitments and as the oracle prover's, now on the REAL halo2 SRS (hash_to_curve("Halo2-Parameters")).
Reference call sites: Params::new benches/shot.rs:58, benches/board.rs:51."""
import random

import numpy as np
import pytest

import blob as B
import coracle as C
import halo2_oracle as H
import pasta as O

pytestmark = pytest.mark.gpu


def _pts(arr):
    return [C.array_to_point(arr[i]) for i in range(arr.shape[0])]


@pytest.mark.parametrize("k", [1, 4, 7])
def test_g_lagrange_matches_its_definition(gpu_ctx, oracle_c, k):
    """g_lagrange[i] = sum_j c_ij g[j] with c_i = the coefficients of the i-th Lagrange basis polynomial (the inverse
    DFT of the unit vector e_i): the committed point of a column is the same in either basis."""
    from bzh2 import params as Pm
    cv, F = O.VESTA, O.FP
    g, w, u = Pm.generators(k)
    gl = Pm.group_ifft(gpu_ctx, g)
    n = 1 << k
    omega = F.omega(k)
    for i in sorted({0, 1, n // 2, n - 1}):
        e = [0] * n
        e[i] = 1
        coeffs = O.intt(e, omega, F)
        want = C.array_to_point(C.msm(0, C.ints_to_array(coeffs), g, 1))
        assert C.array_to_point(gl[i]) == want, i


def test_params_cache_roundtrip_and_k11_linearity(gpu_ctx, oracle_c, tmp_path):
    """bzh_params_create at the reference's k = 11: second construction comes from the cache file with identical points;
    commit_lagrange(evals) == commit(coefficients) for a random column (C oracle MSMs over the two bases)."""
    from bzh2 import params as Pm
    F = O.FP
    k = 11
    p1 = Pm.Params(gpu_ctx, k, cache_dir=str(tmp_path))
    g1, gl1, w1, u1, cached1 = p1.points()
    p1.close()
    p2 = Pm.Params(gpu_ctx, k, cache_dir=str(tmp_path))
    g2, gl2, w2, u2, cached2 = p2.points()
    p2.close()
    assert not cached1 and cached2
    assert (g1 == g2).all() and (gl1 == gl2).all() and (w1, u1) == (w2, u2)
    assert w1 == O.hash_to_curve("vesta", "Halo2-Parameters", bytes([1])) and u1 == O.hash_to_curve("vesta", "Halo2-Parameters", bytes([2]))
    rng = random.Random(5)
    evals = [rng.randrange(F.p) for _ in range(1 << k)]
    coeffs = C.array_to_ints(C.ntt(0, C.ints_to_array(evals), F.omega(k), inverse=True, threads=8))
    a = C.array_to_point(C.msm(0, C.ints_to_array(evals), gl1, 8))
    b = C.array_to_point(C.msm(0, C.ints_to_array(coeffs), g1, 8))
    assert a == b and a is not None


def test_proofs_on_the_real_srs_lagrange_equals_coefficient_commitments_and_oracle(gpu_ctx, oracle_c):
    """A real circuit of the reference (bitify test circuit, k = 6) on Params::new(6): bzh_prove_batch with Lagrange-basis
    commitments (bzh_pk_set_lagrange) emits the bytes of the coefficient-basis run and of the oracle prover."""
    import bzh2
    from bzh2 import circuits as Cm, native as N, params as Pm
    from bzh2.game import BinaryValue
    cv, F = O.VESTA, O.FP
    k, bits = 6, 20
    lay = Cm.CircuitLayout(Cm.NUM2BITS_TEST, k, bits)
    prm = Pm.Params(gpu_ctx, k, cache_dir="")
    try:
        blob = lay.blob()
        circ = B.decode(blob)
        g_arr, _, w, u, _ = prm.points(want_lagrange=False)
        g = _pts(g_arr)
        cs = H.ConstraintSystem(circ.k, circ.num_advice, circ.num_fixed, circ.num_instance, circ.gates, circ.perm_columns, circ.lookups,
                                degree=circ.min_degree, queries=circ.queries)
        keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, circ.fixed, circ.copies, vk_repr=circ.vk_repr)
        pk = N.NativeProvingKey(gpu_ctx, blob, bzh2.CURVE_VESTA, params=prm)
        try:
            rng = random.Random(66)
            value = rng.getrandbits(bits)
            adv = lay.synthesize_bitify_test(value, BinaryValue(value))
            rbytes = bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes))
            rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(pk.rng_bytes // 64)]
            want = H.create_proof(keys, [C.array_to_ints(adv[0, c]) for c in range(adv.shape[1])], [], rs, O.Blake2bTranscript(F))
            got_lagrange = pk.prove_batch(adv, [[]], [rbytes])
            pk.set_lagrange(None)
            got_coeff = pk.prove_batch(adv, [[]], [rbytes])
            assert got_lagrange == got_coeff == [want]
            assert pk.verify_batch([[]], got_lagrange) == [True]
        finally:
            pk.close()
    finally:
        prm.close()
        lay.close()


def test_shot_circuit_on_params_new_11(gpu_ctx, oracle_c, tmp_path):
    """The reference's `production` flow with its own SRS: Params::new(11), ShotCircuit, create_proof (Lagrange commits),
    verify_proof -- native verifier on a batch, and identical bytes with coefficient-basis commitments."""
    import bzh2
    from bzh2 import circuits as Cm, native as N, params as Pm
    from bzh2.game import BinaryValue
    lay = Cm.CircuitLayout(Cm.SHOT, 11)
    prm = Pm.Params(gpu_ctx, 11, cache_dir=str(tmp_path))
    pk = N.NativeProvingKey(gpu_ctx, lay.blob(), bzh2.CURVE_VESTA, params=prm)
    try:
        rng = random.Random(7)
        _, state = Cm.board_witness([(3, 3, True), (5, 4, False), (0, 1, False), (0, 5, True), (6, 1, False)], None)
        circuits = [Cm.ShotCircuit(state, rng.randrange(O.FQ.p), Cm.shot_serialize([3], [5]), BinaryValue.from_u8(1)),
                    Cm.ShotCircuit(state, rng.randrange(O.FQ.p), Cm.shot_serialize([4], [3]), BinaryValue.from_u8(0))]
        adv, insts = lay.synthesize(circuits)
        rbs = [rng.randbytes(pk.rng_bytes) for _ in circuits]
        proofs = pk.prove_batch(adv, insts, rbs)
        assert pk.verify_batch(insts, proofs) == [True, True]
        pk.set_lagrange(None)
        assert pk.prove_batch(adv, insts, rbs) == proofs
    finally:
        pk.close()
        prm.close()
        lay.close()
