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
from helpers.real_parity import accelerated_oracle

pytestmark = pytest.mark.gpu


def _setup(cs, seed):
    cv = O.VESTA
    rng = random.Random(seed)
    g = [cv.random_point(rng) for _ in range(cs.n)]
    return rng, g, cv.random_point(rng), cv.random_point(rng)


def _adv_array(adv, n):
    return np.stack([C.ints_to_array(list(col) + [0] * (n - len(col))) for col in adv])


@pytest.mark.parametrize("k,with_lookup,degree,batch", [(4, False, None, 1), (5, True, None, 3), (6, True, 9, 2)])
@accelerated_oracle
def test_native_prove_batch_matches_oracle(gpu_ctx, oracle_c, k, with_lookup, degree, batch):
    import bzh2
    from bzh2 import native as N, circuit_data as P
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


@accelerated_oracle
def test_native_prover_battlezips_shaped_and_unsatisfied_witness(gpu_ctx, oracle_c):
    """The benchmark circuit (tests/helpers/synth.py) at k = 7 through the native entry point; a broken witness must come back
    as BZH_E_RANGE (surplus quotient coefficients) or as a proof the oracle verifier rejects."""
    import bzh2
    from bzh2 import native as N
    from helpers import synth
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


@pytest.mark.parametrize("k,with_lookup,degree", [(4, False, None), (5, True, None), (6, True, 9)])
@accelerated_oracle
def test_native_verify_batch_agrees_with_oracle_verifier(gpu_ctx, oracle_c, k, with_lookup, degree):
    """bzh_verify_batch (verify_proof, benches/board.rs:80-86): accepts exactly what the big-int oracle verifier accepts --
    valid proofs (made by the ORACLE prover, so prover and verifier here are independent), and rejects a wrong instance,
    flipped bytes in every section of the proof, a truncated proof and a non-canonical scalar."""
    import bzh2
    from bzh2 import native as N, circuit_data as P
    cv, F = O.VESTA, O.FP
    cs, fixed, copies, adv, inst = S.build(k=k, seed=900 + k, with_lookup=with_lookup, degree=degree)
    rng, g, w, u = _setup(cs, 5000 + k)
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, fixed, copies)
    circ = P.Circuit(cs.k, cs.num_advice, cs.num_fixed, cs.num_instance, cs.gates, cs.perm_columns, cs.lookups, fixed, copies,
                     degree=degree)
    pk = N.NativeProvingKey(gpu_ctx, circ, bzh2.CURVE_VESTA, g, w, u)
    try:
        ndraws = pk.rng_bytes // 64
        rs = [rng.randrange(F.p) for _ in range(ndraws)]
        good = H.create_proof(keys, adv, inst, rs, O.Blake2bTranscript(F))
        assert H.verify_proof(keys, inst, good, O.Blake2bTranscript(F))
        cases = [(inst, good)]
        cases.append(([[(inst[0][0] + 1) % F.p]], good))                       # wrong public input
        step = max(1, len(good) // 12)
        for pos in range(5, len(good), step):                                   # one flipped bit per section
            cases.append((inst, good[:pos] + bytes([good[pos] ^ 0x04]) + good[pos + 1:]))
        cases.append((inst, good[:-32]))                                        # truncated
        cases.append((inst, good[:-32] + (F.p + 1).to_bytes(32, "little")))     # non-canonical final scalar
        cases.append((inst, good + b"\x00" * 32))                               # trailing bytes
        cases.append((inst, bytes(32) + good[32:]))                             # an identity commitment: upstream's transcript refuses it
        ipa_start = len(good) - 32 * (2 * k + 3)
        cases.append((inst, good[:ipa_start] + bytes(32) + good[ipa_start + 32:]))   # identity as the IPA's S
        got = pk.verify_batch([c[0] for c in cases], [c[1] for c in cases])
        want = [H.verify_proof(keys, c[0], c[1], O.Blake2bTranscript(F)) for c in cases]
        assert got == want
        assert got[0] is True and not any(got[1:])
    finally:
        pk.close()


def test_native_prove_then_verify_roundtrip_shaped(gpu_ctx, oracle_c):
    """The benchmark circuit at k = 8: bzh_prove_batch -> bzh_verify_batch accepts all; swapping instances rejects."""
    import bzh2
    from bzh2 import native as N
    from helpers import synth
    built = [synth.battlezips_shaped(8, seed=80 + b) for b in range(3)]
    circ = built[0][0]
    cs = H.ConstraintSystem(circ.k, 11, 8, 1, circ.gates, circ.perm_columns, circ.lookups, degree=9)
    rng, g, w, u = _setup(cs, 81)
    pk = N.NativeProvingKey(gpu_ctx, circ, bzh2.CURVE_VESTA, g, w, u)
    try:
        rbs = [bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes)) for _ in range(3)]
        adv = np.stack([_adv_array(built[b][1], cs.n) for b in range(3)])
        insts = [built[b][2] for b in range(3)]
        proofs = pk.prove_batch(adv, insts, rbs)
        assert pk.verify_batch(insts, proofs) == [True, True, True]
        if insts[0] != insts[1]:
            assert pk.verify_batch([insts[1], insts[0], insts[2]], proofs) == [False, False, True]
        # G_0 / U / W of another SRS are an argument error, not a silent rejection of every proof
        good = pk._g0_u_w.copy()
        pk._g0_u_w[1], pk._g0_u_w[2] = good[2].copy(), good[1].copy()
        with pytest.raises(bzh2.BzhError) as e:
            pk.verify_batch(insts, proofs)
        assert e.value.status == bzh2.E_ARG
        pk._g0_u_w[:] = good
        assert pk.verify_batch(insts, proofs) == [True, True, True]
    finally:
        pk.close()


def test_native_prover_device_resident_witness_and_error_paths(gpu_ctx, oracle_c):
    """bench.py's path: the witness tensor resident in HBM in Montgomery form (BZH_MEM_DEVICE) gives the same bytes as the
    host canonical path; a short randomness stream is BZH_E_ARG; a lookup input outside its table is BZH_E_RANGE."""
    import torch
    import bzh2
    from bzh2 import native as N, circuit_data as P
    from helpers.device import DeviceOps
    cv, F = O.VESTA, O.FP
    cs, fixed, copies, adv, inst = S.build(k=5, seed=31, with_lookup=True)
    rng, g, w, u = _setup(cs, 6001)
    circ = P.Circuit(cs.k, cs.num_advice, cs.num_fixed, cs.num_instance, cs.gates, cs.perm_columns, cs.lookups, fixed, copies)
    ctx = bzh2.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    pk = N.NativeProvingKey(ctx, circ, bzh2.CURVE_VESTA, g, w, u)
    try:
        rbs = [bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes)) for _ in range(2)]
        host = np.stack([_adv_array(adv, cs.n)] * 2)
        want = pk.prove_batch(host, [inst, inst], rbs)
        assert want[0] != want[1]                                       # same witness, different blinding randomness
        ops = DeviceOps(ctx, pk.field, pk.curve, pk.p, torch.device("cuda", 0))
        dev = torch.stack([torch.stack([ops.upload(list(col) + [0] * (cs.n - len(col))) for col in adv])] * 2).contiguous()
        torch.cuda.synchronize()
        assert pk.prove_batch(None, [inst, inst], rbs, device_ptr=dev.data_ptr()) == want
        assert pk.verify_batch([inst, inst], want) == [True, True]
        with pytest.raises(AssertionError):
            pk.prove_batch(host, [inst, inst], [r[:-64] for r in rbs])   # the binding refuses a short stream itself
        L = bzh2.load()
        import ctypes
        lens = (ctypes.c_size_t * 2)()
        out = np.zeros((2, pk.max_proof_bytes), dtype=np.uint8)
        short = b"".join(r[:-64] for r in rbs)
        rc = L.bzh_prove_batch(ctx.handle, pk.handle, 2, ctypes.c_void_p(host.ctypes.data), bzh2.FORM_CANONICAL, bzh2.MEM_HOST, None, 0,
                               short, len(rbs[0]) - 64, ctypes.c_void_p(out.ctypes.data), pk.max_proof_bytes, lens)
        assert rc == bzh2.E_ARG
        bad = host.copy()
        lookup_rows = [r for r in range(cs.usable_rows) if fixed[3][r]]
        bad[1, 0, lookup_rows[0]] = C.ints_to_array([12345])[0]         # advice 0 feeds the lookup where q_lookup = 1
        with pytest.raises(bzh2.BzhError) as ei:
            pk.prove_batch(bad, [inst, inst], rbs)
        assert ei.value.status == bzh2.E_RANGE
    finally:
        pk.close()
        ctx.close()


def test_native_prover_minimal_circuit_without_instance_permutation_or_lookup(gpu_ctx, oracle_c):
    """Degenerate shapes: no instance column, no permutation argument, no lookup -- one multiplication gate with a rotated
    query.  Prover bytes equal the oracle's, both verifiers accept, a broken row is rejected."""
    import bzh2
    from bzh2 import native as N, circuit_data as P
    cv, F = O.VESTA, O.FP
    p = F.p
    k, n = 4, 16
    A = lambda c, r=0: ('advice', c, r)
    gates = [('mul', ('fixed', 0, 0), ('add', ('mul', A(0), A(1)), ('neg', A(0, 1))))]     # q * (a0 * a1 - a0[next]) = 0
    cs = H.ConstraintSystem(k, 2, 1, 0, gates, [], [])
    usable = cs.usable_rows
    rr = random.Random(17)
    a0, a1 = [rr.randrange(p)], []
    for r in range(usable - 1):
        a1.append(rr.randrange(p))
        a0.append(a0[-1] * a1[-1] % p)
    a1.append(0)
    fixed = [[1] * (usable - 1) + [0] * (n - usable + 1)]
    adv = [a0 + [0] * (n - usable), a1 + [0] * (n - usable)]
    rng, g, w, u = _setup(cs, 7001)
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, fixed, [])
    circ = P.Circuit(k, 2, 1, 0, gates, [], [], fixed, [])
    pk = N.NativeProvingKey(gpu_ctx, circ, bzh2.CURVE_VESTA, g, w, u)
    try:
        rbytes = bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes))
        rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(pk.rng_bytes // 64)]
        want = H.create_proof(keys, adv, [], rs, O.Blake2bTranscript(F))
        assert H.verify_proof(keys, [], want, O.Blake2bTranscript(F))
        got = pk.prove_batch(np.stack([_adv_array(adv, n)]), [[]], [rbytes])
        assert got == [want]
        assert pk.verify_batch([[]], got) == [True]
        broken = [list(adv[0]), list(adv[1])]
        broken[0][3] = (broken[0][3] + 1) % p
        try:
            bad = pk.prove_batch(np.stack([_adv_array(broken, n)]), [[]], [rbytes])
        except bzh2.BzhError as e:
            assert e.status == bzh2.E_RANGE
        else:
            assert pk.verify_batch([[]], bad) == [False]
    finally:
        pk.close()


@pytest.mark.parametrize("k", [11])     # (k = 14 / 17 of the REAL BoardCircuit: tests/test_gpu_real_circuit_parity.py)
def test_reference_size_native_proofs_pass_the_oracle_verifier(gpu_ctx, oracle_c, monkeypatch, k):
    """Reference sizes (k = 11: benches/shot.rs:22; k = 14: the Board scale-up bench.py measures): proofs of the BattleZips-shaped circuit made by bzh_prove_batch are
    checked by the ORACLE's verify_proof (its n-term MSMs delegated to the C oracle, or the big-int sums would take hours);
    a proof with one flipped byte is rejected by both verifiers."""
    import bzh2
    from bzh2 import native as N
    from helpers import synth
    cv, F = O.VESTA, O.FP
    circ, adv, inst = synth.battlezips_shaped(k, seed=123)
    cs = H.ConstraintSystem(circ.k, 11, 8, 1, circ.gates, circ.perm_columns, circ.lookups, degree=9)
    rng = random.Random(1111)
    g0 = cv.random_point(rng)
    walk = C.point_walk(0, C.points_to_array([g0])[0], cs.n + 2)                   # G_i = [i+1] G_0 via the C oracle
    pts = [C.array_to_point(walk[i]) for i in range(cs.n + 2)]
    g, u, w = pts[:cs.n], pts[cs.n], pts[cs.n + 1]
    fast = lambda self, scalars, points: C.array_to_point(C.msm(0, C.ints_to_array([int(s) % F.p for s in scalars]),
                                                                   C.points_to_array(points), 8))
    monkeypatch.setattr(type(cv), "msm_naive", fast)
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, circ.fixed, circ.copies)
    pk = N.NativeProvingKey(gpu_ctx, circ, bzh2.CURVE_VESTA, g, w, u)
    try:
        rbs = [bytes(rng.getrandbits(8) for _ in range(pk.rng_bytes)) for _ in range(2)]
        advs = np.stack([_adv_array(adv, cs.n)] * 2)
        proofs = pk.prove_batch(advs, [inst, inst], rbs)
        assert proofs[0] != proofs[1]
        assert H.verify_proof(keys, inst, proofs[0], O.Blake2bTranscript(F))
        assert H.verify_proof(keys, inst, proofs[1], O.Blake2bTranscript(F))
        bad = proofs[0][:700] + bytes([proofs[0][700] ^ 1]) + proofs[0][701:]
        assert not H.verify_proof(keys, inst, bad, O.Blake2bTranscript(F))
        assert pk.verify_batch([inst, inst, inst], [proofs[0], proofs[1], bad]) == [True, True, False]
    finally:
        pk.close()
