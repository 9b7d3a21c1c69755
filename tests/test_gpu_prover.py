"""Row a1: the staged Python drivers over the leaf C ABI (tests/helpers/prover*.py: test helpers, not the product prover) against the big-int oracle prover
(oracle/halo2_oracle.py): SAME proof bytes under a shared RNG byte stream, and the oracle verifier
accepts them (and rejects a wrong instance).  Call sites of the reference: create_proof
benches/shot.rs:68, src/circuits/board.rs:913-920; verify_proof benches/board.rs:80-86."""
import random

import pytest

import halo2_oracle as H
import pasta as O
import sample_circuit as S
from helpers.real_parity import accelerated_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,with_lookup,degree", [(4, False, None), (5, True, None), (5, True, 6)])
def test_create_proof_bytes_match_oracle(gpu_ctx, oracle_c, k, with_lookup, degree):
    import bzh2
    from helpers import prover as P
    cv, F = O.VESTA, O.FP
    cs, fixed, copies, adv, inst = S.build(k=k, seed=10 + k, with_lookup=with_lookup, degree=degree)
    rng = random.Random(1000 + k)
    g = [cv.random_point(rng) for _ in range(cs.n)]
    w, u = cv.random_point(rng), cv.random_point(rng)
    ndraws = 4000 + 3 * cs.n
    rbytes = bytes(rng.getrandbits(8) for _ in range(64 * ndraws))
    rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(ndraws)]
    # oracle
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, fixed, copies)
    want = H.create_proof(keys, adv, inst, rs, O.Blake2bTranscript(F))
    assert H.verify_proof(keys, inst, want, O.Blake2bTranscript(F))
    # GPU driver
    circ = P.Circuit(cs.k, cs.num_advice, cs.num_fixed, cs.num_instance, cs.gates, cs.perm_columns, cs.lookups, fixed, copies,
                     degree=degree)
    assert (circ.degree, circ.blinding_factors, circ.extended_k) == (cs.degree, cs.blinding_factors, cs.extended_k)
    pk = P.ProvingKey(gpu_ctx, circ, bzh2.CURVE_VESTA, g, w, u)
    t = bzh2.Transcript(bzh2.FIELD_FP)
    got = P.create_proof(pk, adv, inst, rbytes, t)
    assert got == want
    assert not H.verify_proof(keys, [[inst[0][0] + 1]], got, O.Blake2bTranscript(F))


def test_unsatisfied_witness_yields_a_rejected_proof(gpu_ctx, oracle_c):
    """create_proof does not check satisfiability (neither does upstream: benches/shot.rs proves an
    unsatisfiable witness); the proof it emits for a broken witness must not verify."""
    import bzh2
    from helpers import prover as P
    cv, F = O.VESTA, O.FP
    cs, fixed, copies, adv, inst = S.build(k=5, seed=3, with_lookup=False)
    adv[2][0] = (adv[2][0] + 1) % F.p                   # break the running-sum gate
    rng = random.Random(5)
    g = [cv.random_point(rng) for _ in range(cs.n)]
    w, u = cv.random_point(rng), cv.random_point(rng)
    circ = P.Circuit(cs.k, cs.num_advice, cs.num_fixed, cs.num_instance, cs.gates, cs.perm_columns, cs.lookups, fixed, copies)
    pk = P.ProvingKey(gpu_ctx, circ, bzh2.CURVE_VESTA, g, w, u)
    rbytes = bytes(rng.getrandbits(8) for _ in range(64 * 3000))
    try:
        proof = P.create_proof(pk, adv, inst, rbytes, bzh2.Transcript(bzh2.FIELD_FP))
    except ValueError:
        return                                           # surplus quotient coefficients exposed it already
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, fixed, copies)
    assert not H.verify_proof(keys, inst, proof, O.Blake2bTranscript(F))


@pytest.fixture(scope="module")
def stream_ctx(gpu_ctx):
    """A bzh2 context on torch's current stream (device-resident pipeline: torch owns the allocations)."""
    import torch
    import bzh2
    ctx = bzh2.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    yield ctx
    torch.cuda.synchronize()
    ctx.close()


@pytest.mark.parametrize("k,with_lookup,degree", [(4, False, None), (5, True, 6)])
def test_device_resident_prover_bytes_match_oracle(stream_ctx, oracle_c, k, with_lookup, degree):
    import torch
    import bzh2
    from helpers import prover as P, prover_dev as D
    cv, F = O.VESTA, O.FP
    cs, fixed, copies, adv, inst = S.build(k=k, seed=20 + k, with_lookup=with_lookup, degree=degree)
    rng = random.Random(2000 + k)
    g = [cv.random_point(rng) for _ in range(cs.n)]
    w, u = cv.random_point(rng), cv.random_point(rng)
    ndraws = 4000 + 3 * cs.n
    rbytes = bytes(rng.getrandbits(8) for _ in range(64 * ndraws))
    rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(ndraws)]
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, fixed, copies)
    want = H.create_proof(keys, adv, inst, rs, O.Blake2bTranscript(F))
    circ = P.Circuit(cs.k, cs.num_advice, cs.num_fixed, cs.num_instance, cs.gates, cs.perm_columns, cs.lookups, fixed, copies,
                     degree=degree)
    pk = D.DeviceProvingKey(stream_ctx, circ, bzh2.CURVE_VESTA, g, w, u, torch.device("cuda", 0))
    got = D.create_proof(pk, adv, inst, rbytes, bzh2.Transcript(bzh2.FIELD_FP))
    assert got == want
    assert H.verify_proof(keys, inst, got, O.Blake2bTranscript(F))


@accelerated_oracle
def test_battlezips_shaped_circuit_proof_matches_oracle(stream_ctx, oracle_c):
    """The benchmark circuit (tests/helpers/synth.py: 11 advice / 8 fixed / 13 permutation columns / degree 9 / one lookup /
    24 gates) at k = 7: device-resident proof == oracle proof, and the oracle verifier accepts it."""
    import torch
    import bzh2
    from helpers import prover_dev as D, synth
    cv, F = O.VESTA, O.FP
    circ, adv, inst = synth.battlezips_shaped(7, seed=5)
    cs = H.ConstraintSystem(circ.k, 11, 8, 1, circ.gates, circ.perm_columns, circ.lookups, degree=9)
    rng = random.Random(77)
    g = [cv.random_point(rng) for _ in range(cs.n)]
    w, u = cv.random_point(rng), cv.random_point(rng)
    ndraws = 6000
    rbytes = bytes(rng.getrandbits(8) for _ in range(64 * ndraws))
    rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(ndraws)]
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, circ.fixed, circ.copies)
    want = H.create_proof(keys, adv, inst, rs, O.Blake2bTranscript(F))
    assert H.verify_proof(keys, inst, want, O.Blake2bTranscript(F))
    pk = D.DeviceProvingKey(stream_ctx, circ, bzh2.CURVE_VESTA, g, w, u, torch.device("cuda", 0))
    adv_dev = [pk.ops.upload(col) for col in adv]          # witness resident in HBM, as in bench.py
    got = D.create_proof(pk, adv_dev, inst, rbytes, bzh2.Transcript(bzh2.FIELD_FP))
    assert got == want


@pytest.mark.parametrize("k,with_lookup,degree,batch", [(4, False, None, 2), (5, True, 6, 2)])
def test_lockstep_batch_prover_each_proof_matches_oracle(stream_ctx, oracle_c, k, with_lookup, degree, batch):
    """tests/helpers/prover_batch.create_proofs: `batch` different witnesses of one circuit proven in lockstep (one launch per
    kernel class per phase for all of them); proof b must be byte-identical to the oracle's proof of witness b under
    proof b's own randomness stream."""
    import torch
    import bzh2
    from helpers import prover as P, prover_batch as PB, prover_dev as D
    cv, F = O.VESTA, O.FP
    cases = [S.build(k=k, seed=300 + 7 * b + k, with_lookup=with_lookup, degree=degree) for b in range(batch)]
    cs, fixed, copies = cases[0][:3]
    assert all(cse[1] == fixed and cse[2] == copies for cse in cases)        # one circuit, different witnesses
    rng = random.Random(3000 + k)
    g = [cv.random_point(rng) for _ in range(cs.n)]
    w, u = cv.random_point(rng), cv.random_point(rng)
    ndraws = 4000 + 3 * cs.n
    keys = H.Keys(cs, H.Domain(cs, F), cv, g, w, u, fixed, copies)
    rbs, want = [], []
    for b in range(batch):
        rbytes = bytes(rng.getrandbits(8) for _ in range(64 * ndraws))
        rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(ndraws)]
        want.append(H.create_proof(keys, cases[b][3], cases[b][4], rs, O.Blake2bTranscript(F)))
        rbs.append(rbytes)
    circ = P.Circuit(cs.k, cs.num_advice, cs.num_fixed, cs.num_instance, cs.gates, cs.perm_columns, cs.lookups, fixed, copies,
                     degree=degree)
    pk = D.DeviceProvingKey(stream_ctx, circ, bzh2.CURVE_VESTA, g, w, u, torch.device("cuda", 0))
    bp = PB.BatchProver(pk)
    for _ in range(2):                                                        # second pass runs on the cached programs
        got = PB.create_proofs(bp, [cse[3] for cse in cases], [cse[4] for cse in cases], rbs,
                               [bzh2.Transcript(bzh2.FIELD_FP) for _ in range(batch)])
        assert got == want
    assert len(set(got)) == batch
    for b in range(batch):
        assert H.verify_proof(keys, cases[b][4], got[b], O.Blake2bTranscript(F))
