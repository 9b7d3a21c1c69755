"""GPU parity of the inner-product-argument opening.

The GPU prover never folds generators (MSMs against the fixed SRS table with re-derived scalars);
the oracle restates upstream's collapsing algorithm.  Under the same RNG byte stream both must emit
the SAME proof bytes, and both verifiers (big-int oracle, bzh_ipa_verify) must accept them.
Reference call chain: create_proof (benches/shot.rs:68) -> multiopen -> commitment::create_proof;
verify_proof (benches/board.rs:80-86)."""
import random

import numpy as np
import pytest

import coracle as C
import pasta as O
from helpers.real_parity import accelerated_oracle

pytestmark = pytest.mark.gpu


def setup_case(cid, k, seed):
    cv = O.CURVE_BY_ID[cid]
    F = cv.scalar
    rng = random.Random(seed)
    n = 1 << k
    g = [cv.random_point(rng) for _ in range(n)]
    w, u = cv.random_point(rng), cv.random_point(rng)
    poly = [rng.randrange(F.p) for _ in range(n)]
    blind, x3 = rng.randrange(F.p), rng.randrange(F.p)
    rbytes = bytes(rng.getrandbits(8) for _ in range(64 * (n + 1 + 2 * k)))
    rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(n + 1 + 2 * k)]
    return cv, F, g, w, u, poly, blind, x3, rbytes, rs


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("k", [1, 2, 5])
@pytest.mark.parametrize("precompute", [False, True])
def test_ipa_open_bytes_match_oracle_and_verify(gpu_ctx, cid, k, precompute):
    import bzh2
    cv, F, g, w, u, poly, blind, x3, rbytes, rs = setup_case(cid, k, 100 * cid + k)
    fid = bzh2.CURVE_SCALAR_FIELD[cid]
    # oracle: upstream's algorithm with generator collapse
    ot = O.Blake2bTranscript(F)
    ot.common_scalar(12345)                       # some prior transcript state
    v_want = O.ipa_open(cv, g, w, u, poly, blind, x3, rs, ot)
    # GPU
    hb = gpu_ctx.upload_bases(cid, C.points_to_array(g + [u, w]))
    if precompute:
        hb.precompute()
    try:
        t = bzh2.Transcript(fid)
        t.common_scalar(12345)
        v = gpu_ctx.ipa_open(hb, C.ints_to_array(poly), blind, x3, rbytes, t)
        proof = t.proof()
        assert v == v_want == O.eval_polynomial(poly, x3, F)
        assert proof == bytes(ot.proof)
        # verification, three ways
        P = cv.add(cv.msm_naive(poly, g), cv.mul(blind, w))
        vt = O.Blake2bTranscript(F)
        vt.common_scalar(12345)
        assert O.ipa_verify(cv, g, w, u, P, x3, v, proof, vt)
        t2 = bzh2.Transcript(fid)
        t2.common_scalar(12345)
        assert gpu_ctx.ipa_verify(hb, P, x3, v, proof, t2, [g[0], u, w])
        # tampering is rejected: wrong evaluation, flipped proof byte, wrong commitment
        for bad_v, bad_proof, bad_P in (((v + 1) % F.p, proof, P),
                                        (v, proof[:40] + bytes([proof[40] ^ 1]) + proof[41:], P),
                                        (v, proof, cv.add(P, g[0]))):
            t3 = bzh2.Transcript(fid)
            t3.common_scalar(12345)
            assert not gpu_ctx.ipa_verify(hb, bad_P, x3, bad_v, bad_proof, t3, [g[0], u, w])
    finally:
        hb.free()


@pytest.mark.parametrize("k,batch", [(3, 4), (6, 3)])
@accelerated_oracle
def test_ipa_open_batch_matches_per_proof_oracle(gpu_ctx, k, batch):
    """bzh_ipa_open_batch: independent openings in lockstep, each with its own polynomial, blind, point, randomness
    and transcript state, must emit exactly the bytes the oracle emits for that opening alone."""
    import bzh2
    cid = 0
    cases = [setup_case(cid, k, 900 + k) for _ in range(1)]
    cv, F, g, w, u = cases[0][:5]
    rng = random.Random(77 + k)
    n = 1 << k
    hb = gpu_ctx.upload_bases(cid, C.points_to_array(g + [u, w])).precompute()
    try:
        polys, blinds, x3s, rbs, want_v, want_proof, trs = [], [], [], [], [], [], []
        for b in range(batch):
            poly = [rng.randrange(F.p) for _ in range(n)]
            if b == 1:
                poly = [0] * n                      # an all-zero polynomial: every L/R scalar vector is sparse
            blind, x3 = rng.randrange(F.p), rng.randrange(F.p)
            rbytes = bytes(rng.getrandbits(8) for _ in range(64 * (n + 1 + 2 * k)))
            rs = [O.from_u512(rbytes[64 * i:64 * (i + 1)], F) for i in range(n + 1 + 2 * k)]
            ot = O.Blake2bTranscript(F)
            ot.common_scalar(1000 + b)
            want_v.append(O.ipa_open(cv, g, w, u, poly, blind, x3, rs, ot))
            want_proof.append(bytes(ot.proof))
            t = bzh2.Transcript(bzh2.CURVE_SCALAR_FIELD[cid])
            t.common_scalar(1000 + b)
            trs.append(t)
            polys.append(C.ints_to_array(poly))
            blinds.append(blind)
            x3s.append(x3)
            rbs.append(rbytes)
        got_v = gpu_ctx.ipa_open_batch(hb, np.stack(polys), blinds, x3s, rbs, trs)
        assert got_v == want_v
        for t, wp in zip(trs, want_proof):
            assert t.proof() == wp
    finally:
        hb.free()


def test_ipa_k11_roundtrip_shot_size(gpu_ctx, oracle_c):
    """Shot-circuit size (k = 11, benches/shot.rs:22): prove on the GPU, verify on the GPU; the commitment
    is checked against the C oracle's MSM."""
    import bzh2
    k, n = 11, 1 << 11
    cv, F = O.VESTA, O.FP
    rng = random.Random(11)
    gen = cv.random_point(rng)
    table = C.point_walk(0, C.points_to_array([gen])[0], n + 2)
    poly = np.frombuffer(np.random.default_rng(11).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    poly[:, 3] &= (1 << 61) - 1
    blind, x3 = rng.randrange(F.p), rng.randrange(F.p)
    rbytes = np.random.default_rng(12).bytes(64 * (n + 1 + 2 * k))
    hb = gpu_ctx.upload_bases(0, table).precompute()
    try:
        sc = np.concatenate([poly, C.ints_to_array([0, blind])])
        P = C.array_to_point(C.msm(0, sc, table, 8))
        t = bzh2.Transcript(bzh2.FIELD_FP)
        v = gpu_ctx.ipa_open(hb, poly, blind, x3, rbytes, t)
        assert v == C.eval_poly(0, poly, x3)
        proof = t.proof()
        assert len(proof) == 32 * (1 + 2 * k + 2)
        pts = [C.array_to_point(table[i]) for i in (0, n, n + 1)]
        assert gpu_ctx.ipa_verify(hb, P, x3, v, proof, bzh2.Transcript(bzh2.FIELD_FP), pts)
        assert not gpu_ctx.ipa_verify(hb, P, (x3 + 1) % F.p, v, proof, bzh2.Transcript(bzh2.FIELD_FP), pts)
    finally:
        hb.free()
