"""BASELINE.json configs at their OWN sizes, through the C ABI, against the C oracle (oracle/oracle.c):
  configs[4]  2^24-point MSM on Vesta, Pallas and BN254 G1 (plain bases and fixed-base window table) == best_multiexp restatement
  configs[4]  2^22 NTT over Fp, Fq and BN254 Fr: forward, ZETA-coset and inverse, the full vector bit for bit
(The metric's k = 14 / 17 proofs of the REAL BoardCircuit -- native and oracle verifier, golden bytes at k = 17 -- are in
tests/test_gpu_real_circuit_parity.py; the synthetic k = 17 look-alike that stood here in round 2 is gone.)
Reference seams: halo2_proofs `best_multiexp` / `best_fft` as reached from create_proof (benches/shot.rs:68,
benches/board.rs:61-68).  Scalars / elements are uniform below the modulus (tests/randutil.py)."""
import os
import random

import numpy as np
import pytest

import coracle as C
import pasta as O
from randutil import FIELD_MODULUS, SCALAR_MODULUS, uniform_below

pytestmark = pytest.mark.gpu

THREADS = max(1, min(32, os.cpu_count() or 1))


@pytest.mark.parametrize("cid", [0, 1, 2], ids=["vesta", "pallas", "bn254_g1"])
def test_msm_2_24_matches_oracle_plain_and_window_table(gpu_ctx, oracle_c, cid):
    """configs[4] names Pallas + BN256; SURVEY section 8d lists Vesta/Fp (the reference's commitment curve), Pallas/Fq and
    BN254 G1/Fr (no reference, SURVEY F3: checked against the C oracle's own best_multiexp restatement)."""
    import bzh2
    n = 1 << 24
    cv = O.CURVE_BY_ID[cid]
    g = cv.random_point(random.Random(2024 + cid))
    bases = C.point_walk(cid, C.points_to_array([g])[0], n)         # G_i = [i+1]G, affine, from the C oracle
    rng = np.random.default_rng(24 + cid)
    sc = uniform_below(rng, n, SCALAR_MODULUS[cid])
    sc[7] = 0
    sc[8] = C.int_to_limbs(SCALAR_MODULUS[cid] - 1)
    want = cv.compress(C.array_to_point(C.msm(cid, sc, bases, THREADS)))
    assert want != bytes(32)
    hb = gpu_ctx.upload_bases(cid, bases)
    try:
        jac = gpu_ctx.msm(hb, sc)
        got = bzh2.affine_compress(cid, bzh2.jacobian_to_affine(cid, jac))
        assert got == [want]
        hb.precompute(13)                                            # 20 table rows of 2^24 points: 20 GiB of HBM
        jac = gpu_ctx.msm(hb, sc)
        got = bzh2.affine_compress(cid, bzh2.jacobian_to_affine(cid, jac))
        assert got == [want]
    finally:
        hb.free()


@pytest.mark.parametrize("fid", [0, 1, 2], ids=["fp", "fq", "bn254_fr"])
def test_ntt_2_22_full_vector_matches_oracle(gpu_ctx, oracle_c, fid):
    F = O.FIELD_BY_ID[fid]
    k = 22
    n = 1 << k
    rng = np.random.default_rng(2222 + fid)
    a = uniform_below(rng, n, FIELD_MODULUS[fid])
    w = F.omega(k)
    zeta = pow(F.g, (F.p - 1) // 3, F.p)
    fwd = gpu_ctx.ntt(fid, a, omega=w)
    assert (fwd == C.ntt(fid, a, w, threads=THREADS)).all()
    cos = gpu_ctx.ntt(fid, a, omega=w, coset_shift=zeta)
    assert (cos == C.ntt(fid, a, w, coset_shift=zeta, threads=THREADS)).all()
    inv = gpu_ctx.ntt(fid, a, omega=w, inverse=True)
    assert (inv == C.ntt(fid, a, w, inverse=True, threads=THREADS)).all()
    cinv = gpu_ctx.ntt(fid, a, omega=w, inverse=True, coset_shift=F.g)
    assert (cinv == C.ntt(fid, a, w, inverse=True, coset_shift=F.g, threads=THREADS)).all()
