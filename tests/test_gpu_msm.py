"""GPU parity: bzh_msm (HIP, through the C ABI) vs the CPU oracle, bit-exact
on canonical compressed points.  Reference seam: halo2_proofs best_multiexp as
reached from create_proof (benches/shot.rs:68, src/circuits/board.rs:913-920)."""
import random

import numpy as np
import pytest

import coracle as C
import pasta as O
from randutil import SCALAR_MODULUS, uniform_below

pytestmark = pytest.mark.gpu


def rand_scalars(rng, n, cid=0):
    """uniform below the scalar-field modulus of curve `cid` (tests/randutil.py)"""
    return uniform_below(rng, n, SCALAR_MODULUS[cid])


def walk_bases(cid, n, seed):
    cv = O.CURVE_BY_ID[cid]
    g = cv.random_point(random.Random(seed))
    return C.point_walk(cid, C.points_to_array([g])[0], n)


def gpu_msm_compressed(bzh2, ctx, cid, bases_np, scalars_np, form=0, precompute=None):
    b = ctx.upload_bases(cid, bases_np)
    if precompute is not None:
        b.precompute(precompute)
    try:
        jac = ctx.msm(b, scalars_np, form=form)
    finally:
        b.free()
    aff = bzh2.jacobian_to_affine(cid, jac, form)
    return bzh2.affine_compress(cid, aff, form)


def oracle_compressed(cid, bases_np, scalars_np, threads=8):
    cv = O.CURVE_BY_ID[cid]
    out = []
    s = scalars_np if scalars_np.ndim == 3 else scalars_np.reshape(1, *scalars_np.shape)
    for v in s:
        out.append(cv.compress(C.array_to_point(C.msm(cid, np.ascontiguousarray(v), bases_np[: v.shape[0]], threads))))
    return out


@pytest.mark.parametrize("cid", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 2, 3, 31, 64, 257, 2048])
def test_msm_matches_oracle(gpu_ctx, oracle_c, cid, n):
    import bzh2
    rng = np.random.default_rng(1000 * cid + n)
    bases = walk_bases(cid, n, seed=cid)
    sc = rand_scalars(rng, n, cid)
    assert gpu_msm_compressed(bzh2, gpu_ctx, cid, bases, sc) == oracle_compressed(cid, bases, sc)


@pytest.mark.parametrize("cid", [0, 1])
def test_msm_edge_cases(gpu_ctx, oracle_c, cid):
    """zeros, ones, r-1, repeated points, P and -P, identity bases."""
    import bzh2
    cv = O.CURVE_BY_ID[cid]
    r = cv.scalar.p
    rng = random.Random(77 + cid)
    pts = [cv.random_point(rng) for _ in range(24)]
    pts += [pts[0], pts[0], cv.neg(pts[1]), None, pts[2]]
    sc = [rng.randrange(r) for _ in pts]
    sc[0], sc[24], sc[25] = 5, 5, r - 10          # same point thrice, scalars summing to 0 mod r
    sc[1], sc[26] = 9, 9                           # P and -P with equal scalars cancel
    sc[3], sc[4], sc[5], sc[6] = 0, 1, r - 1, 2
    sc[27] = 12345                                 # scalar on the identity base
    want = cv.compress(cv.msm_naive(sc, pts))
    got = gpu_msm_compressed(bzh2, gpu_ctx, cid, C.points_to_array(pts), C.ints_to_array(sc))
    assert got == [want]
    # all-zero scalars -> identity (32 zero bytes); all-equal scalars hit a single bucket per window
    z = np.zeros((len(pts), 4), dtype=np.uint64)
    assert gpu_msm_compressed(bzh2, gpu_ctx, cid, C.points_to_array(pts), z) == [bytes(32)]
    eq = C.ints_to_array([r - 1] * len(pts))
    assert gpu_msm_compressed(bzh2, gpu_ctx, cid, C.points_to_array(pts), eq) == \
        [cv.compress(cv.msm_naive([r - 1] * len(pts), pts))]


def test_msm_empty_and_prefix(gpu_ctx, oracle_c):
    import bzh2
    bases = walk_bases(0, 64, seed=3)
    b = gpu_ctx.upload_bases(0, bases)
    try:
        out = gpu_ctx.msm(b, np.zeros((0, 4), dtype=np.uint64))
        assert (out == 0).all()                     # empty sum = identity (Z = 0)
        rng = np.random.default_rng(4)
        sc = rand_scalars(rng, 40)            # shorter than the table: prefix, like Params::commit
        jac = gpu_ctx.msm(b, sc)
        got = bzh2.affine_compress(0, bzh2.jacobian_to_affine(0, jac))
        assert got == oracle_compressed(0, bases, sc)
        with pytest.raises(bzh2.BzhError):          # longer than the table: upstream asserts
            gpu_ctx.msm(b, rand_scalars(rng, 65))
    finally:
        b.free()


def test_msm_batched_shared_bases_and_montgomery(gpu_ctx, oracle_c):
    """28 scalar vectors against one table (the per-proof commit set), Montgomery in/out."""
    import bzh2
    n, batch = 2048, 28
    rng = np.random.default_rng(5)
    bases = walk_bases(0, n, seed=9)
    sc = rand_scalars(rng, n * batch).reshape(batch, n, 4)
    want = oracle_compressed(0, bases, sc)
    assert gpu_msm_compressed(bzh2, gpu_ctx, 0, bases, sc) == want
    R, p = O.FP.R, O.FP.p
    sc_m = C.ints_to_array([x * R % p for x in C.array_to_ints(sc.reshape(-1, 4))]).reshape(batch, n, 4)
    assert gpu_msm_compressed(bzh2, gpu_ctx, 0, bases, sc_m, form=bzh2.FORM_MONTGOMERY) == \
        gpu_msm_compressed(bzh2, gpu_ctx, 0, bases, sc)  # same points (different coordinate form in, same bytes out)


@pytest.mark.parametrize("log_n", [14, 16])
def test_msm_medium_matches_oracle(gpu_ctx, oracle_c, log_n):
    import bzh2
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    bases = walk_bases(0, n, seed=log_n)
    sc = rand_scalars(rng, n)
    assert gpu_msm_compressed(bzh2, gpu_ctx, 0, bases, sc) == oracle_compressed(0, bases, sc)


def test_msm_large_linearity(gpu_ctx, oracle_c):
    """2^18 points (multi-chunk path): MSM(a) + MSM(b) == MSM(a+b) and chunk-wise additivity;
    one half-size MSM is also checked against the oracle directly."""
    import bzh2
    n = 1 << 18
    cv = O.VESTA
    rng = np.random.default_rng(18)
    bases = walk_bases(0, n, seed=18)
    a = rand_scalars(rng, n)
    b = rand_scalars(rng, n)
    ai, bi = C.array_to_ints(a), C.array_to_ints(b)
    ab = C.ints_to_array([(x + y) % O.P for x, y in zip(ai, bi)])
    hb = gpu_ctx.upload_bases(0, bases)
    try:
        jac = gpu_ctx.msm(hb, np.stack([a, b, ab]))
    finally:
        hb.free()
    pa, pb, pab = [C.array_to_point(x) for x in bzh2.jacobian_to_affine(0, jac)]
    assert cv.add(pa, pb) == pab
    half = n // 2
    assert cv.compress(pa) != bytes(32)
    lo = gpu_msm_compressed(bzh2, gpu_ctx, 0, bases[:half], a[:half])
    assert lo == oracle_compressed(0, bases[:half], a[:half])


@pytest.mark.parametrize("cid", [0, 1, 2])
@pytest.mark.parametrize("n,c", [(1, 0), (3, 5), (257, 0), (2048, 0), (2048, 8), (5000, 11)])
def test_msm_precomputed_table_matches_oracle(gpu_ctx, oracle_c, cid, n, c):
    """bzh_bases_precompute (fixed-base window table) must not change a single byte."""
    import bzh2
    rng = np.random.default_rng(7000 * cid + n + c)
    bases = walk_bases(cid, n, seed=40 + cid)
    sc = rand_scalars(rng, n, cid)
    want = oracle_compressed(cid, bases, sc)
    assert gpu_msm_compressed(bzh2, gpu_ctx, cid, bases, sc, precompute=c) == want


def test_msm_precomputed_prefix_batch_and_edges(gpu_ctx, oracle_c):
    """Window table + (a) MSMs over a prefix of the table, (b) a batch, (c) skewed digits:
    all-equal scalars, r-1, 0, 1, and the identity as a base."""
    import bzh2
    cv = O.VESTA
    r = cv.scalar.p
    n = 3000
    rng = np.random.default_rng(99)
    bases = walk_bases(0, n, seed=77)
    bases[17] = 0                                   # identity base
    hb = gpu_ctx.upload_bases(0, bases).precompute()
    try:
        for m in (1, 2, 1000, 2999, 3000):          # prefixes (row_len != row_stride)
            sc = rand_scalars(rng, m)
            jac = gpu_ctx.msm(hb, sc)
            got = bzh2.affine_compress(0, bzh2.jacobian_to_affine(0, jac))
            assert got == oracle_compressed(0, bases, sc), m
        batch = np.stack([rand_scalars(rng, n) for _ in range(5)])
        batch[1] = C.ints_to_array([r - 1] * n)     # every digit identical: one bucket per window
        batch[2] = 0
        batch[3] = C.ints_to_array([1] * n)
        jac = gpu_ctx.msm(hb, batch)
        got = bzh2.affine_compress(0, bzh2.jacobian_to_affine(0, jac))
        assert got == oracle_compressed(0, bases, batch)
        assert got[2] == bytes(32)
    finally:
        hb.free()


def test_msm_precomputed_2_16_matches_oracle(gpu_ctx, oracle_c):
    import bzh2
    n = 1 << 16
    rng = np.random.default_rng(16)
    bases = walk_bases(0, n, seed=16)
    sc = rand_scalars(rng, n)
    assert gpu_msm_compressed(bzh2, gpu_ctx, 0, bases, sc, precompute=0) == oracle_compressed(0, bases, sc)


@pytest.mark.parametrize("nvec", [24, 13])
def test_msm_precomputed_many_vectors_chunk_presum(gpu_ctx, oracle_c, nvec):
    """Throughput regime of the window-table MSM: 24 (13) vectors of 2^14 + 2 scalars (an advice-commitment launch of a
    lockstep proof batch) cut into >= 256 chunk segments, whose bucket sets are summed per vector before the
    running-sum reduction (k_msm_chunksum); the accumulate workgroups run in the XCD-aware order (8 runs of (chunk, vector)
    pairs; 13 vectors leave the runs ragged).  Skewed vectors included; every result against the C oracle."""
    import bzh2
    rng = np.random.default_rng(141 + nvec)
    n = (1 << 14) + 2
    bases = walk_bases(0, n, seed=5)
    hb = gpu_ctx.upload_bases(0, bases).precompute(11)
    try:
        r = O.VESTA.scalar.p
        batch = np.stack([rand_scalars(rng, n) for _ in range(nvec)])
        batch[3] = 0
        batch[5] = C.ints_to_array([r - 1] * n)          # every digit identical
        batch[7, : n // 2] = 0                           # half-empty vector: empty chunks
        batch[9] = C.ints_to_array([(i % 2) for i in range(n)])
        jac = gpu_ctx.msm(hb, batch)
        got = bzh2.affine_compress(0, bzh2.jacobian_to_affine(0, jac))
        assert got == oracle_compressed(0, bases, batch)
    finally:
        hb.free()


@pytest.mark.parametrize("cid", [0, 1, 2])
def test_bases_walk_is_the_oracles_point_walk(gpu_ctx, oracle_c, cid):
    """bzh_bases_walk (the device-made base set of the 2^24-point microbench, BASELINE.json configs[4]): bases[i] = [i + 1] G,
    against the oracle's orc_point_walk -- sizes that end inside a thread's run of 16, on it, and across many workgroups --
    read back through bzh_bases_points; and an MSM over the walked table equals the oracle's over its own walk."""
    import random
    import numpy as np
    import pasta as O
    cv = O.CURVE_BY_ID[cid]
    g = oracle_c.points_to_array([cv.random_point(random.Random(90 + cid))])[0]
    for n in (1, 15, 16, 17, 1000, 70001):
        want = oracle_c.point_walk(cid, g, n)
        b = gpu_ctx.bases_walk(cid, g, n)
        try:
            got = gpu_ctx.bases_points(b, 0, n)
            assert (got == want).all(), (cid, n, int(np.argmax((got != want).any(axis=1))))
            if n >= 1000:
                assert (gpu_ctx.bases_points(b, n - 7, 7) == want[n - 7:]).all()
                rng = np.random.default_rng(n)
                sc = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
                sc[:, 3] &= (1 << 60) - 1
                import bzh2
                got_pt = oracle_c.array_to_point(bzh2.jacobian_to_affine(cid, gpu_ctx.msm(b, sc))[0])
                assert got_pt == oracle_c.array_to_point(oracle_c.msm(cid, sc, want, 8))
                with pytest.raises(bzh2.BzhError):
                    gpu_ctx.bases_points(b, n - 1, 2)
        finally:
            b.free()
