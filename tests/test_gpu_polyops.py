"""GPU parity of the prover-stage vector primitives (N4/N5/N6) against the
big-int oracle, bit-exact.  Reference seam: the helpers halo2_proofs::plonk::create_proof
(benches/shot.rs:68) runs between its MSMs and FFTs."""
import random

import numpy as np
import pytest

import coracle as C
import pasta as O

pytestmark = pytest.mark.gpu


def rand_ints(rng, n, p, zeros=()):
    v = [rng.randrange(p) for _ in range(n)]
    for z in zeros:
        if z < n:
            v[z] = 0
    return v


@pytest.mark.parametrize("fid", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 16384])
def test_batch_invert(gpu_ctx, fid, n):
    F = O.FIELD_BY_ID[fid]
    rng = random.Random(10 * fid + n)
    v = rand_ints(rng, n, F.p, zeros=(0, 5, n - 1))
    got = C.array_to_ints(gpu_ctx.batch_invert(fid, C.ints_to_array(v)))
    assert got == O.batch_invert(v, F)


@pytest.mark.parametrize("n", [1, 7, 2047, 2048, 2049, 5000, 16384])
def test_prefix_product(gpu_ctx, n):
    F = O.FP
    rng = random.Random(n)
    v = rand_ints(rng, n, F.p)
    got = C.array_to_ints(gpu_ctx.prefix_product(0, C.ints_to_array(v)))
    assert got == O.prefix_product(v, F)


def test_prefix_product_batch_and_zero(gpu_ctx):
    F = O.FQ
    rng = random.Random(3)
    vs = [rand_ints(rng, 4100, F.p) for _ in range(3)]
    vs[1][2050] = 0                                   # everything after a zero factor is zero
    arr = np.stack([C.ints_to_array(v) for v in vs])
    got = gpu_ctx.prefix_product(1, arr)
    for b in range(3):
        assert C.array_to_ints(got[b]) == O.prefix_product(vs[b], F)


@pytest.mark.parametrize("n", [1, 3, 255, 256, 257, 2048, 16384])
def test_eval_polynomial(gpu_ctx, n):
    F = O.FP
    rng = random.Random(100 + n)
    polys = [rand_ints(rng, n, F.p) for _ in range(4)]
    xs = [rng.randrange(F.p), 0, 1, F.p - 1]
    arr = np.stack([C.ints_to_array(p) for p in polys])
    got = C.array_to_ints(gpu_ctx.eval_polynomial(0, arr, C.ints_to_array(xs)))
    assert got == [O.eval_polynomial(p, x, F) for p, x in zip(polys, xs)]
    same = C.array_to_ints(gpu_ctx.eval_polynomial(0, arr, C.ints_to_array(xs[:1])))   # one point for all
    assert same == [O.eval_polynomial(p, xs[0], F) for p in polys]


def test_inner_product_fold_vecmul(gpu_ctx):
    F = O.FP
    rng = random.Random(9)
    n = 4096
    a, b = rand_ints(rng, n, F.p), rand_ints(rng, n, F.p)
    A, B = C.ints_to_array(a), C.ints_to_array(b)
    assert C.array_to_ints(gpu_ctx.inner_product(0, A, B)) == [O.inner_product(a, b, F)]
    u = rng.randrange(F.p)
    assert C.array_to_ints(gpu_ctx.fold(0, A, C.ints_to_array([u]))[0]) == O.fold_scalars(a, u, F)
    assert C.array_to_ints(gpu_ctx.vec_mul(0, A, B)) == [x * y % F.p for x, y in zip(a, b)]


def test_grand_product_pipeline(gpu_ctx):
    """The permutation-argument shape: z[i+1] = z[i] * num[i] / den[i] via batch_invert + vec_mul +
    prefix_product, checked against the big-int definition."""
    F = O.FP
    rng = random.Random(12)
    n = 4096
    num, den = rand_ints(rng, n, F.p), [rng.randrange(1, F.p) for _ in range(n)]
    inv = gpu_ctx.batch_invert(0, C.ints_to_array(den))
    ratio = gpu_ctx.vec_mul(0, C.ints_to_array(num), inv)
    z = C.array_to_ints(gpu_ctx.prefix_product(0, ratio))
    want, acc = [], 1
    for x, y in zip(num, den):
        want.append(acc)
        acc = acc * x % F.p * F.inv(y) % F.p
    assert z == want


def test_montgomery_form_roundtrip(gpu_ctx):
    import bzh2
    F = O.FP
    rng = random.Random(77)
    v = rand_ints(rng, 300, F.p)
    vm = C.ints_to_array([x * F.R % F.p for x in v])
    got = C.array_to_ints(gpu_ctx.batch_invert(0, vm, form=bzh2.FORM_MONTGOMERY))
    assert got == [F.inv(x) * F.R % F.p for x in v]


@pytest.mark.parametrize("n", [1, 2, 3, 257, 2048, 5000, 16384])
def test_kate_division(gpu_ctx, n):
    F = O.FP
    rng = random.Random(500 + n)
    c = rand_ints(rng, n, F.p)
    for x in (rng.randrange(F.p), 0, 1):
        got = C.array_to_ints(gpu_ctx.kate_division(0, C.ints_to_array(c), x))
        assert got == O.kate_division(c, x, F), (n, x)


@pytest.mark.parametrize("field,n,batch", [(0, 1000, 5), (1, 4097, 3), (0, 131072, 2), (2, 777, 4), (0, 9, 70)])
def test_kate_division_batch(gpu_ctx, field, n, batch):
    """arithmetic::kate_division over a batch, one point per polynomial (the multiopen's use: one workgroup per polynomial,
    segments of the Horner recurrence stitched through LDS): ragged lengths, every field, x = 0 and x = 1 among the points;
    big n checked through the identity p(X) = q(X) (X - x) + p(x) at a random point instead of the big-int division."""
    import numpy as np
    F = [O.FP, O.FQ, O.BN_FR][field]
    rng = random.Random(900 + n + field)
    cs = [rand_ints(rng, n, F.p) for _ in range(batch)]
    xs = [rng.randrange(F.p) for _ in range(batch)]
    xs[0] = 0
    if batch > 1:
        xs[1] = 1
    got = gpu_ctx.kate_division_batch(field, np.stack([C.ints_to_array(c) for c in cs]), xs)
    for v in range(batch):
        q = C.array_to_ints(got[v])
        if n <= 5000:
            assert q == O.kate_division(cs[v], xs[v], F), (v, xs[v])
        else:
            z = rng.randrange(F.p)
            ev = lambda poly, at: C.eval_poly(field, C.ints_to_array(poly), at)
            pz, qz, px = ev(cs[v], z), ev(q, z), ev(cs[v], xs[v])
            assert (qz * (z - xs[v]) + px - pz) % F.p == 0, v
            assert q[-1] == cs[v][-1]
