"""GPU parity: bzh_ntt (HIP, through the C ABI) vs the CPU oracle, bit-exact.
Reference seam: halo2_proofs best_fft / EvaluationDomain::{ifft, coeff_to_extended,
extended_to_coeff} as reached from create_proof (benches/shot.rs:68)."""
import numpy as np
import pytest

import coracle as C
import pasta as O
from randutil import FIELD_MODULUS, uniform_below

pytestmark = pytest.mark.gpu


def rand_elems(rng, n, fid=0):
    """uniform below the modulus of field `fid` (tests/randutil.py)"""
    return uniform_below(rng, n, FIELD_MODULUS[fid])


@pytest.mark.parametrize("fid", [0, 1, 2])
@pytest.mark.parametrize("k", [0, 1, 2, 5, 8, 11, 12, 13, 14, 17])
def test_ntt_matches_oracle(gpu_ctx, oracle_c, fid, k):
    F = O.FIELD_BY_ID[fid]
    rng = np.random.default_rng(100 * fid + k)
    a = rand_elems(rng, 1 << k, fid)
    w = F.omega(k)
    got = gpu_ctx.ntt(fid, a, omega=w)
    assert (got == C.ntt(fid, a, w, threads=8)).all()


@pytest.mark.parametrize("k", [3, 11, 14, 17])
def test_intt_and_coset_match_oracle(gpu_ctx, oracle_c, k):
    F = O.FP
    rng = np.random.default_rng(k)
    a = rand_elems(rng, 1 << k)
    w, zeta = F.omega(k), F.g
    assert (gpu_ctx.ntt(0, a, omega=w, inverse=True) == C.ntt(0, a, w, inverse=True, threads=8)).all()
    assert (gpu_ctx.ntt(0, a, omega=w, coset_shift=zeta) == C.ntt(0, a, w, coset_shift=zeta, threads=8)).all()
    back = gpu_ctx.ntt(0, gpu_ctx.ntt(0, a, omega=w, coset_shift=zeta), omega=w, inverse=True, coset_shift=zeta)
    assert (back == a).all()


def test_ntt_batched_and_montgomery(gpu_ctx, oracle_c):
    import bzh2
    F = O.FP
    k, batch = 11, 19
    rng = np.random.default_rng(7)
    a = rand_elems(rng, batch << k).reshape(batch, 1 << k, 4)
    w = F.omega(k)
    got = gpu_ctx.ntt(0, a, omega=w)
    for b in range(batch):
        assert (got[b] == C.ntt(0, a[b], w)).all()
    # Montgomery in / out: same transform on x*R
    R, p = F.R, F.p
    am = C.ints_to_array([x * R % p for x in C.array_to_ints(a[0])])
    gm = gpu_ctx.ntt(0, am, omega=w * R % p, form=bzh2.FORM_MONTGOMERY)
    assert C.array_to_ints(gm) == [x * R % p for x in C.array_to_ints(got[0])]


def test_ntt_default_omega_is_domain_generator(gpu_ctx, oracle_c):
    F = O.FQ
    rng = np.random.default_rng(8)
    a = rand_elems(rng, 1 << 9, 1)
    assert (gpu_ctx.ntt(1, a) == C.ntt(1, a, F.omega(9))).all()


def test_ntt_errors(gpu_ctx):
    import bzh2
    with pytest.raises(bzh2.BzhError):
        gpu_ctx.ntt(0, np.zeros((3, 4), dtype=np.uint64))       # not a power of two
    with pytest.raises(bzh2.BzhError):
        gpu_ctx.ntt(3, np.zeros((4, 4), dtype=np.uint64), omega=1)  # BN254 Fq has 2-adicity 1


def test_ntt_2_22_roundtrip_and_spot_values(gpu_ctx, oracle_c):
    """Config 5 size: round trip + linearity-free spot check of 3 outputs by the definition
    restricted to a sparse input (oracle finishes in seconds)."""
    F = O.FP
    k = 22
    n = 1 << k
    rng = np.random.default_rng(22)
    a = rand_elems(rng, n)
    w = F.omega(k)
    fwd = gpu_ctx.ntt(0, a, omega=w)
    assert (gpu_ctx.ntt(0, fwd, omega=w, inverse=True) == a).all()
    sparse = np.zeros((n, 4), dtype=np.uint64)
    idx = [0, 1, 2049, n // 2 + 3, n - 1]
    sparse[idx] = a[idx]
    out = gpu_ctx.ntt(0, sparse, omega=w)
    vals = [C.limbs_to_int(a[i]) for i in idx]
    for i in (0, 5, 123457, n - 1):
        want = sum(v * pow(w, i * j, F.p) for j, v in zip(idx, vals)) % F.p
        assert C.limbs_to_int(out[i]) == want


@pytest.mark.parametrize("k", [2, 3, 11, 14, 17])
def test_coset_zeta_fast_path(gpu_ctx, oracle_c, k):
    """halo2's extended coset generator is ZETA (a primitive cube root of unity): the library takes a
    3-constant fast path for shift^3 == 1; results must equal the general-shift oracle."""
    F = O.FP
    zeta = pow(F.g, (F.p - 1) // 3, F.p)
    assert pow(zeta, 3, F.p) == 1 and zeta != 1
    rng = np.random.default_rng(300 + k)
    a = rand_elems(rng, 1 << k)
    w = F.omega(k)
    fwd = gpu_ctx.ntt(0, a, omega=w, coset_shift=zeta)
    assert (fwd == C.ntt(0, a, w, coset_shift=zeta, threads=8)).all()
    inv = gpu_ctx.ntt(0, a, omega=w, inverse=True, coset_shift=zeta)
    assert (inv == C.ntt(0, a, w, inverse=True, coset_shift=zeta, threads=8)).all()
    assert (gpu_ctx.ntt(0, fwd, omega=w, inverse=True, coset_shift=zeta) == a).all()


@pytest.mark.parametrize("k", [1, 11, 12, 19])
def test_plain_inverse_all_pass_counts(gpu_ctx, oracle_c, k):
    """n^-1 is a constant post-multiply for one pass and folded into the first inter-pass twiddle for
    two (k=12) and three (k=19) passes."""
    F = O.FQ
    rng = np.random.default_rng(400 + k)
    a = rand_elems(rng, 1 << k, 1)
    w = F.omega(k)
    assert (gpu_ctx.ntt(1, a, omega=w, inverse=True) == C.ntt(1, a, w, inverse=True, threads=8)).all()


@pytest.mark.parametrize("k,ext", [(3, 3), (5, 2), (8, 3), (9, 3), (11, 3), (11, 2), (12, 1), (14, 3), (10, 4), (6, 0), (12, 0), (4, 8), (3, 10)])
@pytest.mark.parametrize("shift", ["zeta", "generator", None])
def test_coeff_to_extended_matches_padded_oracle_ntt(gpu_ctx, oracle_c, k, ext, shift):
    """bzh_coeff_to_extended reads only the 2^k coefficients; the result is the coset NTT of the zero-padded vector
    (EvaluationDomain::coeff_to_extended).  ZETA is the shift create_proof uses (a cube root of unity: three-valued
    scaling path); the multiplicative generator exercises the general power table; None the plain transform."""
    F = O.FP
    batch = 3
    rng = np.random.default_rng(1000 * k + ext)
    a = rand_elems(rng, batch << k).reshape(batch, 1 << k, 4)
    ek = k + ext
    w = F.omega(ek)
    s = {"zeta": pow(F.g, (F.p - 1) // 3, F.p), "generator": F.g, None: None}[shift]
    got = gpu_ctx.coeff_to_extended(0, a, ek, omega_ext=w, coset_shift=s)
    assert got.shape == (batch, 1 << ek, 4)
    for b in range(batch):
        padded = np.zeros((1 << ek, 4), dtype=np.uint64)
        padded[: 1 << k] = a[b]
        assert (got[b] == C.ntt(0, padded, w, coset_shift=s, threads=8)).all()


def test_coeff_to_extended_montgomery_device_path_leaves_source_intact(gpu_ctx, oracle_c):
    import ctypes
    import torch
    import bzh2
    F = O.FP
    k, ek, batch = 11, 14, 5
    rng = np.random.default_rng(99)
    a = rand_elems(rng, batch << k).reshape(batch, 1 << k, 4)
    R, p = F.R, F.p
    am = C.ints_to_array([x * R % p for x in C.array_to_ints(a.reshape(-1, 4))]).reshape(batch, 1 << k, 4)
    src = torch.from_numpy(am.view(np.int64)).cuda()
    dst = torch.empty((batch, 1 << ek, 4), dtype=torch.int64, device="cuda")
    zeta = pow(F.g, (F.p - 1) // 3, F.p)
    L = bzh2.load()
    L.bzh_coeff_to_extended.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint,
                                        ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                        ctypes.c_int, ctypes.c_int]
    wm = bzh2.int_to_limbs(F.omega(ek) * R % p)
    zm = bzh2.int_to_limbs(zeta * R % p)
    torch.cuda.synchronize()
    rc = L.bzh_coeff_to_extended(gpu_ctx.handle, 0, ctypes.c_void_p(src.data_ptr()), k, ctypes.c_void_p(dst.data_ptr()), ek, batch,
                                 wm.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), zm.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                 bzh2.FORM_MONTGOMERY, bzh2.MEM_DEVICE)
    assert rc == 0
    gpu_ctx.sync()
    assert (src.cpu().numpy().view(np.uint64) == am).all()
    got = dst.cpu().numpy().view(np.uint64)
    for b in range(batch):
        padded = np.zeros((1 << ek, 4), dtype=np.uint64)
        padded[: 1 << k] = a[b]
        want = C.array_to_ints(C.ntt(0, padded, F.omega(ek), coset_shift=zeta, threads=8))
        assert C.array_to_ints(got[b]) == [x * R % p for x in want]
