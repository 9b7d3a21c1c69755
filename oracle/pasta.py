"""CPU ORACLE (test infrastructure, NOT product code) -- Python big-int restatement.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing under battlezips-halo2_amd/ imports it.

What it restates
----------------
The arithmetic behind the reference's `create_proof` hot path lives in crates
that are NOT vendored under /root/reference:
  pasta_curves 0.4.1 (Cargo.lock:567-570)  -- Fp/Fq, Pallas/Vesta, hash_to_curve
  halo2_proofs 0.2.0 (Cargo.lock:382-385)  -- best_multiexp, best_fft, domain
  ff 0.12.1 / group 0.12.1                  -- encodings
Their *published* algorithms are restated here from first principles with
Python integers; the reference's own call sites anchor the semantics:
  src/utils/pedersen.rs:17-28        [m]V + [t]R on Pallas, V,R = hash_to_curve
  src/utils/binary.rs:36,58          canonical 32-byte little-endian to_repr
  benches/shot.rs:58-71              Params::new(k) / create_proof (MSM + FFT sizes)
  src/chips/bitify.rs:461            the Fp modulus literal

Pinning (see tests/test_oracle_golden.py):
  * modulus literal (bitify.rs:461)
  * both fixed-base GENERATORs on Pallas (board_commit_{v,r}.rs:5-14)
  * U/Z window-table relation U^2 - Z = y([(k+2)*8^w]B)   (board_commit_*.rs:17-2927)
    -> pins Fp arithmetic, Pallas add/double/scalar-mul, Fq reduction of scalars
  * hash_to_curve("battlezips:hash2curve")(b"v"/b"r") == GENERATOR
    (board_commit_*.rs:2941-2948) -> pins Blake2b XMD, SWU, 3-isogeny
MSM / NTT outputs themselves have no golden vector in the reference (SURVEY.md
F5, section 8c): they are mathematically unique under canonical encodings, and
are pinned by linearity / round-trip / O(n^2)-definition checks instead.
"""
from __future__ import annotations

import hashlib

# --------------------------------------------------------------------------
# Fields (modulus literal also at src/chips/bitify.rs:461)
# --------------------------------------------------------------------------
P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001  # Fp: Pallas base = Vesta scalar
Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001  # Fq: Pallas scalar = Vesta base
# BN254 (config 5 microbench only; no reference)
BN_Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
BN_R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001

R256 = 1 << 256


class FieldSpec:
    """Static description of a prime field (modulus, 2-adicity, generator)."""

    def __init__(self, name: str, modulus: int, two_adicity: int, mult_gen: int):
        self.name = name
        self.p = modulus
        self.S = two_adicity
        self.g = mult_gen
        assert (modulus - 1) % (1 << two_adicity) == 0
        # primitive 2^S-th root of unity: g^((p-1)/2^S)  (pasta_curves ROOT_OF_UNITY)
        self.root = pow(mult_gen, (modulus - 1) >> two_adicity, modulus)
        assert pow(self.root, 1 << (two_adicity - 1), modulus) == modulus - 1
        self.R = R256 % modulus
        self.R2 = (self.R * self.R) % modulus
        self.inv32 = (-pow(modulus, -1, 1 << 32)) % (1 << 32)
        self.inv64 = (-pow(modulus, -1, 1 << 64)) % (1 << 64)

    def omega(self, k: int) -> int:
        """Primitive 2^k-th root: root^(2^(S-k))  (halo2 EvaluationDomain::new)."""
        assert 0 <= k <= self.S
        return pow(self.root, 1 << (self.S - k), self.p)

    def inv(self, a: int) -> int:
        return pow(a, self.p - 2, self.p)

    def sqrt(self, a: int):
        """Tonelli-Shanks; returns a root or None."""
        p = self.p
        a %= p
        if a == 0:
            return 0
        if pow(a, (p - 1) // 2, p) != 1:
            return None
        s, t = self.S, (p - 1) >> self.S
        z = self.root  # generator of the 2-Sylow subgroup
        x = pow(a, (t + 1) // 2, p)
        b = pow(a, t, p)
        m = s
        while b != 1:
            i, b2 = 0, b
            while b2 != 1:
                b2 = b2 * b2 % p
                i += 1
            w = pow(z, 1 << (m - i - 1), p)
            z = w * w % p
            x = x * w % p
            b = b * z % p
            m = i
        assert x * x % p == a
        return x


FP = FieldSpec("fp", P, 32, 5)
FQ = FieldSpec("fq", Q, 32, 5)
BN_FR = FieldSpec("bn254_fr", BN_R, 28, 7)
BN_FQ = FieldSpec("bn254_fq", BN_Q, 1, 3)

FIELD_BY_ID = {0: FP, 1: FQ, 2: BN_FR, 3: BN_FQ}


def to_repr(x: int) -> bytes:
    """Canonical 32-byte little-endian (ff::PrimeField::to_repr; src/utils/binary.rs:36)."""
    return int(x).to_bytes(32, "little")


def from_repr(b: bytes, field: FieldSpec = FP):
    v = int.from_bytes(b, "little")
    return v if v < field.p else None


def from_u512(b64: bytes, field: FieldSpec) -> int:
    """pasta_curves from_bytes_wide / Field::random: 512-bit LE reduced mod p."""
    return int.from_bytes(b64, "little") % field.p


# --------------------------------------------------------------------------
# Short-Weierstrass curves  y^2 = x^3 + a x + b  (affine big-int arithmetic)
# --------------------------------------------------------------------------
class Curve:
    def __init__(self, name: str, base: FieldSpec, scalar: FieldSpec, a: int, b: int):
        self.name, self.base, self.scalar, self.a, self.b = name, base, scalar, a, b
        self.p = base.p

    def is_on_curve(self, pt) -> bool:
        if pt is None:
            return True
        x, y = pt
        return (y * y - (x * x * x + self.a * x + self.b)) % self.p == 0

    def neg(self, pt):
        return None if pt is None else (pt[0], (-pt[1]) % self.p)

    def add(self, p1, p2):
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        p = self.p
        x1, y1 = p1
        x2, y2 = p2
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = (3 * x1 * x1 + self.a) * pow(2 * y1, p - 2, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, p - 2, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return (x3, (lam * (x1 - x3) - y1) % p)

    def mul(self, k: int, pt):
        k %= self.scalar.p
        acc = None
        while k:
            if k & 1:
                acc = self.add(acc, pt)
            pt = self.add(pt, pt)
            k >>= 1
        return acc

    def msm_naive(self, scalars, points):
        """Definition of multiexp: sum_i [s_i] G_i."""
        acc = None
        for s, g in zip(scalars, points):
            acc = self.add(acc, self.mul(s, g))
        return acc

    def msm_pippenger(self, scalars, points, c: int | None = None):
        """Bucket method as published for halo2 `best_multiexp` (arithmetic.rs,
        UPSTREAM/unvendored): c = 3 if n<4, ceil(ln n) if n<32, else ceil(ln n)
        (upstream: `if n < 4 {1} else if n < 32 {3} else {ceil(ln n)}`); unsigned
        c-bit segments of the canonical repr, running-sum bucket reduction,
        c doublings between segments (high to low)."""
        import math
        n = len(scalars)
        if c is None:
            c = 1 if n < 4 else (3 if n < 32 else math.ceil(math.log(n)))
        segments = (256 // c) + 1
        acc = None
        for seg in reversed(range(segments)):
            for _ in range(c):
                acc = self.add(acc, acc)
            buckets = [None] * ((1 << c) - 1)
            for s, g in zip(scalars, points):
                d = (s >> (seg * c)) & ((1 << c) - 1)
                if d:
                    buckets[d - 1] = self.add(buckets[d - 1], g)
            run = None
            for b in reversed(buckets):
                run = self.add(run, b)
                acc = self.add(acc, run)
        return acc

    def compress(self, pt) -> bytes:
        """pasta_curves to_bytes: x LE, bit 255 = parity of y; identity = zeros."""
        if pt is None:
            return bytes(32)
        x, y = pt
        b = bytearray(to_repr(x))
        b[31] |= (y & 1) << 7
        return bytes(b)

    def random_point(self, rng) -> tuple:
        while True:
            x = rng.randrange(self.p)
            y = self.base.sqrt((x * x * x + self.a * x + self.b) % self.p)
            if y is not None:
                if rng.getrandbits(1):
                    y = (-y) % self.p
                return (x, y)


PALLAS = Curve("pallas", FP, FQ, 0, 5)
VESTA = Curve("vesta", FQ, FP, 0, 5)
BN254 = Curve("bn254", BN_FQ, BN_FR, 0, 3)
CURVE_BY_ID = {0: VESTA, 1: PALLAS, 2: BN254}

# --------------------------------------------------------------------------
# hash_to_curve (pasta_curves 0.4.1 src/hashtocurve.rs, UPSTREAM/unvendored;
# call site: src/utils/pedersen.rs:19-21).  draft-irtf-cfrg-hash-to-curve-10
# XMD with BLAKE2b-512, simplified SWU on the 3-isogenous curve, then iso_map.
# --------------------------------------------------------------------------
ISO_A = {
    "pallas": 0x18354a2eb0ea8c9c49be2d7258370742b74134581a27a59f92bb4b0b657a014b,
    "vesta": 0x267f9b2ee592271a81639c4d96f787739673928c7d01b212c515ad7242eaa6b1,
}
ISO_B = 1265
SWU_Z = -13


def _blake2b(data: bytes) -> bytes:
    return hashlib.blake2b(data, digest_size=64, person=bytes(16)).digest()


def hash_to_field(curve_id: str, domain_prefix: str, message: bytes, field: FieldSpec):
    """Two field elements from expand_message_xmd(BLAKE2b), len_in_bytes = 128."""
    dst = (domain_prefix.encode() + b"-" + curve_id.encode() + b"_XMD:BLAKE2b_SSWU_RO_")
    dst_prime = dst + bytes([len(dst)])
    assert len(dst) == 22 + len(curve_id) + len(domain_prefix)
    b0 = _blake2b(bytes(128) + message + bytes([0, 128, 0]) + dst_prime)
    b1 = _blake2b(b0 + b"\x01" + dst_prime)
    b2 = _blake2b(bytes(x ^ y for x, y in zip(b0, b1)) + b"\x02" + dst_prime)
    # big-endian OS2IP of each 64-byte chunk, reduced mod p
    return [int.from_bytes(b, "big") % field.p for b in (b1, b2)]


def map_to_curve_simple_swu(u: int, iso: Curve):
    """Simplified SWU for AB != 0 (RFC 9380 6.6.2) onto the iso-curve; affine."""
    F = iso.base
    p = F.p
    A, B, Z = iso.a, iso.b, SWU_Z % p
    zu2 = Z * u * u % p
    ta = (zu2 * zu2 + zu2) % p
    if ta == 0:
        x1 = B * F.inv(Z * A % p) % p
    else:
        x1 = (-B) * F.inv(A) % p * (1 + F.inv(ta)) % p
    gx1 = (x1 * x1 * x1 + A * x1 + B) % p
    y1 = F.sqrt(gx1)
    if y1 is not None:
        x, y = x1, y1
    else:
        x = zu2 * x1 % p
        gx2 = (x * x * x + A * x + B) % p
        y = F.sqrt(gx2)
        assert y is not None
    if (u & 1) != (y & 1):  # sgn0(u) == sgn0(y)
        y = (-y) % p
    return (x, y)


def _poly_eval(cs, x, p):
    acc = 0
    for c in cs:  # highest degree first
        acc = (acc * x + c) % p
    return acc


def derive_isogeny(iso: Curve, target: Curve):
    """3-isogeny iso -> target by Velu's formulas followed by the scaling
    (X,Y) -> (X/9, Y/27), i.e. the *normalised* isogeny.

    Kernel: the unique rational root x0 of the 3-division polynomial
        psi3(x) = 3x^4 + 6A x^2 + 12B x - A^2.
    Velu (kernel {O, +-(x0,y0)}):
        t = 6x0^2 + 2A, u = 4 y0^2, w = u + x0 t
        X = x + t/(x-x0) + u/(x-x0)^2
        Y = y * (1 - t/(x-x0)^2 - 2u/(x-x0)^3)
        image curve: A' = A - 5t (= 0 here), B' = B - 7w
    and s = 1/3 satisfies s^6 * B' = b_target.  Among the six scalings with
    s^6 = b_target/B' (the automorphisms of a j=0 curve) s = 1/3 is the one
    the reference's `generator` KATs select for Pallas
    (board_commit_{v,r}.rs:2941-2948; tests/test_oracle_golden.py).  For Vesta
    the reference holds no KAT ("parity unpinned"); the same rule is applied.
    """
    F = iso.base
    p = F.p
    A, B = iso.a, iso.b
    psi = [3, 0, 6 * A % p, 12 * B % p, (-A * A) % p]  # high -> low

    def pmod(a, m):
        a = a[:]
        dm = len(m) - 1
        inv_lead = F.inv(m[0])
        while len(a) - 1 >= dm:
            if a[0]:
                f = a[0] * inv_lead % p
                for i in range(len(m)):
                    a[i] = (a[i] - f * m[i]) % p
            a.pop(0)
        while len(a) > 1 and a[0] == 0:
            a.pop(0)
        return a or [0]

    def pmul(a, b, m):
        r = [0] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    r[i + j] = (r[i + j] + x * y) % p
        return pmod(r, m)

    # gcd(x^p - x, psi3) isolates the rational roots
    result, base, e = [1], [1, 0], p
    while e:
        if e & 1:
            result = pmul(result, base, psi)
        base = pmul(base, base, psi)
        e >>= 1
    a_poly = result[:]
    while len(a_poly) < 2:
        a_poly.insert(0, 0)
    a_poly[-2] = (a_poly[-2] - 1) % p
    while len(a_poly) > 1 and a_poly[0] == 0:
        a_poly.pop(0)
    g, h = psi, a_poly
    while not (len(h) == 1 and h[0] == 0):
        g, h = h, pmod(g, h)
    assert len(g) == 2, "expected exactly one rational 3-torsion x-coordinate"
    x0 = (-g[1]) * F.inv(g[0]) % p
    y0sq = (x0 * x0 * x0 + A * x0 + B) % p
    t = (6 * x0 * x0 + 2 * A) % p
    u = 4 * y0sq % p
    w = (u + x0 * t) % p
    assert (A - 5 * t) % p == 0, "image of the 3-isogeny must have j = 0"
    B2 = (B - 7 * w) % p
    s = F.inv(3)
    assert pow(s, 6, p) * B2 % p == target.b % p
    return (x0, t, u, s)


def iso_map_apply(pt, mp, iso: Curve):
    """Evaluate the normalised Velu isogeny at an affine iso-curve point."""
    if pt is None:
        return None
    F = iso.base
    p = F.p
    x0, t, u, s = mp
    x, y = pt
    d = (x - x0) % p
    if d == 0:
        return None  # kernel point
    di = F.inv(d)
    di2 = di * di % p
    X = (x + t * di + u * di2) % p
    Y = y * (1 - t * di2 - 2 * u * di2 % p * di) % p
    s2 = s * s % p
    return (s2 * X % p, s2 * s % p * Y % p)


_ISO_CURVES = {
    "pallas": Curve("iso-pallas", FP, FQ, ISO_A["pallas"], ISO_B),
    "vesta": Curve("iso-vesta", FQ, FP, ISO_A["vesta"], ISO_B),
}
_TARGET = {"pallas": PALLAS, "vesta": VESTA}
_ISO_MAP_CACHE: dict = {}


def iso_map(curve_id: str):
    if curve_id not in _ISO_MAP_CACHE:
        _ISO_MAP_CACHE[curve_id] = derive_isogeny(_ISO_CURVES[curve_id], _TARGET[curve_id])
    return _ISO_MAP_CACHE[curve_id]


def hash_to_curve(curve_id: str, domain_prefix: str, message: bytes):
    """CurveExt::hash_to_curve(domain_prefix)(message) -> affine (x, y)."""
    iso = _ISO_CURVES[curve_id]
    us = hash_to_field(curve_id, domain_prefix, message, iso.base)
    q0 = map_to_curve_simple_swu(us[0], iso)
    q1 = map_to_curve_simple_swu(us[1], iso)
    return iso_map_apply(iso.add(q0, q1), iso_map(curve_id), iso)


def pedersen_commit(message: int, trapdoor: int):
    """src/utils/pedersen.rs:17-28: [m]V + [t]R on Pallas, V/R hashed per call;
    the Fp message is re-read as an Fq scalar through its canonical repr
    (from_repr(...).unwrap(): values >= q are an error upstream)."""
    V = hash_to_curve("pallas", "battlezips:hash2curve", b"v")
    Rr = hash_to_curve("pallas", "battlezips:hash2curve", b"r")
    if not (0 <= message < Q):
        raise ValueError("message repr is not a canonical Fq element")
    return PALLAS.add(PALLAS.mul(message, V), PALLAS.mul(trapdoor, Rr))


# --------------------------------------------------------------------------
# NTT (halo2_proofs 0.2.0 arithmetic::best_fft, UPSTREAM/unvendored):
# in-place radix-2 DIT, natural order in, natural order out, no scaling;
# the inverse uses omega^-1 and the caller multiplies by n^-1
# (EvaluationDomain::ifft).
# --------------------------------------------------------------------------
def ntt(vals, omega: int, field: FieldSpec):
    p = field.p
    n = len(vals)
    k = n.bit_length() - 1
    assert 1 << k == n
    a = list(vals)
    for i in range(n):
        j = int(format(i, "0%db" % k)[::-1], 2) if k else 0
        if i < j:
            a[i], a[j] = a[j], a[i]
    m = 1
    while m < n:
        wm = pow(omega, n // (2 * m), p)
        for s in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[s + j + m] * w % p
                a[s + j + m] = (a[s + j] - t) % p
                a[s + j] = (a[s + j] + t) % p
                w = w * wm % p
        m *= 2
    return a


def dft_naive(vals, omega: int, field: FieldSpec):
    p = field.p
    n = len(vals)
    return [sum(v * pow(omega, i * j, p) for j, v in enumerate(vals)) % p for i in range(n)]


def intt(vals, omega: int, field: FieldSpec):
    p = field.p
    n = len(vals)
    ninv = field.inv(n)
    return [v * ninv % p for v in ntt(vals, field.inv(omega), field)]


def coset_ntt(coeffs, omega: int, shift: int, field: FieldSpec):
    """Evaluate on the coset shift*<omega>: scale coeff i by shift^i, then NTT
    (EvaluationDomain::coeff_to_extended distributes powers of zeta first)."""
    p = field.p
    s, out = 1, []
    for c in coeffs:
        out.append(c * s % p)
        s = s * shift % p
    return ntt(out, omega, field)


def eval_polynomial(coeffs, x: int, field: FieldSpec) -> int:
    """arithmetic::eval_polynomial: Horner."""
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % field.p
    return acc


# --------------------------------------------------------------------------
# Prover-stage vector primitives (halo2_proofs 0.2.0, UPSTREAM/unvendored),
# restated from their published definitions; call chain: create_proof
# (benches/shot.rs:68) -> permutation/lookup/vanishing/multiopen provers.
# --------------------------------------------------------------------------
def batch_invert(vals, field: FieldSpec):
    """ff::BatchInvert semantics: every non-zero element is inverted, zeros stay zero."""
    return [0 if v % field.p == 0 else field.inv(v) for v in vals]


def prefix_product(vals, field: FieldSpec):
    """Exclusive running product: out[0] = 1, out[i] = prod_{j<i} vals[j]  (the z(X) grand
    products of permutation::Argument::commit / lookup commit_product start at 1)."""
    out, acc = [], 1
    for v in vals:
        out.append(acc)
        acc = acc * v % field.p
    return out


def inner_product(a, b, field: FieldSpec) -> int:
    """arithmetic::compute_inner_product."""
    return sum(x * y for x, y in zip(a, b)) % field.p


def fold_scalars(v, u: int, field: FieldSpec):
    """IPA round fold of a scalar vector: lo + u * hi  (commitment::prover, p' and b)."""
    h = len(v) // 2
    return [(v[i] + u * v[i + h]) % field.p for i in range(h)]


def fold_bases(curve: "Curve", g, u: int):
    """parallel_generator_collapse: g_lo[i] + [u] g_hi[i], affine."""
    h = len(g) // 2
    return [curve.add(g[i], curve.mul(u, g[i + h])) for i in range(h)]


def kate_division(coeffs, x: int, field: FieldSpec):
    """arithmetic::kate_division: quotient of p(X) by (X - x), remainder dropped; len = len(coeffs)-1."""
    p = field.p
    q = [0] * (len(coeffs) - 1)
    tmp = 0
    for i in range(len(coeffs) - 1, 0, -1):
        tmp = (coeffs[i] + tmp * x) % p
        q[i - 1] = tmp
    return q


# --------------------------------------------------------------------------
# Transcript (halo2_proofs 0.2.0 transcript.rs, UPSTREAM/unvendored):
# Blake2b-512 with personal "Halo2-Transcript"; point = 0x01? no: prefixes are
# BLAKE2B_PREFIX_CHALLENGE = 0, _POINT = 1, _SCALAR = 2; a challenge hashes the
# running state + [0] and reduces the 64-byte digest as a 512-bit LE integer.
# --------------------------------------------------------------------------
class Blake2bTranscript:
    def __init__(self, field: FieldSpec = FP):
        self.field = field
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof = bytearray()

    def common_point(self, curve: "Curve", pt):
        x, y = (0, 0) if pt is None else pt
        self.state.update(b"\x01" + to_repr(x) + to_repr(y))

    def common_scalar(self, s: int):
        self.state.update(b"\x02" + to_repr(s))

    def write_point(self, curve: "Curve", pt):
        self.common_point(curve, pt)
        self.proof += curve.compress(pt)

    def write_scalar(self, s: int):
        self.common_scalar(s)
        self.proof += to_repr(s)

    def squeeze_challenge(self) -> int:
        self.state.update(b"\x00")
        digest = self.state.copy().digest()
        return int.from_bytes(digest, "little") % self.field.p


# --------------------------------------------------------------------------
# Inner-product-argument opening (halo2_proofs 0.2.0 poly/commitment/prover.rs
# `create_proof` and the matching verification equation, UPSTREAM/unvendored;
# reached from plonk::create_proof step 9).  Straight
# restatement WITH the generator collapse, used to pin the GPU prover (which
# never collapses generators) byte for byte under a shared randomness stream.
# --------------------------------------------------------------------------
def ipa_open(curve: "Curve", g, w, u, poly, blind: int, x3: int, rand_scalars, transcript: "Blake2bTranscript"):
    """g: n affine bases, w/u: blinding / inner-product bases, poly: n coefficients.
    rand_scalars: iterator of field elements in upstream draw order
    (n for s(X), 1 s_blind, then l_j, r_j per round).  Writes S, (L_j, R_j)*, c, f."""
    F = curve.scalar
    p = F.p
    n = len(poly)
    k = n.bit_length() - 1
    assert 1 << k == n == len(g)
    rnd = iter(rand_scalars)
    s_poly = [next(rnd) for _ in range(n)]
    s_poly[0] = (s_poly[0] - eval_polynomial(s_poly, x3, F)) % p
    s_blind = next(rnd)
    S = curve.add(curve.msm_naive(s_poly, g), curve.mul(s_blind, w))
    transcript.write_point(curve, S)
    xi = transcript.squeeze_challenge()
    z = transcript.squeeze_challenge()
    pp = [(a * xi + b) % p for a, b in zip(s_poly, poly)]
    v = eval_polynomial(pp, x3, F)
    pp[0] = (pp[0] - v) % p
    f = (s_blind * xi + blind) % p
    b = [pow(x3, i, p) for i in range(n)]
    gp = list(g)
    for j in range(k):
        half = 1 << (k - j - 1)
        l_pt = curve.msm_naive(pp[half:], gp[:half])
        r_pt = curve.msm_naive(pp[:half], gp[half:])
        vl = inner_product(pp[half:], b[:half], F)
        vr = inner_product(pp[:half], b[half:], F)
        lr, rr = next(rnd), next(rnd)
        l_pt = curve.add(l_pt, curve.add(curve.mul(vl * z % p, u), curve.mul(lr, w)))
        r_pt = curve.add(r_pt, curve.add(curve.mul(vr * z % p, u), curve.mul(rr, w)))
        transcript.write_point(curve, l_pt)
        transcript.write_point(curve, r_pt)
        uj = transcript.squeeze_challenge()
        uj_inv = F.inv(uj)
        pp = [(pp[i] + pp[i + half] * uj_inv) % p for i in range(half)]
        b = [(b[i] + b[i + half] * uj) % p for i in range(half)]
        gp = fold_bases(curve, gp, uj)
        f = (f + lr * uj_inv + rr * uj) % p
    transcript.write_scalar(pp[0])
    transcript.write_scalar(f)
    return v


def ipa_verify(curve: "Curve", g, w, u, commitment, x3: int, v: int, proof: bytes, transcript: "Blake2bTranscript") -> bool:
    """sum_j (u_j^-1 L_j + u_j R_j) + P - [v]G_0 + [xi]S == [c]G'_0 + [c b_0 z]U + [f]W.  A malformed proof (a point
    off the curve, the identity where a transcript point is read) does not verify."""
    try:
        return _ipa_verify(curve, g, w, u, commitment, x3, v, proof, transcript)
    except ValueError:
        return False


def _ipa_verify(curve: "Curve", g, w, u, commitment, x3: int, v: int, proof: bytes, transcript: "Blake2bTranscript") -> bool:
    F = curve.scalar
    p = F.p
    n = len(g)
    k = n.bit_length() - 1

    def read_point(off):
        raw = bytearray(proof[off:off + 32])
        ysign = raw[31] >> 7
        raw[31] &= 0x7f
        x = int.from_bytes(raw, "little")
        if x == 0 and ysign == 0:
            # upstream's Blake2bRead::common_point errors on the identity: such a proof does not verify
            raise ValueError("identity point in proof")
        y = curve.base.sqrt((x * x * x + curve.a * x + curve.b) % curve.p)
        if y is None:
            raise ValueError("not on curve")
        if (y & 1) != ysign:
            y = curve.p - y
        return (x, y)

    off = 0
    S = read_point(off); off += 32
    transcript.common_point(curve, S)
    xi = transcript.squeeze_challenge()
    z = transcript.squeeze_challenge()
    acc = curve.add(curve.add(commitment, curve.neg(curve.mul(v, g[0]))), curve.mul(xi, S))
    us = []
    for _ in range(k):
        L = read_point(off); R = read_point(off + 32); off += 64
        transcript.common_point(curve, L)
        transcript.common_point(curve, R)
        uj = transcript.squeeze_challenge()
        us.append(uj)
        acc = curve.add(acc, curve.add(curve.mul(F.inv(uj), L), curve.mul(uj, R)))
    c = int.from_bytes(proof[off:off + 32], "little"); off += 32
    f = int.from_bytes(proof[off:off + 32], "little"); off += 32
    if off != len(proof) or c >= p or f >= p:
        return False
    s = [1]
    for uj in us:                                   # s^{(j+1)}[2t + beta] = s^{(j)}[t] * u_j^beta
        s = [x * (uj if beta else 1) % p for x in s for beta in (0, 1)]
    g0 = curve.msm_naive(s, g)
    b0 = 1
    for j, uj in enumerate(us):
        b0 = b0 * (1 + uj * pow(x3, 1 << (k - 1 - j), p)) % p
    rhs = curve.add(curve.mul(c, g0), curve.add(curve.mul(c * b0 % p * z % p, u), curve.mul(f, w)))
    return acc == rhs


def permute_expression_pair(input_vals, table_vals, usable_rows: int, field: FieldSpec):
    """lookup::prover::permute_expression_pair (halo2_proofs 0.2.0, UPSTREAM/unvendored), on the first
    `usable_rows` rows (blinding rows are appended by the caller):
      A' = input sorted ascending (by canonical value);
      S'[i] = A'[i] wherever A'[i] starts a new run (i == 0 or A'[i] != A'[i-1]); the table values not
      consumed that way are taken in ascending order (BTreeMap iteration) and written to the repeated rows
      from the LAST repeated row backwards (`repeated_input_rows.pop()`).
    Raises if some input value is missing from the table (Error::ConstraintSystemFailure upstream)."""
    a = sorted(input_vals[:usable_rows])
    leftover = {}
    for t in table_vals[:usable_rows]:
        leftover[t] = leftover.get(t, 0) + 1
    s = [None] * usable_rows
    repeated = []
    for i, v in enumerate(a):
        if i == 0 or v != a[i - 1]:
            if leftover.get(v, 0) == 0:
                raise ValueError("lookup input not in table")
            s[i] = v
            leftover[v] -= 1
        else:
            repeated.append(i)
    rest = []
    for t in sorted(leftover):
        rest += [t] * leftover[t]
    assert len(rest) == len(repeated)
    for t in rest:
        s[repeated.pop()] = t
    return a, s
