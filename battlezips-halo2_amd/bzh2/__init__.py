"""bzh2 -- ctypes host binding over the C ABI of include/bzh2.h (libbzh2.so).

The product path: every call below lands in hand-written HIP kernels for
gfx950.  There is NO CPU fallback: if libbzh2.so is missing or there is no GPU,
calls raise.  This module never imports anything from oracle/.

Host-side mirror of the interface the reference reaches for the create_proof hot
path (halo2_proofs 0.2.0, un-vendored):
    best_multiexp(coeffs, bases)            -> Context.msm(bases, scalars)
    Params::commit / commit_lagrange        -> Context.upload_bases + Context.msm
    best_fft(a, omega, log_n)               -> Context.ntt(...)
    EvaluationDomain::ifft / coeff_to_extended / extended_to_coeff
                                            -> Context.ntt(inverse=, coset_shift=)
Names, argument meaning and error behaviour follow the upstream functions: a
length mismatch between scalars and bases is an error (upstream asserts), field
elements are 4 x u64 little-endian limbs.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(os.path.dirname(_PKG), "libbzh2.so")

OK, E_ARG, E_OOM, E_HIP, E_RANGE, E_NOGPU = 0, -1, -2, -3, -4, -5
CURVE_VESTA, CURVE_PALLAS, CURVE_BN254 = 0, 1, 2
FIELD_FP, FIELD_FQ, FIELD_BN254_FR, FIELD_BN254_FQ = 0, 1, 2, 3
FORM_CANONICAL, FORM_MONTGOMERY = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
T_MSM_DIGITS, T_MSM_ACCUMULATE, T_MSM_REDUCE, T_MSM_FINALIZE, T_NTT, T_COUNT = 0, 1, 2, 3, 4, 8
TIMER_NAMES = {0: "msm_digits", 1: "msm_accumulate", 2: "msm_reduce", 3: "msm_finalize", 4: "ntt"}

# scalar field of each curve (Vesta scalars are Fp, Pallas scalars are Fq)
CURVE_SCALAR_FIELD = {CURVE_VESTA: FIELD_FP, CURVE_PALLAS: FIELD_FQ, CURVE_BN254: FIELD_BN254_FR}

# every symbol include/bzh2.h declares
EXPORTS = [
    "bzh_version", "bzh_strerror", "bzh_device_count", "bzh_ctx_create", "bzh_ctx_create_on_stream",
    "bzh_ctx_destroy", "bzh_ctx_sync", "bzh_last_error", "bzh_ctx_profile", "bzh_ctx_timings", "bzh_ctx_work", "bzh_ctx_msm_additions",
    "bzh_bases_upload", "bzh_bases_precompute", "bzh_bases_free", "bzh_bases_len", "bzh_bases_walk", "bzh_bases_points", "bzh_msm", "bzh_ntt", "bzh_coeff_to_extended",
    "bzh_jacobian_to_affine", "bzh_jacobian_sum", "bzh_affine_compress", "bzh_field_omega",
]


class BzhError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__("%s failed: %s (%d)%s" % (where, _strerror(status), status, (": " + detail) if detail else ""))


_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load libbzh2.so or raise -- the product path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError("libbzh2.so not built at %s: run `python -c 'import __graft_entry__ as g; g.build()'`" % _LIB_PATH)
    L = ctypes.CDLL(_LIB_PATH)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    vp = ctypes.c_void_p
    L.bzh_version.restype = ctypes.c_char_p
    L.bzh_strerror.restype = ctypes.c_char_p
    L.bzh_strerror.argtypes = [ctypes.c_int]
    L.bzh_device_count.restype = ctypes.c_int
    L.bzh_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.bzh_ctx_create_on_stream.argtypes = [ctypes.c_int, vp, ctypes.POINTER(vp)]
    L.bzh_ctx_destroy.argtypes = [vp]
    L.bzh_ctx_sync.argtypes = [vp]
    L.bzh_last_error.argtypes = [vp]
    L.bzh_last_error.restype = ctypes.c_char_p
    L.bzh_ctx_profile.argtypes = [vp, ctypes.c_int]
    L.bzh_ctx_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_double), u64p]
    L.bzh_bases_upload.argtypes = [vp, ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]
    L.bzh_bases_precompute.argtypes = [vp, vp, ctypes.c_int]
    L.bzh_bases_free.argtypes = [vp, vp]
    L.bzh_bases_len.argtypes = [vp]
    L.bzh_bases_len.restype = ctypes.c_size_t
    L.bzh_msm.argtypes = [vp, vp, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, vp]
    L.bzh_ntt.argtypes = [vp, ctypes.c_int, vp, ctypes.c_uint, ctypes.c_size_t, u64p, u64p, ctypes.c_int, ctypes.c_int,
                          ctypes.c_int]
    L.bzh_jacobian_to_affine.argtypes = [ctypes.c_int, u64p, ctypes.c_size_t, ctypes.c_int, u64p]
    L.bzh_affine_compress.argtypes = [ctypes.c_int, u64p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint8)]
    L.bzh_field_omega.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_int, u64p]
    _lib = L
    return L


def _strerror(status: int) -> str:
    try:
        return load().bzh_strerror(status).decode()
    except Exception:  # pragma: no cover
        return "status %d" % status


def device_count() -> int:
    return load().bzh_device_count()


def _u64(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], "need contiguous uint64"
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def int_to_limbs(x: int) -> np.ndarray:
    return np.frombuffer(int(x).to_bytes(32, "little"), dtype=np.uint64).copy()


def limbs_to_int(a) -> int:
    return int.from_bytes(np.ascontiguousarray(a, dtype=np.uint64).tobytes(), "little")


class Bases:
    """Device-resident commitment bases (halo2 Params.g / g_lagrange)."""

    def __init__(self, ctx: "Context", handle, curve: int, n: int):
        self.ctx, self.handle, self.curve, self.n = ctx, handle, curve, n

    def precompute(self, window_bits: int = 0):
        """Expand into the fixed-base window table (bzh_bases_precompute)."""
        self.ctx._check(load().bzh_bases_precompute(self.ctx.handle, self.handle, window_bits), "bzh_bases_precompute")
        return self

    def free(self):
        if self.handle is not None:
            load().bzh_bases_free(self.ctx.handle, self.handle)
            self.handle = None

    def __len__(self):
        return self.n


class Context:
    """One device + one HIP stream (bzh_ctx)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        L = load()
        h = ctypes.c_void_p()
        if stream is None:
            rc = L.bzh_ctx_create(device, ctypes.byref(h))
        else:
            rc = L.bzh_ctx_create_on_stream(device, ctypes.c_void_p(stream), ctypes.byref(h))
        if rc != OK:
            raise BzhError(rc, "bzh_ctx_create")
        self.handle = h
        self.device = device

    def _check(self, rc: int, where: str):
        if rc != OK:
            raise BzhError(rc, where, load().bzh_last_error(self.handle).decode())

    def close(self):
        if self.handle is not None:
            load().bzh_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._check(load().bzh_ctx_sync(self.handle), "bzh_ctx_sync")

    def profile(self, enable: bool = True):
        self._check(load().bzh_ctx_profile(self.handle, int(enable)), "bzh_ctx_profile")

    def timings(self) -> dict:
        ms = (ctypes.c_double * T_COUNT)()
        n = (ctypes.c_uint64 * T_COUNT)()
        self._check(load().bzh_ctx_timings(self.handle, ms, n), "bzh_ctx_timings")
        wb = (ctypes.c_double * T_COUNT)()
        load().bzh_ctx_work.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        self._check(load().bzh_ctx_work(self.handle, wb), "bzh_ctx_work")
        return {TIMER_NAMES[i]: {"ms": ms[i], "launches": int(n[i]), "algorithmic_bytes": wb[i]} for i in TIMER_NAMES}

    def msm_additions(self) -> int:
        """bucket additions made by the MSM accumulation since profile(True)"""
        v = ctypes.c_uint64()
        L = load()
        L.bzh_ctx_msm_additions.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        self._check(L.bzh_ctx_msm_additions(self.handle, ctypes.byref(v)), "bzh_ctx_msm_additions")
        return int(v.value)

    # ---- bases ----
    def upload_bases(self, curve: int, xy, n: int | None = None, form: int = FORM_CANONICAL, device_ptr: bool = False) -> Bases:
        h = ctypes.c_void_p()
        if device_ptr:
            ptr, mem = ctypes.c_void_p(int(xy)), MEM_DEVICE
            assert n is not None
        else:
            xy = np.ascontiguousarray(xy, dtype=np.uint64)
            n = xy.size // 8 if n is None else n
            ptr, mem = ctypes.c_void_p(xy.ctypes.data), MEM_HOST
        self._check(load().bzh_bases_upload(self.handle, curve, ptr, n, form, mem, ctypes.byref(h)), "bzh_bases_upload")
        return Bases(self, h, curve, n)

    def bases_walk(self, curve: int, g_xy, n: int, form: int = FORM_CANONICAL) -> Bases:
        """bases[i] = [i + 1] G for i < n, made on the device (bzh_bases_walk); g_xy: 8 limbs of one affine point"""
        g = np.ascontiguousarray(g_xy, dtype=np.uint64).reshape(8)
        h = ctypes.c_void_p()
        L = load()
        L.bzh_bases_walk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        self._check(L.bzh_bases_walk(self.handle, curve, ctypes.c_void_p(g.ctypes.data), form, n, ctypes.byref(h)), "bzh_bases_walk")
        return Bases(self, h, curve, n)

    def bases_points(self, bases: Bases, first: int, count: int) -> np.ndarray:
        """points [first, first + count) of a table as (count, 8) canonical limbs (bzh_bases_points)"""
        out = np.zeros((count, 8), dtype=np.uint64)
        L = load()
        L.bzh_bases_points.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
        self._check(L.bzh_bases_points(self.handle, bases.handle, first, count, ctypes.c_void_p(out.ctypes.data)), "bzh_bases_points")
        return out

    # ---- MSM ----
    def msm(self, bases: Bases, scalars: np.ndarray, form: int = FORM_CANONICAL) -> np.ndarray:
        """best_multiexp for `batch` scalar vectors: scalars (batch, n, 4) or (n, 4) uint64 -> (batch, 12) Jacobian."""
        s = np.ascontiguousarray(scalars, dtype=np.uint64)
        if s.ndim == 2:
            s = s.reshape(1, *s.shape)
        batch, n = s.shape[0], s.shape[1]
        assert s.shape[2] == 4
        out = np.zeros((batch, 12), dtype=np.uint64)
        rc = load().bzh_msm(self.handle, bases.handle, ctypes.c_void_p(s.ctypes.data), n, batch, form, MEM_HOST,
                            ctypes.c_void_p(out.ctypes.data))
        self._check(rc, "bzh_msm")
        return out

    def msm_device(self, bases: Bases, scalars_ptr: int, n: int, batch: int, out_ptr: int, form: int = FORM_MONTGOMERY):
        """Device-pointer form: enqueues on the ctx stream, no copies, no sync."""
        rc = load().bzh_msm(self.handle, bases.handle, ctypes.c_void_p(scalars_ptr), n, batch, form, MEM_DEVICE,
                            ctypes.c_void_p(out_ptr))
        self._check(rc, "bzh_msm")

    # ---- NTT ----
    def ntt(self, field: int, data: np.ndarray, omega: int | None = None, inverse: bool = False,
            coset_shift: int | None = None, form: int = FORM_CANONICAL) -> np.ndarray:
        """best_fft over (batch, n, 4) or (n, 4) uint64; returns a new array."""
        a = np.ascontiguousarray(data, dtype=np.uint64).copy()
        shape = a.shape
        if a.ndim == 2:
            a = a.reshape(1, *a.shape)
        batch, n = a.shape[0], a.shape[1]
        log_n = n.bit_length() - 1
        if n == 0 or (1 << log_n) != n:
            raise BzhError(E_ARG, "bzh_ntt", "length must be a power of two")
        w = field_omega(field, log_n, form) if omega is None else int_to_limbs(omega)
        cs = None if coset_shift is None else int_to_limbs(coset_shift)
        rc = load().bzh_ntt(self.handle, field, ctypes.c_void_p(a.ctypes.data), log_n, batch, _u64(w),
                            _u64(cs) if cs is not None else None, int(inverse), form, MEM_HOST)
        self._check(rc, "bzh_ntt")
        return a.reshape(shape)

    def coeff_to_extended(self, field: int, coeffs: np.ndarray, log_ext: int, omega_ext: int | None = None,
                          coset_shift: int | None = None, form: int = FORM_CANONICAL) -> np.ndarray:
        """EvaluationDomain::coeff_to_extended over (batch, n, 4) or (n, 4) uint64 coefficients -> (batch, 2^log_ext, 4)."""
        a = np.ascontiguousarray(coeffs, dtype=np.uint64)
        squeeze = a.ndim == 2
        if squeeze:
            a = a.reshape(1, *a.shape)
        batch, n = a.shape[0], a.shape[1]
        log_n = n.bit_length() - 1
        if n != 1 << log_n:
            raise BzhError(E_ARG, "bzh_coeff_to_extended", "length must be a power of two")
        w = field_omega(field, log_ext, form) if omega_ext is None else int_to_limbs(omega_ext)
        sh = None if coset_shift is None else int_to_limbs(coset_shift)
        out = np.zeros((batch, 1 << log_ext, 4), dtype=np.uint64)
        L = load()
        L.bzh_coeff_to_extended.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_uint,
                                            ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                            ctypes.c_int, ctypes.c_int]
        rc = L.bzh_coeff_to_extended(self.handle, field, ctypes.c_void_p(a.ctypes.data), log_n, ctypes.c_void_p(out.ctypes.data), log_ext,
                                     batch, _u64(w), _u64(sh) if sh is not None else None, form, MEM_HOST)
        self._check(rc, "bzh_coeff_to_extended")
        return out[0] if squeeze else out

    def ntt_device(self, field: int, data_ptr: int, log_n: int, batch: int, omega: np.ndarray,
                   coset_shift: np.ndarray | None = None, inverse: bool = False, form: int = FORM_MONTGOMERY):
        rc = load().bzh_ntt(self.handle, field, ctypes.c_void_p(data_ptr), log_n, batch, _u64(omega),
                            _u64(coset_shift) if coset_shift is not None else None, int(inverse), form, MEM_DEVICE)
        self._check(rc, "bzh_ntt")


# ---- host helpers -----------------------------------------------------------
def jacobian_to_affine(curve: int, xyz: np.ndarray, form: int = FORM_CANONICAL) -> np.ndarray:
    a = np.ascontiguousarray(xyz, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros((a.shape[0], 8), dtype=np.uint64)
    rc = load().bzh_jacobian_to_affine(curve, _u64(a), a.shape[0], form, _u64(out))
    if rc != OK:
        raise BzhError(rc, "bzh_jacobian_to_affine")
    return out


def jacobian_sum(curve: int, xyz: np.ndarray, form: int = FORM_CANONICAL) -> np.ndarray:
    """Sum of Jacobian points on the host (the combine step of an MSM split over several GPUs)."""
    a = np.ascontiguousarray(xyz, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    L = load()
    L.bzh_jacobian_sum.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    rc = L.bzh_jacobian_sum(curve, _u64(a), a.shape[0], form, _u64(out))
    if rc != OK:
        raise BzhError(rc, "bzh_jacobian_sum")
    return out


def affine_compress(curve: int, xy: np.ndarray, form: int = FORM_CANONICAL) -> list:
    a = np.ascontiguousarray(xy, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros((a.shape[0], 32), dtype=np.uint8)
    rc = load().bzh_affine_compress(curve, _u64(a), a.shape[0], form, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    if rc != OK:
        raise BzhError(rc, "bzh_affine_compress")
    return [bytes(r) for r in out]


def field_omega(field: int, log_n: int, form: int = FORM_CANONICAL) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint64)
    rc = load().bzh_field_omega(field, log_n, form, _u64(out))
    if rc != OK:
        raise BzhError(rc, "bzh_field_omega")
    return out


# ---- prover-stage vector primitives (host-buffer forms; see include/bzh2.h) --------------------
def _vp(a: np.ndarray):
    return ctypes.c_void_p(a.ctypes.data)


def _bind_poly(L):
    vp = ctypes.c_void_p
    L.bzh_batch_invert.argtypes = [vp, ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
    L.bzh_prefix_product.argtypes = [vp, ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
    L.bzh_eval_polynomial.argtypes = [vp, ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_int,
                                      ctypes.c_int, vp]
    L.bzh_inner_product.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, vp]
    L.bzh_fold.argtypes = [vp, ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, vp]
    L.bzh_vec_mul.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int]


EXPORTS += ["bzh_field_convert", "bzh_random_field", "bzh_batch_invert", "bzh_prefix_product", "bzh_eval_polynomial", "bzh_inner_product", "bzh_fold", "bzh_vec_mul"]
TIMER_NAMES[5] = "poly"
TIMER_NAMES[6] = "quotient"


def _as_elems(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 4
    return a


def _ctx_batch_invert(self, field: int, data, form: int = FORM_CANONICAL) -> np.ndarray:
    a = _as_elems(data).copy()
    _bind_poly(load())
    self._check(load().bzh_batch_invert(self.handle, field, _vp(a), a.size // 4, form, MEM_HOST), "bzh_batch_invert")
    return a


def _ctx_prefix_product(self, field: int, data, form: int = FORM_CANONICAL) -> np.ndarray:
    a = _as_elems(data).copy()
    n = a.shape[-2]
    batch = a.size // 4 // n if n else 0
    _bind_poly(load())
    self._check(load().bzh_prefix_product(self.handle, field, _vp(a), n, batch, form, MEM_HOST), "bzh_prefix_product")
    return a


def _ctx_eval_polynomial(self, field: int, coeffs, xs, form: int = FORM_CANONICAL) -> np.ndarray:
    c = _as_elems(coeffs)
    if c.ndim == 2:
        c = c.reshape(1, *c.shape)
    x = _as_elems(xs).reshape(-1, 4)
    out = np.zeros((c.shape[0], 4), dtype=np.uint64)
    _bind_poly(load())
    rc = load().bzh_eval_polynomial(self.handle, field, _vp(c), c.shape[1], c.shape[0], _vp(x), x.shape[0], form, MEM_HOST, _vp(out))
    self._check(rc, "bzh_eval_polynomial")
    return out


def _ctx_inner_product(self, field: int, a, b, form: int = FORM_CANONICAL) -> np.ndarray:
    a, b = _as_elems(a), _as_elems(b)
    if a.ndim == 2:
        a, b = a.reshape(1, *a.shape), b.reshape(1, *b.shape)
    out = np.zeros((a.shape[0], 4), dtype=np.uint64)
    _bind_poly(load())
    rc = load().bzh_inner_product(self.handle, field, _vp(a), _vp(b), a.shape[1], a.shape[0], form, MEM_HOST, _vp(out))
    self._check(rc, "bzh_inner_product")
    return out


def _ctx_fold(self, field: int, v, u, form: int = FORM_CANONICAL) -> np.ndarray:
    v = _as_elems(v)
    if v.ndim == 2:
        v = v.reshape(1, *v.shape)
    u = _as_elems(u).reshape(-1, 4)
    half = v.shape[1] // 2
    out = np.zeros((v.shape[0], half, 4), dtype=np.uint64)
    _bind_poly(load())
    rc = load().bzh_fold(self.handle, field, _vp(v), half, v.shape[0], _vp(u), u.shape[0], form, MEM_HOST, _vp(out))
    self._check(rc, "bzh_fold")
    return out


def _ctx_vec_mul(self, field: int, a, b, form: int = FORM_CANONICAL) -> np.ndarray:
    a, b = _as_elems(a).copy(), _as_elems(b)
    _bind_poly(load())
    self._check(load().bzh_vec_mul(self.handle, field, _vp(a), _vp(b), a.size // 4, form, MEM_HOST), "bzh_vec_mul")
    return a


Context.batch_invert = _ctx_batch_invert
Context.prefix_product = _ctx_prefix_product
Context.eval_polynomial = _ctx_eval_polynomial
Context.inner_product = _ctx_inner_product
Context.fold = _ctx_fold
Context.vec_mul = _ctx_vec_mul


# ---- transcript (host) ---------------------------------------------------------------------------
EXPORTS += ["bzh_pk_quotient_select", "bzh_pk_quotient_selected", "bzh_quotient_source_for_circuit", "bzh_builtin_quotients",
            "bzh_quotient_degree_histogram"]
EXPORTS += ["bzh_record_stride", "bzh_record_encode", "bzh_record_decode", "bzh_record_to_json", "bzh_record_from_json"]
EXPORTS += ["bzh_transcript_new", "bzh_transcript_free", "bzh_transcript_common_point", "bzh_transcript_common_scalar",
            "bzh_transcript_write_point", "bzh_transcript_write_scalar", "bzh_transcript_squeeze_challenge",
            "bzh_transcript_proof"]


class Transcript:
    """Blake2bWrite + Challenge255 (bzh_transcript_*); ints in, ints out."""

    def __init__(self, challenge_field: int = FIELD_FP):
        L = load()
        vp = ctypes.c_void_p
        L.bzh_transcript_new.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        for f in ("bzh_transcript_free",):
            getattr(L, f).argtypes = [vp]
        for f in ("bzh_transcript_common_point", "bzh_transcript_common_scalar", "bzh_transcript_write_scalar",
                  "bzh_transcript_squeeze_challenge"):
            getattr(L, f).argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
        L.bzh_transcript_write_point.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
        L.bzh_transcript_proof.argtypes = [vp, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.POINTER(ctypes.c_size_t)]
        self.h = vp()
        rc = L.bzh_transcript_new(challenge_field, ctypes.byref(self.h))
        if rc != OK:
            raise BzhError(rc, "bzh_transcript_new")

    @staticmethod
    def _pt(pt):
        x, y = (0, 0) if pt is None else pt
        return np.concatenate([int_to_limbs(x), int_to_limbs(y)])

    def common_point(self, pt):
        rc = load().bzh_transcript_common_point(self.h, _u64(self._pt(pt)))
        if rc != OK:
            raise BzhError(rc, "bzh_transcript_common_point")

    def common_scalar(self, s: int):
        rc = load().bzh_transcript_common_scalar(self.h, _u64(int_to_limbs(s)))
        if rc != OK:
            raise BzhError(rc, "bzh_transcript_common_scalar")

    def write_point(self, curve: int, pt):
        rc = load().bzh_transcript_write_point(self.h, curve, _u64(self._pt(pt)))
        if rc != OK:
            raise BzhError(rc, "bzh_transcript_write_point")

    def write_scalar(self, s: int):
        rc = load().bzh_transcript_write_scalar(self.h, _u64(int_to_limbs(s)))
        if rc != OK:
            raise BzhError(rc, "bzh_transcript_write_scalar")

    def squeeze_challenge(self) -> int:
        out = np.zeros(4, dtype=np.uint64)
        rc = load().bzh_transcript_squeeze_challenge(self.h, _u64(out))
        if rc != OK:
            raise BzhError(rc, "bzh_transcript_squeeze_challenge")
        return limbs_to_int(out)

    def proof(self) -> bytes:
        p = ctypes.POINTER(ctypes.c_uint8)()
        n = ctypes.c_size_t()
        load().bzh_transcript_proof(self.h, ctypes.byref(p), ctypes.byref(n))
        return bytes(ctypes.string_at(p, n.value)) if n.value else b""

    def close(self):
        if self.h is not None:
            load().bzh_transcript_free(self.h)
            self.h = None


# ---- IPA opening --------------------------------------------------------------------------------
EXPORTS += ["bzh_ipa_open", "bzh_ipa_open_batch", "bzh_ipa_verify"]
EXPORTS += ["bzh_pk_create", "bzh_pk_free", "bzh_pk_set_lagrange", "bzh_pk_quotient_stats", "bzh_pk_quotient_source", "bzh_pk_set_quotient_module", "bzh_pk_info", "bzh_prove_batch", "bzh_prove_batch_seeded", "bzh_rng_expand", "bzh_verify_batch",
            "bzh_vk_digest", "bzh_pk_vk_repr"]
# Params::new (bzh2/params.py)
EXPORTS += ["bzh_hash_to_curve", "bzh_params_generators", "bzh_group_ifft", "bzh_params_create", "bzh_params_free", "bzh_params_bases",
            "bzh_params_points"]
# circuit front end (bzh2/circuits.py)
EXPORTS += ["bzh_circuit_create", "bzh_circuit_free", "bzh_circuit_last_error", "bzh_circuit_blob", "bzh_circuit_describe",
            "bzh_circuit_info", "bzh_synthesize_shot", "bzh_synthesize_board", "bzh_synthesize_bitify_test", "bzh_board_witness",
            "bzh_shot_serialize", "bzh_pedersen_commit_host", "bzh_fixed_base_tables", "bzh_circuit_set_vk_repr", "bzh_circuit_vk_repr",
            "bzh_pedersen_commit_batch"]
E_VERIFY = -6


def _ctx_ipa_open(self, bases: Bases, poly, blind: int, x3: int, rng_bytes: bytes, transcript: "Transcript",
                  form: int = FORM_CANONICAL) -> int:
    """poly::commitment::create_proof: appends S, (L_j, R_j)*, c, f to `transcript`; returns v = p(x3)."""
    L = load()
    vp = ctypes.c_void_p
    L.bzh_ipa_open.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64),
                               ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.c_size_t, vp,
                               ctypes.POINTER(ctypes.c_uint64)]
    p = _as_elems(poly)
    out = np.zeros(4, dtype=np.uint64)
    rc = L.bzh_ipa_open(self.handle, bases.handle, _vp(p), form, MEM_HOST, _u64(int_to_limbs(blind)), _u64(int_to_limbs(x3)),
                        rng_bytes, len(rng_bytes), transcript.h, _u64(out))
    self._check(rc, "bzh_ipa_open")
    return limbs_to_int(out)


def _ctx_ipa_verify(self, bases: Bases, commitment, x3: int, v: int, proof: bytes, transcript: "Transcript", g0_u_w) -> bool:
    """True iff the opening proof verifies (bzh_ipa_verify); raises on any other error."""
    L = load()
    vp = ctypes.c_void_p
    u64p = ctypes.POINTER(ctypes.c_uint64)
    L.bzh_ipa_verify.argtypes = [vp, vp, u64p, u64p, u64p, ctypes.c_char_p, ctypes.c_size_t, vp, u64p]
    cm = Transcript._pt(commitment)
    trip = np.concatenate([Transcript._pt(p) for p in g0_u_w])
    rc = L.bzh_ipa_verify(self.handle, bases.handle, _u64(cm), _u64(int_to_limbs(x3)), _u64(int_to_limbs(v)), proof, len(proof),
                          transcript.h, _u64(trip))
    if rc == E_VERIFY:
        return False
    self._check(rc, "bzh_ipa_verify")
    return True


def _ctx_ipa_open_batch(self, bases: Bases, polys, blinds, x3s, rng_list, transcripts, form: int = FORM_CANONICAL):
    """`len(transcripts)` openings in lockstep (bzh_ipa_open_batch); polys: (batch, n, 4) uint64; returns the v's."""
    L = load()
    vp = ctypes.c_void_p
    u64p = ctypes.POINTER(ctypes.c_uint64)
    L.bzh_ipa_open_batch.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, u64p, u64p, ctypes.c_char_p,
                                     ctypes.c_size_t, ctypes.POINTER(vp), u64p]
    B = len(transcripts)
    p = np.ascontiguousarray(np.asarray(polys, dtype=np.uint64).reshape(B, -1, 4))
    stride = len(rng_list[0])
    assert all(len(r) == stride for r in rng_list)
    bl = np.ascontiguousarray(np.stack([int_to_limbs(v) for v in blinds]))
    xs = np.ascontiguousarray(np.stack([int_to_limbs(v) for v in x3s]))
    out = np.zeros((B, 4), dtype=np.uint64)
    trs = (vp * B)(*[t.h for t in transcripts])
    rc = L.bzh_ipa_open_batch(self.handle, bases.handle, _vp(p), form, MEM_HOST, B, _u64(bl), _u64(xs), b"".join(rng_list), stride,
                              trs, _u64(out))
    self._check(rc, "bzh_ipa_open_batch")
    return [limbs_to_int(o) for o in out]


Context.ipa_open = _ctx_ipa_open
Context.ipa_open_batch = _ctx_ipa_open_batch
Context.ipa_verify = _ctx_ipa_verify


# ---- gate-expression evaluation (row a13) ------------------------------------------------------
EXPORTS += ["bzh_expr_eval", "bzh_expr_eval_batch"]


def _ctx_expr_eval(self, field: int, program, columns, form: int = FORM_CANONICAL) -> np.ndarray:
    """Evaluate a straight-line program (bzh_expr_op records: anything with .as_array() -> ctypes array of them, .ops, .consts,
    .result_slot; tests/helpers/expr.py compiles expression trees into one) at every row; columns: list of (size, 4) uint64 arrays."""
    L = load()
    vp = ctypes.c_void_p
    L.bzh_expr_eval.argtypes = [vp, ctypes.c_int, vp, ctypes.c_size_t, ctypes.POINTER(vp), ctypes.c_size_t,
                                vp, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    cols = [_as_elems(c) for c in columns]
    size = cols[0].shape[0] if cols else 1
    log_size = size.bit_length() - 1
    assert all(c.shape[0] == size for c in cols) and (1 << log_size) == size
    ptrs = (vp * max(len(cols), 1))(*[c.ctypes.data for c in cols])
    consts = np.ascontiguousarray(np.stack([int_to_limbs(v) for v in program.consts]) if program.consts
                                  else np.zeros((1, 4), dtype=np.uint64))
    out = np.zeros((size, 4), dtype=np.uint64)
    ops = program.as_array()
    rc = L.bzh_expr_eval(self.handle, field, ctypes.cast(ops, vp), len(program.ops), ptrs, len(cols), _vp(consts), len(program.consts), log_size,
                         program.result_slot, form, MEM_HOST, _vp(out))
    self._check(rc, "bzh_expr_eval")
    return out


Context.expr_eval = _ctx_expr_eval


# ---- multiopen / lookup helpers ---------------------------------------------------------------------
EXPORTS += ["bzh_kate_division", "bzh_kate_division_batch", "bzh_permute_expression_pair"]


def _ctx_kate_division(self, field: int, coeffs, x: int, form: int = FORM_CANONICAL) -> np.ndarray:
    """arithmetic::kate_division: quotient of p(X) by (X - x) (n-1 coefficients)."""
    c = _as_elems(coeffs)
    n = c.shape[0]
    out = np.zeros((max(n - 1, 0), 4), dtype=np.uint64)
    L = load()
    vp = ctypes.c_void_p
    L.bzh_kate_division.argtypes = [vp, ctypes.c_int, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_int, vp]
    self._check(L.bzh_kate_division(self.handle, field, _vp(c), n, _u64(int_to_limbs(x)), form, MEM_HOST, _vp(out)), "bzh_kate_division")
    return out


Context.kate_division = _ctx_kate_division


def _ctx_kate_division_batch(self, field: int, coeffs, xs, form: int = FORM_CANONICAL) -> np.ndarray:
    """`batch` kate divisions in one pass: coeffs (batch, n, 4), xs a list of `batch` integers; returns (batch, n-1, 4)."""
    c = np.ascontiguousarray(coeffs, dtype=np.uint64)
    batch, n = c.shape[0], c.shape[1]
    out = np.zeros((batch, max(n - 1, 0), 4), dtype=np.uint64)
    xv = np.ascontiguousarray(np.stack([int_to_limbs(x) for x in xs]), dtype=np.uint64)
    L = load()
    vp = ctypes.c_void_p
    L.bzh_kate_division_batch.argtypes = [vp, ctypes.c_int, vp, ctypes.c_size_t, ctypes.c_size_t, vp, ctypes.c_int, ctypes.c_int, vp]
    self._check(L.bzh_kate_division_batch(self.handle, field, _vp(c), n, batch, _vp(xv), form, MEM_HOST, _vp(out)),
                "bzh_kate_division_batch")
    return out


Context.kate_division_batch = _ctx_kate_division_batch


def permute_expression_pair(field: int, input_vals, table_vals, usable_rows: int, form: int = FORM_CANONICAL):
    """lookup::prover::permute_expression_pair over the first usable_rows rows -> (permuted_input, permuted_table)."""
    a, t = _as_elems(input_vals), _as_elems(table_vals)
    oa = np.zeros((usable_rows, 4), dtype=np.uint64)
    ot = np.zeros((usable_rows, 4), dtype=np.uint64)
    L = load()
    vp = ctypes.c_void_p
    L.bzh_permute_expression_pair.argtypes = [ctypes.c_int, vp, vp, ctypes.c_size_t, ctypes.c_int, vp, vp]
    rc = L.bzh_permute_expression_pair(field, _vp(a), _vp(t), usable_rows, form, _vp(oa), _vp(ot))
    if rc != OK:
        raise BzhError(rc, "bzh_permute_expression_pair")
    return oa, ot
