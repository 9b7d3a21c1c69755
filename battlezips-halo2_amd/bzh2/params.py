"""Params::new (halo2_proofs poly::commitment::Params; benches/shot.rs:58, benches/board.rs:51) through the C ABI
(include/bzh2.h, csrc/params.hip): hash-to-curve SRS on the host, g_lagrange by a group FFT on the device, both cached
on disk keyed by (curve, k)."""
from __future__ import annotations

import ctypes

import numpy as np

from . import CURVE_PALLAS, CURVE_VESTA, Bases, BzhError, Context, limbs_to_int, load

_VP = ctypes.c_void_p


def _bind():
    L = load()
    if getattr(L, "_bzh_params_bound", False):
        return L
    L.bzh_hash_to_curve.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, _VP]
    L.bzh_params_generators.argtypes = [ctypes.c_uint, _VP, _VP, _VP, ctypes.c_uint]
    L.bzh_group_ifft.argtypes = [_VP, ctypes.c_int, _VP, ctypes.c_uint, _VP]
    L.bzh_params_create.argtypes = [_VP, ctypes.c_uint, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(_VP)]
    L.bzh_params_free.argtypes = [_VP, _VP]
    L.bzh_params_bases.argtypes = [_VP, ctypes.POINTER(_VP), ctypes.POINTER(_VP)]
    L.bzh_params_points.argtypes = [_VP, _VP, _VP, _VP, _VP, ctypes.POINTER(ctypes.c_int)]
    L._bzh_params_bound = True
    return L


def hash_to_curve(curve: int, domain_prefix: str, message: bytes):
    """CurveExt::hash_to_curve(domain_prefix)(message) -> affine (x, y) ints (host)."""
    out = np.zeros(8, dtype=np.uint64)
    rc = _bind().bzh_hash_to_curve(curve, domain_prefix.encode(), message, len(message), _VP(out.ctypes.data))
    if rc:
        raise BzhError(rc, "bzh_hash_to_curve")
    return limbs_to_int(out[:4]), limbs_to_int(out[4:])


def generators(k: int, threads: int = 0):
    """(g, w, u) of Params::new(k): g as an (n, 8) uint64 array of canonical limbs, w / u as (x, y) ints."""
    n = 1 << k
    g, w, u = np.zeros((n, 8), dtype=np.uint64), np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    rc = _bind().bzh_params_generators(k, _VP(g.ctypes.data), _VP(w.ctypes.data), _VP(u.ctypes.data), threads)
    if rc:
        raise BzhError(rc, "bzh_params_generators")
    return g, (limbs_to_int(w[:4]), limbs_to_int(w[4:])), (limbs_to_int(u[:4]), limbs_to_int(u[4:]))


def group_ifft(ctx: Context, g: np.ndarray) -> np.ndarray:
    g = np.ascontiguousarray(g, dtype=np.uint64)
    n = g.shape[0]
    k = n.bit_length() - 1
    out = np.zeros((n, 8), dtype=np.uint64)
    ctx._check(_bind().bzh_group_ifft(ctx.handle, CURVE_VESTA, _VP(g.ctypes.data), k, _VP(out.ctypes.data)), "bzh_group_ifft")
    return out


class Params:
    """Params::<vesta::Affine>::new(k): the two commitment-base tables live on the device."""

    def __init__(self, ctx: Context, k: int, cache_dir: str | None = None, window_bits: int = 0):
        L = _bind()
        h = _VP()
        ctx._check(L.bzh_params_create(ctx.handle, k, None if cache_dir is None else cache_dir.encode(), window_bits, ctypes.byref(h)),
                   "bzh_params_create")
        self.ctx, self.handle, self.k, self.n = ctx, h, k, 1 << k
        g, gl = _VP(), _VP()
        L.bzh_params_bases(h, ctypes.byref(g), ctypes.byref(gl))
        self.bases = Bases(ctx, g, CURVE_VESTA, self.n + 2)
        self.bases_lagrange = Bases(ctx, gl, CURVE_VESTA, self.n + 3)    # (g_lagrange | u | w | g_0)
        self._borrowers = []     # weak references to the proving keys built on these tables (they borrow, include/bzh2.h)

    def points(self, want_g: bool = True, want_lagrange: bool = True):
        """host copies: (g, g_lagrange, w, u, from_cache) -- arrays of canonical limbs, w / u as (x, y) ints"""
        g = np.zeros((self.n, 8), dtype=np.uint64) if want_g else None
        gl = np.zeros((self.n, 8), dtype=np.uint64) if want_lagrange else None
        w, u = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        fc = ctypes.c_int()
        _bind().bzh_params_points(self.handle, _VP(g.ctypes.data) if want_g else None, _VP(gl.ctypes.data) if want_lagrange else None,
                                  _VP(w.ctypes.data), _VP(u.ctypes.data), ctypes.byref(fc))
        return g, gl, (limbs_to_int(w[:4]), limbs_to_int(w[4:])), (limbs_to_int(u[:4]), limbs_to_int(u[4:])), bool(fc.value)

    def close(self, strict: bool = False):
        """bzh_params_free.  Proving keys borrow the tables: freeing them under a live key would leave it reading freed window
        tables.  close() is what cleanup paths (`finally:`) call, so by default the keys still open on these Params are closed
        first, with a warning -- raising here would mask the exception that skipped their close() and leak the device tables.
        strict=True keeps the hard error for callers that want the ordering enforced."""
        if self.handle is not None:
            live = [r() for r in self._borrowers if r() is not None and r().handle is not None]
            self._borrowers = []          # dead references are dropped either way
            if live and strict:
                self._borrowers = [__import__("weakref").ref(k) for k in live]
                raise RuntimeError("Params.close(strict=True): a NativeProvingKey built on these Params is still open; close the key first")
            if live:
                import warnings
                warnings.warn("Params.close(): closing %d NativeProvingKey(s) still open on these Params first" % len(live), ResourceWarning, stacklevel=2)
                for key in live:
                    key.close()
            _bind().bzh_params_free(self.ctx.handle, self.handle)
            self.handle = None
            self.bases.handle = None            # the tables went with the params
            self.bases_lagrange.handle = None
